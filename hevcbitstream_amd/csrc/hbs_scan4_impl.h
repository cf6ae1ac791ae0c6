/*
 * hbs_scan4_impl.h -- K12, event-sparse form: fused start-code scan + NAL index +
 * RBSP extraction whose per-byte work is one conservative test and one copy.
 *
 * This file is the kernel's SOURCE, compiled once per geometry (HBS4_ROWS rows of 1 KiB per wavefront):
 *   hbs_scan4.hip      48 rows, 192 KiB tiles, up to 512 elements a tile, two element wavefronts that park 20 rows each:
 *                      coded video -- a start code per several KiB (hbs::k_scan_extract4; also the call's prologue, the
 *                      count-ahead and the host-side helpers, which exist once);
 *   hbs_scan4_r24.hip  24 rows, 96 KiB tiles, up to 1024 elements a tile, all four wavefronts on the element batches, nothing
 *                      parked (96 row registers leave room for the element code): streams that are dense but regular -- NALs of
 *                      ~120 to ~450 bytes, one chunk in 6.5 to one in 26 an element (hbs::k_scan_extract4_r24, round 6).
 * The text below says "48 rows" where it describes the first; the row lists themselves are generated (scripts/set_rows4.py).
 *
 * Same contract, descriptors and tile algebra as hbs_scan.hip (reference loop
 * find_nal_unit + nal_to_rbsp, h264_nal.c:38-76 / :147-200, driven as in
 * hevc_analyze.c:135-177).  What differs is who does the exact work:
 *
 *   1. A workgroup is 4 wavefronts of 256 VGPRs; wavefront w holds 48 rows of
 *      1 KiB in named registers (a tile = 192 KiB), fetched inside the flag
 *      pass, a few rows ahead.  Per 16-byte chunk, chunk_flag() (hbs_sparse.h)
 *      decides that no two adjacent zero bytes touch it; a row with more than
 *      two such chunks is asked again, exactly: does a pattern 00 00 {<=3} end
 *      in the chunk (chunk_pattern_any_dev, hbs_wave.h)?  The row's ballot is
 *      its flag mask.  Flagged chunks -- a start code per NAL, a few emulation
 *      prevention bytes, a few false alarms: ~15 of 12288 in coded video --
 *      are listed in LDS per wavefront, in stream order, as they are found, and
 *      their lanes leave the chunk's surroundings in LDS.
 *   2. Wavefront 0 takes the listed chunks as "elements", one per lane: exact
 *      window logic of hbs_tile.h on bytes [-8, 20) of the chunk.  A wave scan
 *      (DPP) with combine() over (gap, chunk) elements gives the tile aggregate.
 *      A tile with more than 64 elements (small NALs, zero-heavy data) has
 *      several batches: wavefront 1 takes every other one.
 *   3. Wavefront 0 runs the decoupled look-back, 256 predecessors per step;
 *      the others wait at a barrier.  Part of an element wavefront's rows are
 *      parked in LDS during 2-4, which need ~100 registers of their own.
 *   4. With the carried state known the elements emit index entries, write
 *      their own kept bytes, and leave one segment word each; every other chunk
 *      finds the word of the nearest element in front of it and, if inside a
 *      NAL, is one byte-aligned 16-byte store straight from its registers
 *      (at most three stores of a wavefront in flight: the CU's memory queue
 *      is shared with the other workgroup's look-back polls).
 *
 * Tiles are handed out by an atomic ticket in arrival order, so a workgroup
 * only ever waits for tiles that are already being worked on: no co-residency
 * requirement, and a slow workgroup delays its successors, not a whole round.
 * Tiles with more than kDenseElems elements (padding, zero stuffing) stay
 * exact the other way round: every chunk is an element, every wavefront walks
 * its own rows (dense_tile).
 */
#include <hip/hip_runtime.h>
#ifndef HBS4_ROWS
#error "compile hbs_scan4.hip / hbs_scan4_r24.hip, which set HBS4_ROWS and include this file"
#endif
#include "hbs_wave.h"
#include "hbs_sparse.h"
#include "hbs_scan.h"
#include "hbs_elems.h"
#if HBS4_ROWS == 48
#include "hbs_scan4_rows48.h"
#define HBS4_KERNEL k_scan_extract4
#elif HBS4_ROWS == 24
#include "hbs_scan4_rows24.h"
#define HBS4_KERNEL k_scan_extract4_r24
#else
#error "no row lists for this HBS4_ROWS: scripts/set_rows4.py"
#endif
#define HBS4_MAIN (HBS4_ROWS == 48)     /* the translation unit that also holds what exists once per library */

namespace hbs {
namespace {                              /* everything but the kernels and the launchers is this translation unit's own */

#ifdef HBS_PHASE_TIMING
static __device__ unsigned long long g_phase_cycles4[1024][8];
#define HBS4_T_DECL unsigned long long t_prev = __builtin_amdgcn_s_memtime(), t_acc[8] = {0,0,0,0,0,0,0,0};
#define HBS4_T_MARK(i) { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); t_acc[i] += t_now - t_prev; t_prev = t_now; }
#define HBS4_T_COUNT(i, v) { t_acc[i] += (v); }
#define HBS4_T_FLUSH if (threadIdx.x == 0 && blockIdx.x < 1024) { for (int i = 0; i < 8; ++i) g_phase_cycles4[blockIdx.x][i] = t_acc[i]; }
static __device__ uint32_t g_dbg4[4096];
/* a timeline per tile (wall clock, 10 ns): taken / aggregate known / look-back done / finished; bit 0 of [1]: walked as a dense tile */
constexpr unsigned kTl4Tiles = 1u << 17;
static __device__ unsigned long long g_tl4[kTl4Tiles][4];
static __device__ unsigned long long g_tlwho4[kTl4Tiles];      /* who took the tile: workgroup | HW_ID << 32, XCC_ID in bits 28-31 of the low word */
#define HBS4_TL(tile, k, bit) { if (threadIdx.x == 0 && (tile) < kTl4Tiles) { g_tl4[tile][k] = (wall_clock64() & ~1ull) | (unsigned long long)(bit); \
    if ((k) == 0) g_tlwho4[tile] = (unsigned long long)blockIdx.x | ((unsigned long long)(__builtin_amdgcn_s_getreg(63508) & 15u) << 28) | \
                                   ((unsigned long long)__builtin_amdgcn_s_getreg(63492) << 32); } }
static __device__ int g_fake_lb4 = 0;       /* experiments: 1 = do not look back at all (wrong results, timing only) */
#define HBS4_DBG(code) code
#else
#define HBS4_DBG(code)
#define HBS4_TL(tile, k, bit)
#define HBS4_T_DECL
#define HBS4_T_MARK(i)
#define HBS4_T_COUNT(i, v)
#define HBS4_T_FLUSH
#endif

#ifndef HBS4_WG_PER_CU
#define HBS4_WG_PER_CU 2      /* workgroups per CU = wavefronts per SIMD the register budget is cut for */
#endif
#ifndef HBS4_TICKET_BARRIER
#define HBS4_TICKET_BARRIER 0  /* 1: a barrier in front of the ordinary tile's ticket too (wavefront 0, which takes it, is nearly always the last to finish) */
#endif
#ifndef HBS4_PROGRESSIVE
#define HBS4_PROGRESSIVE 1     /* the fetch inside the flag pass, four rows at a time */
#endif
#ifndef HBS4_EXACT_FLAG
#define HBS4_EXACT_FLAG 1      /* rows with a zero pair are asked again, exactly: only chunks a pattern 00 00 {<=3} touches become elements */
#endif
#ifndef HBS4_ELEM_WAVES
#if HBS4_ROWS <= 24
#define HBS4_ELEM_WAVES 4      /* 24 rows: nothing is parked, every wavefront takes element batches */
#define HBS4_EMIT_WAVES 4
#endif
#endif
#ifndef HBS4_ELEM_WAVES
#define HBS4_ELEM_WAVES 2      /* wavefronts that share the batches of a tile with more than 64 elements (each parks kParkRows rows in LDS) */
#endif
#ifndef HBS4_EMIT_WAVES
#define HBS4_EMIT_WAVES 2      /* wavefronts that share the second half (emission) of those batches (4: the other two with all their rows in place -- spills 45 registers, some on every tile's path) */
#endif
#ifndef HBS4_EXACT_MIN
#define HBS4_EXACT_MIN kExactFlagMin       /* ... when the row has more flagged chunks than this */
#endif
#ifndef HBS4_COPY_DEPTH
#define HBS4_COPY_DEPTH 3      /* stores of a wavefront in flight during the copy (-1: no limit) */
#endif
#ifdef HBS4_NO_PRIO
#define HBS4_PRIO(p)
#else
#define HBS4_PRIO(p) __builtin_amdgcn_s_setprio(p)
#endif
static_assert(k4Rows == HBS4_ROWS, "hbs_sparse.h takes the geometry from HBS4_ROWS");
constexpr int kParkRows = HBS4_PARK_ROWS;          /* rows of an element wavefront that wait in LDS meanwhile (hbs_scan4_rows*.h) */

/* one wavefront's segment: k4Rows rows of 1 KiB in named registers + the dwords just outside */
struct RowRegs {
#define HBS_DECL(r) u32x4 q##r;
    HBS_ROWS(HBS_DECL)
#undef HBS_DECL
    uint32_t before;          /* dword in front of the segment (0xFFFFFFFF before the stream) */
    uint32_t before2;         /* the dword in front of that one                               */
    uint32_t after;           /* dword behind it (0xFF bytes past the end of the stream)      */
};

/* All of a wavefront's rows, unguarded: the last tile of a stream is read from a padded copy
 * (k_scan_prologue), so every address below exists.  `src` is the stream or that copy. */
__device__ __forceinline__ void fetch_row_regs(RowRegs& R, const uint8_t* __restrict__ src, uint64_t seg, int lane)
{
    const u32x4* p = reinterpret_cast<const u32x4*>(src + seg) + lane;
#define HBS_LD(r) R.q##r = stream_load16(p + r * 64);
    HBS_ROWS(HBS_LD)
#undef HBS_LD
}

constexpr int kDepCap = 64;
constexpr uint32_t kDenseElems = HBS4_ROWS == 48 ? 512 : 1024;      /* a tile with more elements than this is walked by rows (dense tiles, below): 4.2 % of a 192 KiB tile's chunks, 16.7 % of a 96 KiB tile's */

struct Lds4 {
    uint32_t wave_tot[k4Waves];            /* elements per wavefront                        */
    uint16_t wlist[k4Waves][kDenseElems];  /* flagged chunks of each wavefront's rows, in stream order, written during the flag pass (a tile with more than
                                              kDenseElems takes the dense path); element i of the tile = entry i - (elements of the wavefronts in front) */
    uint32_t seg[kDenseElems + 1];         /* segment words: [0] tile start, [i+1] element i of the tile */
    u32x4 rec[kDenseElems][3];             /* tiles with several batches of elements: what the first walk found out about each
                                              (marks, summary, classes, its bytes), so that the second half does not walk it again */
    Deposit dep[k4Waves][kDepCap];         /* bytes of the first elements of each wavefront, left by the flag pass */
    u32x4 park[kParkRows ? HBS4_ELEM_WAVES : 1][kParkRows ? kParkRows : 1][kParkRows ? 64 : 1];   /* rows of the wavefronts that handle elements (wavefront 0 always, and it looks back) meanwhile */
    unsigned long long ex_kept, ex_nals;   /* the tile's exclusive prefix, from wavefront 0 */
    uint32_t ex_inside, ex_ok;
    uint32_t ticket;
    TileAgg wagg[k4Waves];                 /* dense tiles: the aggregate of each wavefront's rows */
    TileAgg bagg[kDenseElems / k4ElemPass];/* tiles with several batches of elements: the aggregate of each batch */
};

/* ---- dense tiles -------------------------------------------------------------------------------------
 * A tile in which zero pairs are everywhere (cabac_zero_words or 00 00 03 padding, zero stuffing between NALs) would
 * keep wavefront 0 walking its thousands of elements 64 at a time while the look-backs of every tile behind it wait:
 * a 1 % share of such bytes made a stream 7.6 times slower (scripts/mixed_time.py).  Past kDenseElems elements a tile is
 * therefore handled the other way round: EVERY chunk is an element (no gaps, no lists, no deposits), each wavefront
 * walks its own 48 rows -- one row per step, one chunk per lane, the bit-parallel rules of hbs_sparse.h -- and the four
 * wavefront aggregates meet in LDS.  The rows are read again for it (they sit in the cache: a rolled loop over 48 named
 * registers does not exist, and unrolled the walk would be 150 KB of code).  Rows without any terminator, which is
 * what padding looks like, fold with one add per chunk; only rows that hold an event pay for the ordered scan. */

/* The rows of a wavefront's segment again, in order, for the two halves of a dense tile: f(r, previous row, row, next row).
 * They are read back from memory (the cache, mostly) kDenseAhead rows at a time, the next batch's loads in flight while
 * this one is walked: with one row in flight -- round 2 -- every row of the aggregate half cost a memory round trip, and that
 * half is what every tile behind a dense one waits for. */
constexpr int kDenseAhead = 8;
static_assert(k4Rows % kDenseAhead == 0, "whole batches of rows");
template <class F>
__device__ __forceinline__ void dense_rows(const uint8_t* src, uint64_t wseg, int lane, F&& f)
{
    u32x4 cur[kDenseAhead + 2];                                   /* rows b - 1 .. b + kDenseAhead */
#pragma unroll
    for (int i = 0; i <= kDenseAhead; ++i) cur[i + 1] = dense_fetch(src, wseg, i < k4Rows ? i : k4Rows - 1, lane);
    cur[0] = cur[1];
#pragma unroll 1
    for (int b = 0; b < k4Rows; b += kDenseAhead) {
        u32x4 nxt[kDenseAhead];                                   /* rows b + kDenseAhead + 1 .. b + 2 kDenseAhead */
#pragma unroll
        for (int i = 0; i < kDenseAhead; ++i) {
            const int r = b + kDenseAhead + 1 + i;
            nxt[i] = dense_fetch(src, wseg, r < k4Rows ? r : k4Rows - 1, lane);
        }
#pragma unroll
        for (int i = 0; i < kDenseAhead; ++i) f(b + i, cur[i], cur[i + 1], cur[i + 2]);
        cur[0] = cur[kDenseAhead]; cur[1] = cur[kDenseAhead + 1];
#pragma unroll
        for (int i = 0; i < kDenseAhead; ++i) cur[i + 2] = nxt[i];
    }
}

/* first half: the aggregate of this wavefront's rows */
__device__ __forceinline__ TileAgg dense_aggregate(const uint8_t* src, uint64_t wseg, uint64_t n, uint32_t before, uint32_t before2, uint32_t after,
                                                   uint32_t chunk0, int lane)
{
    TileAgg acc = agg_identity();
    dense_rows(src, wseg, lane, [&](int r, const u32x4& qp, const u32x4& qc, const u32x4& qn) {
        uint32_t kept;
        if (wseg + 1024ull * (uint64_t)(r + 1) + 64ull <= n && !dense_row_quick<false>(qp, qc, qn, r, k4Rows, before, after, kept)) {
            acc = combine(acc, gap_agg(wave_sum32(kept)));               /* no terminator anywhere in the row: kept bytes, state-dependent */
            return;
        }
        DenseRow d;
        dense_row(d, qp, qc, qn, r, k4Rows, before, before2, after, src, wseg, n, chunk0, lane);
        if (!d.row_has_event) {
            acc = combine(acc, gap_agg(wave_sum32(d.el.s.carry)));       /* chunks without a terminator: (0, 0, carry, none) each */
        } else {
            const TileAgg ea = wave_scan_combine(elem_agg(0u, d.el.s), lane);
            acc = combine(acc, agg_readlane(ea, 63));
        }
    });
    return acc;
}

/* second half: index entries and kept bytes of this wavefront's rows; acc0 = aggregate of the tile in front of them */
__device__ __forceinline__ void dense_emit(const uint8_t* src, uint64_t wseg, uint64_t n, uint32_t before, uint32_t before2, uint32_t after,
                                           uint32_t chunk0, int lane, TileAgg acc0, const Prefix& excl, bool can_store, uint8_t* out,
                                           const EmitTarget& tgt, uint32_t* scratch_word)
{
    TileAgg acc = acc0;
    dense_rows(src, wseg, lane, [&](int r, const u32x4& qp, const u32x4& qc, const u32x4& qn) {
        DenseRow d;
        dense_row(d, qp, qc, qn, r, k4Rows, before, before2, after, src, wseg, n, chunk0, lane);
        const TileAgg ea = wave_scan_combine(elem_agg(0u, d.el.s), lane);
        TileAgg up = agg_prev_lane(ea);
        if (lane == 0) up = agg_identity();
        const TileAgg e = combine(acc, up);
        acc = combine(acc, agg_readlane(ea, 63));
        if (d.el.v.g0 < n) elem_emit(d.el, e, excl, can_store, out, tgt, scratch_word);
    });
}


/* lane `l` of v <- the wave-uniform value s.  The s_nop covers gfx950's wait states between a
 * VALU instruction that writes an SGPR (the v_cmp of a ballot) and a VALU instruction reading
 * it, which the compiler cannot insert across an asm statement. */
#define write_lane(v, s, l) asm volatile("s_nop 1\n\tv_writelane_b32 %0, %1, " #l : "+v"(v) : "s"(s))



__device__ __forceinline__ void rec4_store(u32x4* r, const Elem& el)
{
    const ElemPacked p = elem_pack(el.m, el.s);
    u32x4 a, b, c;
    a.x = el.chunk; a.y = el.gap; a.z = p.a; a.w = p.b;
    b.x = p.c; b.y = el.cls.z; b.z = el.cls.e1; b.w = el.cls.e3;
    c.x = el.v.x0; c.y = el.v.x1; c.z = el.v.x2; c.w = el.v.x3;
    r[0] = a; r[1] = b; r[2] = c;
}
__device__ __forceinline__ void rec4_load(const u32x4* r, Elem& el, const uint8_t* src, uint64_t base, uint64_t n)
{
    const u32x4 a = r[0], b = r[1], c = r[2];
    ElemPacked p; p.a = a.z; p.b = a.w; p.c = b.x;
    elem_unpack(p, el.m, el.s);
    el.cls.z = b.y; el.cls.e1 = b.z; el.cls.e3 = b.w;
    el.chunk = a.x; el.gap = a.y;
    el.v.x0 = c.x; el.v.x1 = c.y; el.v.x2 = c.z; el.v.x3 = c.w;
    el.v.xpp = el.v.xp = el.v.xn = 0;                    /* the second half looks at the chunk's own bytes only */
    el.v.stream = src; el.v.g0 = base + 16ull * a.x; el.v.n = n;
}

/* second half of the batches first, first + HBS4_EMIT_WAVES, ... of a tile with several: index entries, the elements' own
 * bytes, a segment word each; the tile's exclusive prefix is in LDS by now */
__device__ __forceinline__ void emit_batches(Lds4& l, uint32_t first, uint32_t npass, uint32_t nflag, int lane,
                                             const uint8_t* src, uint64_t base, uint64_t n, uint8_t* rbsp, const EmitTarget& tgt)
{
    Prefix exl;
    exl.kept = l.ex_kept; exl.nals = l.ex_nals; exl.inside = l.ex_inside;
    const Prefix exu = prefix_uniform4(exl);
    const bool canu = rbsp != nullptr && l.ex_ok == 1u;
#pragma unroll 1
    for (uint32_t p = first; p < npass; p += (uint32_t)HBS4_EMIT_WAVES) {
        TileAgg accb = agg_identity();
#pragma unroll 1
        for (uint32_t q = 0; q < p; ++q) accb = combine(accb, l.bagg[q]);
        const uint32_t i = p * (uint32_t)k4ElemPass + (uint32_t)lane;
        TileAgg ea = agg_identity();
        Elem el;
        el.gap = 0; el.chunk = 0;
        if (i < nflag) { rec4_load(l.rec[i], el, src, base, n); ea = elem_agg(el.gap, el.s); }
        ea = wave_scan_combine(ea, lane);
        TileAgg up = agg_prev_lane(ea);
        if (lane == 0) up = agg_identity();
        const TileAgg eb = combine(accb, up);
        if (i < nflag) elem_emit(el, eb, exu, canu, rbsp + exu.kept, tgt, &l.seg[i + 1]);
    }
}

/* chunk number of element i of the tile; wb1..wb3 = elements in front of wavefronts 1..3 */
__device__ __forceinline__ uint32_t list_at(const Lds4& l, uint32_t i, uint32_t wb1, uint32_t wb2, uint32_t wb3)
{
    const uint32_t ew = (i >= wb1 ? 1u : 0u) + (i >= wb2 ? 1u : 0u) + (i >= wb3 ? 1u : 0u);
    const uint32_t ej = i - (ew == 0u ? 0u : ew == 1u ? wb1 : ew == 2u ? wb2 : wb3);
    return l.wlist[ew][ej < kDenseElems ? ej : 0u];
}

/* Element i of the tile (lane = i mod 64 of wavefront 0): its bytes come from the deposit its
 * flagging lane left in LDS, or from the stream when there is none; then the exact window rules. */
__device__ __forceinline__ TileAgg elem_make(Elem& el, const Lds4& l, uint32_t i, uint32_t wb1, uint32_t wb2, uint32_t wb3,
                                             const uint8_t* __restrict__ src, uint64_t base, uint64_t n, bool padded)
{
    /* which wavefront flagged it, and as its how-manieth element */
    const uint32_t ew = (i >= wb1 ? 1u : 0u) + (i >= wb2 ? 1u : 0u) + (i >= wb3 ? 1u : 0u);
    const uint32_t ej = i - (ew == 0u ? 0u : ew == 1u ? wb1 : ew == 2u ? wb2 : wb3);
    const uint32_t c = l.wlist[ew][ej < kDenseElems ? ej : 0u];
    const uint64_t prev_end = (i > 0) ? base + 16ull * (list_at(l, i - 1u, wb1, wb2, wb3) + 1u) : base;
    /* field by field: a conditional copy of the whole struct ends up in scratch memory, and every later use of the
     * element's bytes then waits for all outstanding memory operations (s_waitcnt vmcnt(0)) to read them back */
    const bool have_dep = ej < (uint32_t)kDepCap && l.dep[ew < (uint32_t)k4Waves ? ew : 0u][ej < (uint32_t)kDepCap ? ej : 0u].chunk == c;
    if (have_dep) {
        const Deposit& d = l.dep[ew][ej];
        el.v.xpp = d.xpp; el.v.xp = d.xp; el.v.x0 = d.x0; el.v.x1 = d.x1; el.v.x2 = d.x2; el.v.x3 = d.x3; el.v.xn = d.xn;
        el.v.stream = src; el.v.g0 = base + 16ull * c; el.v.n = n;
    } else {
        elem_load(el.v, src, base + 16ull * c, n, padded);
    }
    elem_walk(el.v, el.m, el.s, el.cls);
    el.gap = span_bytes(prev_end, el.v.g0, n);
    el.chunk = c;
    return elem_agg(el.gap, el.s);
}


/* One dense tile, by the whole workgroup (every thread calls it): aggregates, look-back, emission, the next ticket.  A function
 * of its own, not inlined: inlined, its registers add to the 192 the rows occupy and the COMMON path spills 33 of them around
 * every tile's first barrier (index-only scans ran 23 % slower).  false: a look-back timed out, the workgroup gives up. */
/* ---- dense tiles counted ahead (round 5) ---------------------------------------------------------------------------------
 * A dense tile's aggregate is out only when its four wavefronts have walked their 48 rows each -- 40-45 us after the tile was
 * taken, where an ordinary tile's is out after ~15 -- and every tile behind it waits for it in its look-back: 25 us of the
 * whole GPU wherever a stretch of padding or zero stuffing begins (the tiles inside the stretch are walked at the same time as
 * its first one).  But a tile's aggregate does not depend on anything in front of the tile.  So, as k3_tiles does since round 4:
 * the prologue samples every tile -- first a chunk in every 64 KiB, which a stretch that long cannot avoid; a tile that shows
 * something, and the tile on either side of it (where the stretch begins and ends), get the full look: a chunk in every 4 KiB (a
 * tile is dense from 512 flagged chunks = 8 KiB of such a stretch; coded video flags one sample in 500) -- and lists the tiles
 * it marks; k_scan_ahead4 takes the tiles so marked, in front of the main kernel and with nobody waiting -- counts their flagged
 * chunks roughly, and walks the ones that may be dense: their four wavefront aggregates go to a table, the tile's word becomes
 * "counted"; dense_tile takes such a tile's entry instead of walking its rows a first time (the words carry the call's stamp --
 * a number kept on the device and advanced by the call's last launch, so that a call replayed from a HIP graph is a new call --:
 * nothing has to be cleared).  A dense tile the sample misses -- a stretch shorter than 64 KiB may be -- is walked in place as
 * before; a marked tile that turns out ordinary costs its rows once more, read by a kernel that has the memory system to
 * itself.  From kAheadMinBytes up (below, the extra launch costs a call more than mixed content is likely to;
 * hbs_ctx_set_count_ahead).  The bench's mixed stream (16 GiB, 1 % of it in 640 KiB stretches): 7.65 -> 6.4-6.5 ms, 1.29 ->
 * 1.05-1.10 x the uniform stream's time. */
struct AheadEntry { TileAgg w[k4Waves]; };
static_assert(sizeof(AheadEntry) == 64, "four aggregates");
constexpr uint64_t kAheadMinBytes = 3ull << 30;          /* the launch costs ~10 us where nothing is marked: 1.3 % of a 2 GiB call, 0.9 % at 3 GiB, 0.15 % at 16 */
constexpr int kAheadSample = 48;                         /* chunks sampled per tile, 4 KiB apart */
constexpr int kAheadMinHits = 8;                         /* marked from this many flagged samples, or from three in a row (streams of 512-byte NALs
                                                            flag a sample in 20: two in a row marked a tile in 9, and the kernel below took 0.23 ms of 1.2) */
constexpr uint32_t kAheadRoughMin = 400;                 /* k_scan_ahead4 walks a marked tile when its rough count reaches this (chunks taken by themselves:
                                                            a pattern across two chunks is missed, one in eight) */
constexpr unsigned long long kAheadMarked = 1ull, kAheadDone = 2ull;     /* or-ed to the call's stamp in a tile's word */
constexpr int kAheadCoarse = 3;                          /* chunks of the first look, 64 KiB apart */

__device__ __attribute__((noinline))
bool dense_tile(Lds4& l, const uint8_t* src, uint64_t wseg, uint64_t n, uint32_t before, uint32_t before2, uint32_t after,
                uint64_t tile, bool last_tile, uint8_t* rbsp, uint64_t rbsp_cap, unsigned long long* desc, RunHeader* hdr,
                hbs_nal_entry* index, uint64_t index_cap, uint32_t ticket_base)
{
    EmitTarget tgt;                      /* built here: handed over by reference it had to live in scratch memory for the whole kernel */
    tgt.index = index; tgt.index_cap = index_cap; tgt.hdr = hdr;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t chunk0 = (uint32_t)(64 * k4Rows * wv);
    /* counted ahead in this call (k_scan_ahead4)?  then the table has what the walk below would find */
    const bool counted = HBS4_MAIN && hdr->ahead_tab != 0ull && !last_tile && reinterpret_cast<const unsigned long long*>(hdr->ahead_cand)[tile] == (hdr->ahead_stamp | kAheadDone);
    const TileAgg wa = counted ? reinterpret_cast<const AheadEntry*>(hdr->ahead_tab)[tile].w[wv] : dense_aggregate(src, wseg, n, before, before2, after, chunk0, lane);
    if (lane == 0) l.wagg[wv] = wa;
    __syncthreads();
    TileAgg before_me = agg_identity(), tagg = agg_identity();
#pragma unroll
    for (int w = 0; w < k4Waves; ++w) { if (w < wv) before_me = combine(before_me, l.wagg[w]); tagg = combine(tagg, l.wagg[w]); }
    if (wv == 0) {
        Prefix ex;
        uint32_t it, stl;
        HBS4_TL(tile, 1, 1)
        const bool ok = look_back4(desc, tile, tagg, hdr, lane, ex, it, stl);
        HBS4_TL(tile, 2, 0)
        HBS4_PRIO(0);
        const uint32_t tile_kept = tagg.known + (ex.inside ? tagg.sig : 0u);
        const bool can = rbsp != nullptr && ex.kept + tile_kept <= rbsp_cap;
        if (lane == 0) {
            l.ex_kept = ex.kept; l.ex_nals = ex.nals; l.ex_inside = ex.inside;
            l.ex_ok = !ok ? 0u : (rbsp != nullptr && !can) ? 2u : 1u;
            if (ok && rbsp != nullptr && !can) atomicMax(&hdr->error, (uint32_t)(-HBS_E_CAPACITY));
            if (ok && last_tile) {
                const Prefix incl = fold(ex, tagg);
                hdr->final_kept = incl.kept; hdr->final_nals = incl.nals; hdr->final_inside = incl.inside;
            }
        }
    } else {
        HBS4_PRIO(0);
    }
    __syncthreads();
    if (l.ex_ok == 0u) return false;
    Prefix excl;
    {
        Prefix ex;
        ex.kept = l.ex_kept; ex.nals = l.ex_nals; ex.inside = l.ex_inside;
        excl = prefix_uniform4(ex);
    }
    dense_emit(src, wseg, n, before, before2, after, chunk0, lane, before_me, excl, rbsp != nullptr && l.ex_ok == 1u, rbsp + excl.kept, tgt,
               &l.dep[wv][lane & (kDepCap - 1)].xpp);
    /* The next ticket only when EVERY wavefront is through with its rows (round 5).  Where a stretch of padding begins inside a
     * tile, wavefront 0's rows are ordinary and the others' are not: it came here 35 us before them, took a ticket, and sat on
     * it at the barrier below -- and every tile behind that ticket waited in its look-back for a tile nobody had started.  That,
     * once per stretch, was the bench's mixed stream (1 % of it in 640 KiB stretches): 1.32-1.34 x the uniform time, 2 ms of 8. */
    __syncthreads();
    if (tid == 0) l.ticket = ticket_base + atomicAdd(&hdr->ticket, 1u);
    __syncthreads();
    HBS4_TL(tile, 3, 0)
    return true;
}

} // anonymous namespace

#if HBS4_MAIN
/* the tiles the prologue marked, a workgroup each: a rough count, then the first half of dense_tile, into the table */
__global__ __launch_bounds__(k4Threads)
void k_scan_ahead4(const uint8_t* __restrict__ stream, uint64_t n, uint64_t num_tiles, unsigned long long* __restrict__ cand,
                   const uint32_t* __restrict__ list, const AheadCtl* __restrict__ ctl,
                   AheadEntry* __restrict__ tab, const RunHeader* __restrict__ hdr, int gate)
{
    const uint32_t marked = ctl->listed;
    const unsigned long long stamp = ahead_stamp_of(ctl->call);
    if (blockIdx.x >= marked || gate_closed(gate, hdr)) return;
    __shared__ uint32_t rough[k4Waves];
    __shared__ TileAgg wagg[k4Waves];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (uint32_t i = blockIdx.x; i < marked; i += gridDim.x) {
        const uint64_t tile = list[i];
        /* (the stream's last tile is read from a padded copy by the main kernel: never counted ahead) */
        if (tile + 1 >= num_tiles) continue;
        const uint64_t base = tile * (uint64_t)k4TileBytes;
        const uint64_t wseg = base + (uint64_t)(wv * k4WaveBytes);
        /* roughly: chunks that hold a zero pair by themselves (the tile is whole: not the stream's last) */
        uint32_t mine = 0;
#pragma unroll 4
        for (int r = 0; r < k4Rows; ++r) {
            const Quad q = *reinterpret_cast<const Quad*>(stream + wseg + (uint64_t)r * k4RowBytes + (uint64_t)lane * 16u);
            mine += (uint32_t)__builtin_popcountll(__ballot(chunk_flag(0xFFFFFFFFu, q.x, q.y, q.z, q.w, 0xFFFFFFFFu)));
        }
        __syncthreads();                                 /* the previous tile is done with rough / wagg */
        if (lane == 0) rough[wv] = mine;
        __syncthreads();
        const bool walk = rough[0] + rough[1] + rough[2] + rough[3] >= kAheadRoughMin;
        if (!walk) continue;
        const uint32_t before = (wseg >= 4) ? stream_load4(stream + wseg - 4) : 0xFFFFFFFFu;
        const uint32_t before2 = (wseg >= 8) ? stream_load4(stream + wseg - 8) : 0xFFFFFFFFu;
        const uint32_t after = (wv != k4Waves - 1) ? stream_load4(stream + wseg + k4WaveBytes) : load_dword_guarded(stream, (int64_t)(wseg + k4WaveBytes), n);
        const TileAgg wa = dense_aggregate(stream, wseg, n, before, before2, after, (uint32_t)(64 * k4Rows * wv), lane);
        if (lane == 0) wagg[wv] = wa;
        __syncthreads();
        if (tid == 0) {
            AheadEntry e;
#pragma unroll
            for (int w = 0; w < k4Waves; ++w) e.w[w] = wagg[w];
            tab[tile] = e;
            cand[tile] = stamp | kAheadDone;
        }
    }
}

uint64_t scan4_ahead_entry_bytes() { return sizeof(AheadEntry) + sizeof(unsigned long long) + sizeof(uint32_t); }     /* table entry, the tile's word, list word */
bool scan4_counts_ahead(uint64_t n) { return n >= kAheadMinBytes; }

void launch_scan_ahead4(const ScanArgs& a, uint64_t num_tiles, int gate, hipStream_t st)
{
    if (!a.ahead_cand || !a.ahead_tab || num_tiles < 2) return;
    /* four workgroups a CU (119 registers): a stretch's tiles at the same time; with nothing marked they all leave at once */
    k_scan_ahead4<<<dim3(1024), dim3(k4Threads), 0, st>>>(a.stream, a.n, num_tiles, a.ahead_cand, a.ahead_list, a.ahead_ctl,
                                                         static_cast<AheadEntry*>(a.ahead_tab), a.hdr, gate);
}

#endif // HBS4_MAIN

/* tail: the 0xFF-padded copy of the stream from `tail_base` - k4TailLead on (k_scan_prologue: the last 192 KiB tile of the stream,
 * which holds the last tile of either geometry); the stream's last tile is read from it, unguarded */
__global__ __launch_bounds__(k4Threads, HBS4_WG_PER_CU)
void HBS4_KERNEL(const uint8_t* __restrict__ stream, uint64_t n, uint64_t num_tiles,
                 hbs_nal_entry* __restrict__ index, uint64_t index_cap,
                 uint8_t* __restrict__ rbsp, uint64_t rbsp_cap,
                 unsigned long long* __restrict__ desc, RunHeader* __restrict__ hdr, const uint8_t* __restrict__ tail, uint64_t tail_base,
                 int gate, int first_static)
{
    if (gate_closed(gate, hdr)) return;
    __shared__ Lds4 l;
    const int tid0 = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    EmitTarget tgt;
    tgt.index = index; tgt.index_cap = index_cap; tgt.hdr = hdr;
    /* Every tile by ticket (the default): tiles are looked back in ticket order, so a workgroup only ever waits for tiles that a
     * RUNNING workgroup has claimed -- no assumption about which workgroups are resident, whatever else runs on the device.
     * `first_static` (hbs_ctx_set_device_exclusive: the caller says this context has the device to itself): the first tile is the
     * workgroup's number and the others are gridDim.x + ticket.  512 workgroups asking ONE address for a ticket in the kernel's
     * first microsecond are served one after the other -- ~1 % of a 1 GiB call -- but static first tiles are safe only while
     * every workgroup of the grid is resident at once: two persistent scans on one device (two contexts, two processes) could
     * each hold the slots the other's low-numbered workgroups need, and wait for each other until the look-back's guard fires
     * (round 5's advice). */
    const uint32_t ticket_base = first_static ? gridDim.x : 0u;
    if (tid0 == 0) l.ticket = first_static ? blockIdx.x : atomicAdd(&hdr->ticket, 1u);
    __syncthreads();
    HBS4_T_DECL

    uint32_t d_before = 0, d_before2 = 0, d_after = 0;
    uint64_t d_tile = 0;
    for (;;) {
    int pending = 0;                   /* 1: a dense tile -- handled below the tile loop, where no row is live */
    for (;;) {
        int tid = launder_lane(tid0);
        int lane = tid & 63;
        const uint64_t tile = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)l.ticket);
        if (tile >= num_tiles) break;
        HBS4_TL(tile, 0, 0)
        const uint64_t base = tile * (uint64_t)k4TileBytes;
        const uint64_t tile_end = base + (uint64_t)k4TileBytes;
        const uint64_t wseg = base + (uint64_t)(wv * k4WaveBytes);
        const bool last_tile = tile == num_tiles - 1;
        const bool edge_tile = tile_end + 4 > n;          /* some chunk of the tile may be cut by the stream end */

        /* the last tile comes from the padded copy: tail[kTailLead + i] = stream[tail_base + i] */
        const uint8_t* const src = last_tile
            ? reinterpret_cast<const uint8_t*>(reinterpret_cast<uintptr_t>(tail) + (uintptr_t)k4TailLead - (uintptr_t)tail_base) : stream;
        /* until the tile's aggregate is out, this workgroup is what its successors wait for */
        HBS4_PRIO(3);
        RowRegs R;
        /* The dwords around the segment first: the first group of the flag pass needs them, and a load issued behind the rows
         * would make it wait for all of them.  The rows themselves are fetched INSIDE the flag pass, four at a time. */
        R.before = (wseg >= 4) ? stream_load4(src + wseg - 4) : 0xFFFFFFFFu;
        R.before2 = (wseg >= 8) ? stream_load4(src + wseg - 8) : 0xFFFFFFFFu;
        R.after = (last_tile || wv != k4Waves - 1) ? stream_load4(src + wseg + k4WaveBytes)
                                                   : load_dword_guarded(stream, (int64_t)(wseg + k4WaveBytes), n);
        const u32x4* const rowp = reinterpret_cast<const u32x4*>(src + wseg) + lane;
#if !HBS4_PROGRESSIVE
        fetch_row_regs(R, src, wseg, lane);
#endif
        HBS4_T_MARK(0)

        /* ---- 1. flag masks of my rows --------------------------------------------------- */
        /* Straight-line over the named rows; a row's 64-bit mask is stashed in lane r of
         * fm_lo/fm_hi (v_writelane), so nothing per-row lives in SGPRs or LDS. */
        uint32_t fm_lo = 0, fm_hi = 0;     /* lane r: flag mask of my row r (rows without elements: 0) */
        uint32_t wslot = 0;                /* elements of this wavefront so far      */
        {
            /* Four rows at a time: the test of a row is a chain of dependent steps (neighbour dwords through DPP, nine packed
             * minima, compare, ballot), and a branch per row keeps the compiler from overlapping the chains of different rows.
             * So a group's four ballots are formed without a branch, and only a group in which some chunk is flagged -- one in
             * fifteen in coded video -- goes through the per-row bookkeeping. */
#define HBS_FLAG_EVAL(r, e_prev_w, e_next_x) \
                const uint32_t xp##r = from_prev_lane(R.q##r.w, (e_prev_w)); \
                const uint32_t xn##r = from_next_lane(R.q##r.x, (e_next_x)); \
                const bool f##r = chunk_flag(xp##r, R.q##r.x, R.q##r.y, R.q##r.z, R.q##r.w, xn##r); \
                const uint64_t fmask##r = __ballot(f##r);
#define HBS_FLAG_KEEP(r, e_prev_z) \
                if (fmask##r != 0) {     /* a zero pair somewhere: now the exact question -- does a pattern 00 00 {<=3} end in bytes [0, 18)? */ \
                    bool g##r = f##r; \
                    uint64_t gmask##r = fmask##r; \
                    if (HBS4_EXACT_FLAG && __builtin_popcountll(fmask##r) > HBS4_EXACT_MIN) {   /* one or two: a start code, most likely -- nothing to gain */ \
                        g##r = f##r && chunk_pattern_any_dev(xp##r, R.q##r.x, R.q##r.y, R.q##r.z, R.q##r.w, xn##r); \
                        gmask##r = __ballot(g##r); \
                    } \
                    if (gmask##r != 0) { /* stash the mask, leave the chunk's surroundings for its element thread */ \
                    write_lane(fm_lo, (uint32_t)gmask##r, r); \
                    write_lane(fm_hi, (uint32_t)(gmask##r >> 32), r); \
                    const uint32_t xpp = from_prev_lane(R.q##r.z, (e_prev_z)); \
                    const uint32_t slot = wslot + lanes_below(gmask##r); \
                    if (g##r && slot < kDenseElems) { \
                        const uint32_t ch = (uint32_t)(64 * (k4Rows * wv + r) + lane); \
                        l.wlist[wv][slot] = (uint16_t)ch; \
                        if (slot < (uint32_t)kDepCap) { \
                            Deposit d; \
                            d.xpp = xpp; d.xp = xp##r; d.x0 = R.q##r.x; d.x1 = R.q##r.y; d.x2 = R.q##r.z; d.x3 = R.q##r.w; d.xn = xn##r; \
                            d.chunk = ch; \
                            l.dep[wv][slot] = d; \
                        } \
                    } \
                    wslot += (uint32_t)__builtin_popcountll(gmask##r); \
                } }
#define HBS_FLAG_GROUP(a, wa, xa, za, b, wb, xb, zb, c, wc, xc, zc, d, wd, xd, zd) { \
                HBS_FLAG_EVAL(a, wa, xa) HBS_FLAG_EVAL(b, wb, xb) HBS_FLAG_EVAL(c, wc, xc) HBS_FLAG_EVAL(d, wd, xd) \
                if ((fmask##a | fmask##b | fmask##c | fmask##d) != 0) { \
                    HBS_FLAG_KEEP(a, za) HBS_FLAG_KEEP(b, zb) HBS_FLAG_KEEP(c, zc) HBS_FLAG_KEEP(d, zd) } \
                HBS_FLAG_FENCE }
            /* The fetch runs inside the flag pass: HBS_LD4 issues four rows, and a group is flagged when the rows up to its fourth
             * neighbour's have been issued -- loads return in order, so the group waits for `vmcnt(3)`, not for the tile, and a
             * wavefront never has more than 7 row loads in flight.  Two things come of it.  The flag pass hides under the fetch.
             * And the CU's memory queue stays short: with 48 loads per wavefront issued at once (and 48 stores in the copy), the
             * look-back polls of the OTHER workgroup on this CU waited behind them, 2.7 us a poll; a model of this kernel
             * (scripts/ubench/ceiling3.hip, profiles/r03/ceiling3_*.txt) moves 5.9 TB/s with both throttled and 5.1-5.3 without.
             * The sched_barriers keep the compiler from hoisting the loads back to the top. */
#if HBS4_PROGRESSIVE
#define HBS_LD4(a, b, c, d) { R.q##a = stream_load16(rowp + a * 64); R.q##b = stream_load16(rowp + b * 64); \
                R.q##c = stream_load16(rowp + c * 64); R.q##d = stream_load16(rowp + d * 64); __builtin_amdgcn_sched_barrier(0); }
#define HBS_FLAG_FENCE __builtin_amdgcn_sched_barrier(0);
#else
#define HBS_LD4(a, b, c, d)
#define HBS_FLAG_FENCE
#endif
            /* the rows' loads and flag groups in order: hbs_scan4_rows*.h, written by scripts/set_rows4.py */
            HBS_FLAG_PASS
#undef HBS_FLAG_GROUP
#undef HBS_LD4
#undef HBS_FLAG_FENCE
#undef HBS_FLAG_KEEP
#undef HBS_FLAG_EVAL
        }
        if (edge_tile && (n & 15ull) != 0 && n > wseg && n < wseg + (uint64_t)k4WaveBytes) {
            /* the chunk cut by the stream end is always an element */
            const uint32_t cut = (uint32_t)(n - wseg) >> 4;            /* its chunk number in my segment */
            const int cr = (int)(cut >> 6), cl = (int)(cut & 63u);
            const uint64_t have = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)fm_hi, cr) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)fm_lo, cr);
            if (!((have >> cl) & 1ull)) {
                /* not flagged by its bytes: nothing was deposited for it; it is the wavefront's last
                 * element, and its thread must read the stream itself */
                if (lane == 0 && wslot < (uint32_t)kDepCap) l.dep[wv][wslot].chunk = 0xFFFFFFFFu;
                if (lane == 0 && wslot < kDenseElems) l.wlist[wv][wslot] = (uint16_t)(64 * k4Rows * wv + (int)cut);
                if (lane == cr) { if (cl < 32) fm_lo |= 1u << cl; else fm_hi |= 1u << (cl - 32); }
            }
        }
        /* lane r: elements of my rows in front of row r; then the same across wavefronts */
        uint32_t local_pre;
        uint64_t rowmask;                  /* my rows that hold an element */
        {
            const uint32_t cnt = (lane < k4Rows) ? (uint32_t)__builtin_popcount(fm_lo) + (uint32_t)__builtin_popcount(fm_hi) : 0u;
            const uint32_t inc = wave_incl_scan32(cnt, lane);
            local_pre = inc - cnt;
            rowmask = __ballot(cnt != 0u);
            if (lane == 63) l.wave_tot[wv] = inc;
        }
        __syncthreads();
        tid = launder_lane(tid0); lane = tid & 63;
        const uint32_t wt0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)l.wave_tot[0]);
        const uint32_t wt1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)l.wave_tot[1]);
        const uint32_t wt2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)l.wave_tot[2]);
        const uint32_t wt3 = (uint32_t)__builtin_amdgcn_readfirstlane((int)l.wave_tot[3]);
        static_assert(k4Waves == 4, "four wavefront totals");
        const uint32_t wb1 = wt0, wb2 = wt0 + wt1, wb3 = wb2 + wt2, nflag = wb3 + wt3;
        const uint32_t wave_base = (wv == 0) ? 0u : (wv == 1) ? wb1 : (wv == 2) ? wb2 : wb3;
        if (nflag > kDenseElems) {
            /* ---- dense tile: every chunk an element, every wavefront its own rows (dense_tile).  The call is made OUTSIDE the
             * tile loop, where nothing of a tile is live: inside it, what has to survive the call is spilled on the common path. */
            d_before = R.before; d_before2 = R.before2; d_after = R.after; d_tile = tile;
            pending = 1;
            break;
        }
        /* readlanes: only in wave-uniform control flow */
#define HBS_ROW_PRE(r) (wave_base + (uint32_t)__builtin_amdgcn_readlane((int)local_pre, (r)))
#define HBS_ROW_FM(r) (((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)fm_hi, (r)) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)fm_lo, (r)))
        HBS4_DBG(if (tile == 0) { if (tid == 0) { g_dbg4[0] = nflag; } for (int i = tid; i < (int)kDenseElems; i += k4Threads) g_dbg4[256 + i] = list_at(l, (uint32_t)i, wb1, wb2, wb3); })
        HBS4_T_MARK(1)

        /* ---- 2..4 on wavefront 0: elements -> tile aggregate -> look-back -> emit ----------- */
        const uint32_t npass = (nflag + (uint32_t)k4ElemPass - 1u) / (uint32_t)k4ElemPass;
        /* Several batches (a tile of small NALs, a stretch of zero pairs): wavefront 1 parks rows too and takes every other
         * batch, in both halves; what a batch found out about its elements waits in LDS (rec) between the halves, the batch
         * aggregates meet in LDS (bagg), and all segment words are out before anybody copies.  (Until round 3 wavefront 0 walked
         * all batches alone, ~10 k cycles each, the other three waiting; all FOUR on the elements would need 80 KiB of parked
         * rows, or the rows read again: tried as a function of its own, it lost to this below five batches and gained 4 % above.) */
        const bool multi = npass > 1u;
        const bool elem_wave = wv == 0 || (multi && wv < HBS4_ELEM_WAVES);
        if (elem_wave) {
            /* this code needs ~100 registers of its own: part of this wavefront's rows wait in LDS */
            Elem el;
            TileAgg acc = agg_identity(), e = agg_identity();
            Prefix ex;
            bool ok = true, can = false;
#define HBS_PARK(i, r) l.park[wv][i][lane] = R.q##r;
            HBS_PARKED(HBS_PARK)
#undef HBS_PARK
            el.gap = 0; el.chunk = 0;
            if (!multi) {
                TileAgg ea = agg_identity();
                if ((uint32_t)lane < nflag) ea = elem_make(el, l, (uint32_t)lane, wb1, wb2, wb3, src, base, n, last_tile);
                ea = wave_scan_combine(ea, lane);
                e = agg_prev_lane(ea);
                if (lane == 0) e = agg_identity();
                acc = agg_readlane(ea, 63);
            } else {
#pragma unroll 1
                for (uint32_t p = (uint32_t)wv; p < npass; p += (uint32_t)HBS4_ELEM_WAVES) {
                    const uint32_t i = p * (uint32_t)k4ElemPass + (uint32_t)lane;
                    TileAgg ea = agg_identity();
                    if (i < nflag) {
                        ea = elem_make(el, l, i, wb1, wb2, wb3, src, base, n, last_tile);
                        rec4_store(l.rec[i], el);
                    }
                    ea = wave_scan_combine(ea, lane);
                    if (lane == 63) l.bagg[p] = ea;
                }
            }
            if (multi) __syncthreads();                    /* the other wavefronts: below */
            if (wv == 0) {
            if (multi) {
#pragma unroll 1
                for (uint32_t p = 0; p < npass; ++p) acc = combine(acc, l.bagg[p]);
            }
            const uint64_t last_end = (nflag > 0) ? base + 16ull * (list_at(l, nflag - 1u, wb1, wb2, wb3) + 1u) : base;
            const TileAgg tagg = combine(acc, gap_agg(span_bytes(last_end, tile_end, n)));
            HBS4_T_MARK(2)
            HBS4_TL(tile, 1, 0)

            uint32_t it, stl;
            HBS4_DBG(if (g_fake_lb4) { ok = true; it = 0; stl = 0; ex.kept = tile * (uint64_t)(k4TileBytes - 4096); ex.nals = tile * 16; ex.inside = 1; } else)
            ok = look_back4(desc, tile, tagg, hdr, lane, ex, it, stl);
            HBS4_T_COUNT(7, ((unsigned long long)stl << 32) | it)
            HBS4_PRIO(0);
            const uint32_t tile_kept = tagg.known + (ex.inside ? tagg.sig : 0u);
            can = rbsp != nullptr && ex.kept + tile_kept <= rbsp_cap;
            if (lane == 0) {
                l.ex_kept = ex.kept; l.ex_nals = ex.nals; l.ex_inside = ex.inside;
                l.ex_ok = !ok ? 0u : (rbsp != nullptr && !can) ? 2u : 1u;
                l.seg[0] = seg_pack(-1, 0u, ex.inside != 0u);
                if (ok && rbsp != nullptr && !can) atomicMax(&hdr->error, (uint32_t)(-HBS_E_CAPACITY));
                if (ok && last_tile) {
                    const Prefix incl = fold(ex, tagg);
                    hdr->final_kept = incl.kept; hdr->final_nals = incl.nals; hdr->final_inside = incl.inside;
                }
            }
            HBS4_T_MARK(3)
            HBS4_TL(tile, 2, 0)
            } else {
                HBS4_PRIO(0);
            }
            if (multi) __syncthreads();
            if (!multi) {
                /* one batch (nearly always): the elements are still in registers */
                if (ok && (uint32_t)lane < nflag) elem_emit(el, e, ex, can, rbsp + ex.kept, tgt, &l.seg[lane + 1]);
            } else if (l.ex_ok != 0u) {
                emit_batches(l, (uint32_t)wv, npass, nflag, lane, src, base, n, rbsp, tgt);
            }
#define HBS_UNPARK(i, r) R.q##r = l.park[wv][i][lane];
            HBS_PARKED(HBS_UNPARK)
#undef HBS_UNPARK
        } else {
            /* (a barrier counts wavefronts, wherever they are in the code: these two meet the two above) */
            HBS4_PRIO(0);
            if (multi) {
                __syncthreads(); __syncthreads();
#if HBS4_EMIT_WAVES > HBS4_ELEM_WAVES
                /* the second half needs fewer registers than the first (no window rules: what they found is in LDS): these
                 * wavefronts take their share of it with all their rows in place */
                if (l.ex_ok != 0u) emit_batches(l, (uint32_t)wv, npass, nflag, lane, src, base, n, rbsp, tgt);
#endif
            }
        }
        __syncthreads();
        if (l.ex_ok == 0u) return;
        Prefix excl;
        {
            Prefix ex;
            ex.kept = l.ex_kept; ex.nals = l.ex_nals; ex.inside = l.ex_inside;
            excl = prefix_uniform4(ex);
        }
        const bool can_store = rbsp != nullptr && l.ex_ok == 1u;
        uint8_t* const out = rbsp + excl.kept;
        HBS4_T_MARK(4)

        /* ---- 5. copy everything that is not an element ------------------------------------------------------------- */
        tid = launder_lane(tid0); lane = tid & 63;
        if (can_store) {
            const uint32_t whole = (uint32_t)(span_bytes(base, tile_end, n) >> 4);   /* chunks of the tile that are complete */
            const uint32_t cc0 = (uint32_t)(64 * k4Rows * wv + lane);
            /* lane j: segment word j (j = 0..63), word 64 apart: a row without elements needs ONE word (that of the last element
             * in front of it), picked with a readlane instead of an LDS round trip; words past 64 (tiles with several batches of
             * elements) come from LDS, one broadcast read per row */
            const uint32_t segv = l.seg[lane];
            const uint32_t seg64 = (uint32_t)__builtin_amdgcn_readfirstlane((int)l.seg[k4ElemPass]);
            /* At most HBS4_COPY_DEPTH stores of a wavefront in flight (see the flag pass: a short memory queue on the CU is what
             * lets the other workgroup's look-back through; depth 3 is the model's optimum, 5 and more lose all of it). */
#if HBS4_COPY_DEPTH >= 0
#define HBS_COPY_THROTTLE asm volatile("s_waitcnt vmcnt(%0)" :: "n"(HBS4_COPY_DEPTH) : "memory");
#else
#define HBS_COPY_THROTTLE
#endif
            /* Straight-line over the named rows.  A chunk with k elements in front of it goes where segment word k says. */
#define HBS_COPY(r) { \
                const uint32_t cc = cc0 + 64u * r; \
                if (!((rowmask >> r) & 1ull)) {          /* no element in this row: one k, one word for all lanes */ \
                    const uint32_t k = HBS_ROW_PRE(r); \
                    const uint32_t w = (k < (uint32_t)k4ElemPass) ? (uint32_t)__builtin_amdgcn_readlane((int)segv, (int)(k & 63u)) \
                                     : (k == (uint32_t)k4ElemPass) ? seg64 : (uint32_t)__builtin_amdgcn_readfirstlane((int)l.seg[k]); \
                    if (seg_inside(w) && cc < whole) \
                        arena_store16(out + (int64_t)seg_bias(w) + 16u * cc, R.q##r); \
                } else { \
                    const uint64_t f = HBS_ROW_FM(r); \
                    const uint32_t k = HBS_ROW_PRE(r) + lanes_below(f); \
                    if (!((f >> lane) & 1ull) && cc < whole) { \
                        const uint32_t w = l.seg[k]; \
                        if (seg_inside(w)) arena_store16(out + (int64_t)(seg_bias(w) + (int32_t)(16u * cc)), R.q##r); \
                    } \
                } \
                HBS_COPY_THROTTLE }
            HBS_ROWS(HBS_COPY)
#undef HBS_COPY
#undef HBS_COPY_THROTTLE
        }
        /* The next tile is claimed only now: tiles are looked back in ticket order, and a ticket
         * taken before the copy (whose duration varies with memory load) makes successors wait for
         * a tile that has not even been started (measured: 2.1 instead of 3.6 polls per tile).  Tiles dealt
         * out in stripes instead (tile = workgroup + k x grid, no atomic, no drain of this wavefront's stores
         * in front of it) ran 8.6 ms against 7.06 on the 16 GiB bench stream: workgroups do not progress
         * evenly, and with stripes the fast ones wait in their look-backs for the slow ones. */
#if HBS4_TICKET_BARRIER
        __syncthreads();
#endif
        if (tid == 0) l.ticket = ticket_base + atomicAdd(&hdr->ticket, 1u);
        __syncthreads();
        HBS4_T_MARK(5)
        HBS4_TL(tile, 3, 0)
#undef HBS_ROW_PRE
#undef HBS_ROW_FM
    }
    if (pending == 0) break;
    {
        const uint64_t base = d_tile * (uint64_t)k4TileBytes;
        const bool last_tile = d_tile == num_tiles - 1;
        const uint8_t* const src = last_tile
            ? reinterpret_cast<const uint8_t*>(reinterpret_cast<uintptr_t>(tail) + (uintptr_t)k4TailLead - (uintptr_t)tail_base) : stream;
        if (!dense_tile(l, src, base + (uint64_t)(wv * k4WaveBytes), n, d_before, d_before2, d_after, d_tile, last_tile, rbsp, rbsp_cap, desc, hdr, index, index_cap, ticket_base)) return;
    }
    }
    HBS4_T_FLUSH
}

#if HBS4_MAIN
#ifdef HBS_PHASE_TIMING
extern "C" int hbs_debug_fake_lb4(int on)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_fake_lb4), &on, sizeof(int));
}
extern "C" int hbs_debug_dump4(uint32_t* host_out /* [4096] */)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dbg4), sizeof(uint32_t) * 4096);
}
extern "C" int hbs_debug_timeline4(unsigned long long* host_out /* [tiles][4] */, unsigned tiles)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_tl4), sizeof(unsigned long long) * 4 * (tiles < kTl4Tiles ? tiles : kTl4Tiles));
}
extern "C" int hbs_debug_timeline_who4(unsigned long long* host_out /* [tiles] */, unsigned tiles)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_tlwho4), sizeof(unsigned long long) * (tiles < kTl4Tiles ? tiles : kTl4Tiles));
}
extern "C" int hbs_debug_phase_cycles4(unsigned long long* host_out /* [1024][8] */)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_phase_cycles4), sizeof(unsigned long long) * 1024 * 8);
}
#endif

int scan4_tile_bytes() { return k4TileBytes; }
int scan4_tail_bytes() { return k4TailBytes; }

/* ---- one launch in front of the main kernel --------------------------------------------------------
 * Everything a call needs before its tiles: the run header, the density probe of the automatic mode, the
 * 0xFF-padded copy of the stream's last tile (the main kernel reads that tile from it, unguarded), and the
 * cleared index and look-back words -- five launches' worth of work whose latency mattered for streams of
 * tens of MiB.  Workgroup roles by number: [0, 64) probe windows (and workgroup 0 the header),
 * [64, 64 + kTailBlocks) the tail copy, the rest clears. */
constexpr int kProbeBlocks = 64;
constexpr int kTailBlocks = 32;
constexpr int kClearBlocksMin = 32, kClearBlocksMax = 4096;      /* sized by the words to clear: ~16 stores per thread */

__global__ __launch_bounds__(256)
void k_scan_prologue(const uint8_t* __restrict__ stream, uint64_t n, RunHeader* __restrict__ hdr, uint8_t* __restrict__ tail,
                     unsigned long long* __restrict__ index_words, uint64_t n_index_words,
                     unsigned long long* __restrict__ desc, uint64_t n_desc_words, int do_probe, int tail_tile_bytes,
                     unsigned long long* __restrict__ ahead_cand, void* ahead_tab, uint32_t* __restrict__ ahead_list, AheadCtl* __restrict__ ahead_ctl,
                     int sample_blocks)
{
    if ((int)blockIdx.x >= (int)gridDim.x - sample_blocks) {
        /* the sample of the count-ahead (see dense_tile), with the exact question the flag pass asks (neighbouring chunks ignored).
         * A wavefront takes sixteen tiles at a time: lanes 0-47 a chunk each of the first look */
        const int lane = threadIdx.x & 63;
        const unsigned long long ahead_stamp = ahead_stamp_of(ahead_ctl->call);       /* (left by the finish launch of the call before, or the allocation) */
        const uint64_t wave = ((uint64_t)(blockIdx.x - ((int)gridDim.x - sample_blocks)) * blockDim.x + threadIdx.x) >> 6;
        const uint64_t nwaves = ((uint64_t)sample_blocks * blockDim.x) >> 6;
        const uint64_t tiles = (n + (uint64_t)k4TileBytes - 1) / (uint64_t)k4TileBytes;
        static_assert(kAheadSample * 4096 == k4TileBytes && kAheadCoarse * 65536 == k4TileBytes, "a sample every 4 KiB, a coarse one every 64 KiB");
        auto hit = [&](uint64_t off) -> bool {
            if (off + 16 > n) return false;
            const Quad q = *reinterpret_cast<const Quad*>(stream + off);
            return chunk_flag(0xFFFFFFFFu, q.x, q.y, q.z, q.w, 0xFFFFFFFFu) && chunk_pattern_any_dev(0xFFFFFFFFu, q.x, q.y, q.z, q.w, 0xFFFFFFFFu);
        };
        /* (its sixteen tiles are nwaves apart: the tiles of one stretch are looked at by different wavefronts) */
        for (uint64_t t0 = wave; t0 < tiles; t0 += nwaves * 16u) {
            const uint64_t ct = t0 + (uint64_t)(lane / kAheadCoarse) * nwaves;
            const bool ch = lane < 16 * kAheadCoarse && ct < tiles &&
                            hit(ct * (uint64_t)k4TileBytes + (uint64_t)(lane % kAheadCoarse) * 65536u + 32768u + 2048u);      /* (one of the full look's places) */
            unsigned long long coarse = __ballot(ch);
            while (coarse != 0ull) {                                 /* (coded video: not once) */
                const int slot = __builtin_ctzll(coarse) / kAheadCoarse;
                const uint64_t t = t0 + (uint64_t)slot * nwaves;
                coarse &= ~(7ull << (kAheadCoarse * slot));
                static_assert(kAheadCoarse == 3, "three bits a tile");
                for (int dt = -1; dt <= 1; ++dt) {
                    const uint64_t u = t + (uint64_t)(int64_t)dt;
                    if ((dt < 0 && t == 0) || u >= tiles) continue;
                    /* (looked at already, as the neighbour of its neighbour?  stamp | 0: looked at and not marked) */
                    if (__hip_atomic_load(&ahead_cand[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= ahead_stamp) continue;
                    const unsigned long long hits = __ballot(lane < kAheadSample && hit(u * (uint64_t)k4TileBytes + (uint64_t)lane * 4096u + 2048u));
                    /* a sixth of the tile, or 12 KiB in one piece, or 8 KiB at the edge a stretch comes in by */
                    const bool mark = __builtin_popcountll(hits) >= kAheadMinHits || (hits & (hits >> 1) & (hits >> 2)) != 0ull ||
                                      (hits & 3ull) == 3ull || ((hits >> (kAheadSample - 2)) & 3ull) == 3ull;
                    if (lane == 0) {
                        const unsigned long long word = ahead_stamp | (mark ? kAheadMarked : 0ull);
                        if (atomicMax(&ahead_cand[u], word) < word && mark) ahead_list[atomicAdd(&ahead_ctl->listed, 1u)] = (uint32_t)u;
                    }
                }
            }
        }
        return;
    }
    const int b = blockIdx.x;
    if (b < kProbeBlocks) {
        if (b == 0 && threadIdx.x == 0) {
            hdr->final_kept = 0; hdr->final_nals = 0; hdr->final_inside = 0;
            hdr->error = 0; hdr->first_empty = ~0ull; hdr->abort_flag = 0; hdr->ticket = 0;
            hdr->probe_chunks = 0; hdr->probe_flagged = 0; hdr->rewalk_count = 0;
            hdr->pad_a = 0; hdr->ahead_stamp = ahead_ctl ? ahead_stamp_of(ahead_ctl->call) : 0ull;
            hdr->ahead_cand = ahead_tab ? reinterpret_cast<unsigned long long>(ahead_cand) : 0ull;
            hdr->ahead_tab = ahead_cand ? reinterpret_cast<unsigned long long>(ahead_tab) : 0ull;
        }
        /* Density probe: kProbeBlocks windows of 16 KiB spread evenly over the stream; counts the chunks that
         * chunk_flag() would hand to the element path (neighbouring chunks ignored: an estimate is all the
         * choice needs).  Every probe workgroup writes its slot, so nothing has to be zeroed beforehand. */
        uint32_t chunks = 0, flagged = 0;
        if (do_probe) {
            const uint64_t stride = (n / kProbeBlocks) & ~15ull;
            const uint64_t base = (uint64_t)b * stride;
            for (int k = 0; k < 4; ++k) {
                const uint64_t off = base + (uint64_t)(k * 256 + (int)threadIdx.x) * 16u;
                const bool in = off + 16 <= n;
                bool f = false;
                if (in) {
                    /* with the dwords around the chunk, as the kernels see it (round 3: a start code across two chunks makes two
                     * elements, and on streams of 384-byte NALs a probe blind to that was 25 % low -- on the wrong side of the rule) */
                    const Quad q = *reinterpret_cast<const Quad*>(stream + off);
                    const uint32_t xp = off >= 4 ? *reinterpret_cast<const uint32_t*>(stream + off - 4) : 0xFFFFFFFFu;
                    const uint32_t xn = off + 20 <= n ? *reinterpret_cast<const uint32_t*>(stream + off + 16) : 0xFFFFFFFFu;
                    f = chunk_flag(xp, q.x, q.y, q.z, q.w, xn) && (!HBS4_EXACT_FLAG || chunk_pattern_any_dev(xp, q.x, q.y, q.z, q.w, xn));
                }
                chunks += (uint32_t)__builtin_popcountll(__ballot(in));
                flagged += (uint32_t)__builtin_popcountll(__ballot(f));
            }
        }
        __shared__ uint32_t part[4][2];
        if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6][0] = chunks; part[threadIdx.x >> 6][1] = flagged; }
        __syncthreads();
        if (threadIdx.x == 0) {
            hdr->probe_slot[b][0] = part[0][0] + part[1][0] + part[2][0] + part[3][0];
            hdr->probe_slot[b][1] = part[0][1] + part[1][1] + part[2][1] + part[3][1];
        }
    } else if (b < kProbeBlocks + kTailBlocks) {
        /* tail[k4TailLead + i] = stream[last_base + i] for i in [-k4TailLead, tile + pad), 0xFF where the stream has no byte;
         * the tile size is that of the kernel which will read the copy */
        if (tail_tile_bytes == 0 || n == 0) return;
        const uint64_t last_base = ((n - 1) / (uint64_t)tail_tile_bytes) * (uint64_t)tail_tile_bytes;
        const uint32_t tail_bytes = (uint32_t)(k4TailLead + tail_tile_bytes + 64);
        /* 16 bytes per thread and step (round 4; byte by byte this copy was the longest thing in the launch): last_base and the lead
         * are multiples of 16, so chunk c of the copy is an aligned chunk of the stream -- whole, cut by an end of the stream, or outside */
        static_assert(k4TailLead % 16 == 0 && (k4TailLead + 64) % 16 == 0, "whole chunks");
        for (uint32_t c = (uint32_t)(b - kProbeBlocks) * 256u + threadIdx.x; c < tail_bytes / 16u; c += (uint32_t)kTailBlocks * 256u) {
            const int64_t q = (int64_t)last_base + (int64_t)c * 16 - k4TailLead;
            if (q >= 0 && (uint64_t)q + 16 <= n) {
                *reinterpret_cast<u32x4*>(tail + 16u * c) = *reinterpret_cast<const u32x4*>(stream + q);
            } else {
                for (int j = 0; j < 16; ++j) tail[16u * c + j] = (q + j >= 0 && (uint64_t)(q + j) < n) ? stream[q + j] : (uint8_t)0xFF;
            }
        }
    } else {
        const uint64_t t0 = (uint64_t)(b - kProbeBlocks - kTailBlocks) * 256u + threadIdx.x;
        const uint64_t step = (uint64_t)((int)gridDim.x - sample_blocks - kProbeBlocks - kTailBlocks) * 256u;
        /* the index in 16-byte stores (a stream of small NALs has an index an eighth of its size: 268 MB behind 2 GiB of 256-byte
         * NALs, cleared 8 bytes a store until round 6); the index is 8-byte aligned at least: a word in front and one behind as needed */
        {
            const uint64_t head = (reinterpret_cast<uintptr_t>(index_words) & 8u) && n_index_words ? 1u : 0u;
            const uint64_t quads = (n_index_words - head) >> 1, rest = head + 2u * quads;
            u32x4* const q = reinterpret_cast<u32x4*>(index_words + head);
            for (uint64_t i = t0; i < quads; i += step) q[i] = u32x4{0u, 0u, 0u, 0u};
            if (t0 == 0 && head) index_words[0] = 0ull;
            if (t0 == 1 && rest < n_index_words) index_words[rest] = 0ull;
        }
        for (uint64_t i = t0; i < n_desc_words; i += step) desc[i] = 0ull;
    }
}

void launch_scan_prologue(const ScanArgs& a, uint64_t desc_words, bool probe, int tail_tile_bytes, hipStream_t st)
{
    const uint64_t index_words = a.index_cap * (sizeof(hbs_nal_entry) / 8);
    uint64_t clear_blocks = (index_words + desc_words) / (256u * 16u);
    if (clear_blocks < (uint64_t)kClearBlocksMin) clear_blocks = kClearBlocksMin;
    if (clear_blocks > (uint64_t)kClearBlocksMax) clear_blocks = kClearBlocksMax;
    /* the count-ahead's sample rides in this launch: a wavefront per sixteen 192 KiB tiles, at most 512 workgroups of four */
    unsigned sample_blocks = 0;
    if (a.ahead_cand && a.ahead_tab) {
        const uint64_t tiles = (a.n + (uint64_t)k4TileBytes - 1) / (uint64_t)k4TileBytes;
        const uint64_t want = (tiles + 63) / 64;
        sample_blocks = (unsigned)(want < 1 ? 1 : want > 512 ? 512 : want);
    }
    k_scan_prologue<<<dim3(kProbeBlocks + kTailBlocks + (unsigned)clear_blocks + sample_blocks), dim3(256), 0, st>>>(
        a.stream, a.n, a.hdr, a.tail, reinterpret_cast<unsigned long long*>(a.index), index_words,
        a.desc, desc_words, probe ? 1 : 0, tail_tile_bytes, a.ahead_cand, a.ahead_tab, a.ahead_list, a.ahead_ctl, (int)sample_blocks);
}

#endif // HBS4_MAIN
#if !HBS4_MAIN && defined(HBS_PHASE_TIMING)
extern "C" int hbs_debug_phase_cycles4_r24(unsigned long long* host_out /* [1024][8] */)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_phase_cycles4), sizeof(unsigned long long) * 1024 * 8);
}
#endif

#if HBS4_MAIN
#define HBS4_FN(name) scan4_##name
#else
#define HBS4_FN(name) scan4r24_##name
#endif
int HBS4_FN(grid_blocks)(int device, int* blocks_per_cu_out)
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return -1;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, HBS4_KERNEL, k4Threads, 0) != hipSuccess) return -1;
    if (per_cu < 1) per_cu = 1;
    if (blocks_per_cu_out) *blocks_per_cu_out = per_cu;
    return prop.multiProcessorCount * per_cu;
}

#if !HBS4_MAIN
int scan4r24_tile_bytes() { return k4TileBytes; }
#endif

/* num_tiles: tiles of THIS geometry over the stream */
void HBS4_FN(launch_kernel)(const ScanArgs& a, uint64_t num_tiles, int gate, hipStream_t st)
{
    uint64_t grid = (uint64_t)(HBS4_MAIN ? a.grid_blocks4 : a.grid_blocks4r24);
    if (grid > num_tiles) grid = num_tiles;
    /* the padded copy starts at the stream's last 192 KiB tile, whichever geometry reads it (scan4_tile_bytes) */
    const uint64_t tail_tile = (uint64_t)scan4_tile_bytes();
    const uint64_t tail_base = a.n ? ((a.n - 1) / tail_tile) * tail_tile : 0;
    HBS4_KERNEL<<<dim3((unsigned)grid), dim3(k4Threads), 0, st>>>(
        a.stream, a.n, num_tiles, a.index, a.index_cap, a.rbsp, a.rbsp_cap, a.desc, a.hdr, a.tail, tail_base, gate, a.first_static);
}

} // namespace hbs
