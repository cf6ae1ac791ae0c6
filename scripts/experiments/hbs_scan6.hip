/*
 * hbs_scan6.hip -- K12, event-sparse form with independent wavefronts.
 *
 * Same contract, tile algebra, descriptors and element code as hbs_scan4.hip (reference loop
 * find_nal_unit + nal_to_rbsp, h264_nal.c:38-76 / :147-200, driven as in hevc_analyze.c:135-177).
 * What differs is the schedule.  hbs_scan4.hip walks a tile through its phases -- fetch, flags,
 * elements, look-back, copy -- with the whole workgroup in step and one wavefront doing all the
 * serial work, so a compute unit has two phase sequences to overlap and its memory pipeline idles for
 * 40 % of a tile's life (round 1's phase table).  Here:
 *
 *   - a wavefront holds TWO sets of k6Rows rows (set 0 and set 1, each a quarter of its own tile) and
 *     alternates between them: while the look-back of one set's tile ripens and the loads of its next
 *     tile fly, the wavefront flags, walks and copies the other set.  Per wavefront the order is
 *         flags+elements(0)  flags+elements(1)  prefix+copy(0) load(0')  prefix+copy(1) load(1') ...
 *   - every wavefront handles the elements of its own rows (their bytes are in its own registers);
 *     the four wavefront aggregates of a tile meet in LDS;
 *   - there is no s_barrier in the loop.  The wavefront whose aggregate arrives last publishes the
 *     tile's aggregate and, when it comes back to that set, resolves the look-back (taking the ticket
 *     for the set's next tile meanwhile); the others wait for an LDS word.  One wavefront writes both
 *     states of a tile's descriptor, so they cannot overtake each other.
 *
 * A compute unit so has sixteen phase sequences in flight instead of two, and a dense tile (every chunk
 * an element) is walked by four wavefronts, 64 elements at a time each, instead of one.
 */
#include <hip/hip_runtime.h>
#include <utility>
#include "hbs_wave.h"
#include "hbs_sparse.h"
#include "hbs_scan.h"
#include "hbs_elems.h"

namespace hbs {

#ifndef HBS6_ROWS
#define HBS6_ROWS 16
#endif
constexpr int k6Rows        = HBS6_ROWS;                 /* rows of 1 KiB per wavefront per set */
constexpr int k6Waves       = 4;
constexpr int k6Threads     = 64 * k6Waves;
constexpr int k6WaveBytes   = k6Rows * 1024;
constexpr int k6TileBytes   = k6Waves * k6WaveBytes;
constexpr int k6WaveChunks  = k6Rows * 64;
constexpr int k6DepCap      = 64;
constexpr int k6TailLead    = 16;
constexpr int k6TailBytes   = k6TailLead + k6TileBytes + 64;
static_assert(k6TileBytes >= kTileBytes, "the descriptor workspace is sized for kTileBytes tiles");
static_assert(k6TailBytes <= k4TailBytes, "the padded last-tile copy shares hbs_scan4's workspace");
static_assert(k6Rows <= 32, "row numbers index the lanes of the mask registers; the chunk list holds 16-bit numbers");

constexpr uint32_t kNoTile = 0xFFFFFFFFu;

#ifdef HBS_PHASE_TIMING
/* diagnostic build only: shader-clock stamps of workgroup 0..7's wavefronts over their first iterations */
constexpr int k6TlWgs = 8, k6TlIters = 48, k6TlMarks = 12;
__device__ unsigned long long g_timeline6[k6TlWgs][k6Waves][k6TlIters][k6TlMarks];
#define HBS6_MARK(i) { if (blockIdx.x < k6TlWgs && tl_iter < k6TlIters && (threadIdx.x & 63) == 0) g_timeline6[blockIdx.x][threadIdx.x >> 6][tl_iter][i] = __builtin_amdgcn_s_memtime(); }
#define HBS6_ITER_DECL int tl_iter = 0;
#define HBS6_ITER_NEXT ++tl_iter;
#define HBS6_TL_ARG , int tl_iter, int tl_set
#define HBS6_TL_PASS(set) , tl_iter, set
#else
#define HBS6_MARK(i)
#define HBS6_ITER_DECL
#define HBS6_ITER_NEXT
#define HBS6_TL_ARG
#define HBS6_TL_PASS(set)
#endif

/* compile-time row loop: the row number must be a constant wherever it names a lane or a register */
template <class F, int... Is>
__device__ __forceinline__ void rows_apply(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <class F>
__device__ __forceinline__ void for_rows(F&& f) { rows_apply(static_cast<F&&>(f), std::make_integer_sequence<int, k6Rows>{}); }

/* one set of a wavefront: its rows and what the flag pass found in them */
struct Set6 {
    u32x4 q[k6Rows];
    uint32_t before, before2, after;   /* the two dwords in front of the segment, the dword behind it */
    uint32_t fm_lo, fm_hi;             /* lane r: flag mask of row r                                   */
    uint32_t local_pre;                /* lane r: elements of the segment in front of row r           */
    uint64_t rowmask;                  /* rows that hold an element                                    */
    uint32_t nelem;                    /* elements of this wavefront's segment                         */
    uint32_t tile;                     /* kNoTile: the set has run out of tiles                        */
    uint32_t gen;                      /* generation of the set (1, 2, ...)                            */
};

/* LDS, per wavefront per set: what its flag pass leaves for the tile's element wavefront */
struct WaveSet6 {
    Deposit dep[k6DepCap];             /* bytes of the segment's first elements                        */
    uint16_t list[k6WaveChunks];       /* flagged chunks of the segment (numbered in the segment), in stream order */
};
/* LDS, per set: where the four wavefronts of a workgroup meet.  Counters only grow. */
struct Slot6 {
    uint32_t tile_gen;                 /* `tile` is the tile of generation tile_gen                   */
    uint32_t tile;
    uint32_t arrived;                  /* flag passes finished, all generations (4 per generation)    */
    uint32_t stamp;                    /* 1024 gen + p + 1: the segment words of pass p are in seg[]  */
    uint32_t copied;                   /* tiles of several passes: copies finished (4 per pass)       */
    uint32_t ok;                       /* 0 abort, 1 fine, 2 arena too small                          */
    uint32_t nel[2][k6Waves];          /* elements per wavefront, by generation parity: a fast wavefront writes the next
                                          generation's while a slow one still reads this one's */
    uint32_t agg_cnt, agg_known, agg_sig, agg_last;    /* the tile's aggregate, kept for the look-back */
    uint32_t ex_inside;
    unsigned long long ex_kept, ex_nals;               /* the tile's exclusive prefix                  */
    uint32_t seg[k4ElemPass + 1];      /* segment words: [0] tile start, [i+1] element i of the pass   */
};
struct Lds6 {
    Slot6 slot[2];
    uint32_t abort_all;
    WaveSet6 ws[2][k6Waves];
};
static_assert(k6WaveChunks * k6Waves / k4ElemPass < 1023, "pass numbers share a word with the generation");

__device__ __forceinline__ uint32_t lds_load(const uint32_t* p)
{
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_store(uint32_t* p, uint32_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

/* wait until *p >= want (another wavefront of this workgroup sets it); false when the workgroup gave up */
__device__ __forceinline__ bool lds_wait_ge(const uint32_t* p, uint32_t want, const uint32_t* abort_all)
{
    for (uint32_t spins = 0;; ++spins) {
        if ((int32_t)(lds_load(p) - want) >= 0) return true;
        if (lds_load(abort_all) != 0u || spins > (1u << 24)) return false;
        __builtin_amdgcn_s_sleep(2);
    }
}

/* the wavefront that walks the elements of the set's tile of generation g and resolves its prefix: it rotates, and the two
 * sets of a generation have different ones, so the serial work of consecutive tiles runs on different wavefronts */
__device__ __forceinline__ int element_wave(int set, uint32_t gen) { return (int)((gen + 2u * (uint32_t)set) & 3u); }
static_assert(k6Waves == 4, "element_wave() rotates over four wavefronts");

struct TileGeo6 {
    uint64_t base, tile_end;
    const uint8_t* src;                /* the stream, or the padded copy for the last tile */
    bool last_tile;
};
__device__ __forceinline__ TileGeo6 tile_geo(uint32_t tile, const uint8_t* __restrict__ stream, const uint8_t* __restrict__ tail, uint64_t num_tiles)
{
    TileGeo6 g;
    g.base = (uint64_t)tile * (uint64_t)k6TileBytes;
    g.tile_end = g.base + (uint64_t)k6TileBytes;
    g.last_tile = (uint64_t)tile == num_tiles - 1;
    /* the last tile comes from its padded copy: tail[k6TailLead + i] = stream[base + i] */
    g.src = g.last_tile ? reinterpret_cast<const uint8_t*>(reinterpret_cast<uintptr_t>(tail) + (uintptr_t)k6TailLead - (uintptr_t)g.base) : stream;
    return g;
}

/* ---- load --------------------------------------------------------------------------------------- */
__device__ __forceinline__ void set_load(Set6& S, const uint8_t* __restrict__ stream, const uint8_t* __restrict__ tail,
                                         uint64_t n, uint64_t num_tiles, int wv, int lane)
{
    const TileGeo6 g = tile_geo(S.tile, stream, tail, num_tiles);
    const uint64_t wseg = g.base + (uint64_t)(wv * k6WaveBytes);
    const u32x4* p = reinterpret_cast<const u32x4*>(g.src + wseg) + lane;
#pragma unroll
    for (int r = 0; r < k6Rows; ++r) S.q[r] = stream_load16(p + r * 64);
    S.before = (wseg >= 4) ? stream_load4(g.src + wseg - 4) : 0xFFFFFFFFu;
    S.before2 = (wseg >= 8) ? stream_load4(g.src + wseg - 8) : 0xFFFFFFFFu;
    S.after = (g.last_tile || wv != k6Waves - 1) ? stream_load4(g.src + wseg + k6WaveBytes)
                                                 : load_dword_guarded(stream, (int64_t)(wseg + k6WaveBytes), n);
}

/* ---- flags: the set's rows have landed ------------------------------------------------------------ */
__device__ __forceinline__ void set_flags(Set6& S, WaveSet6& ws, Slot6& sl, uint64_t n, uint64_t num_tiles, int wv, int lane)
{
    const uint64_t wbase = (uint64_t)S.tile * (uint64_t)k6TileBytes + (uint64_t)(wv * k6WaveBytes);
    const uint64_t wend = wbase + (uint64_t)k6WaveBytes;
    const bool last_tile = (uint64_t)S.tile == num_tiles - 1;

    uint32_t fm_lo = 0, fm_hi = 0, wslot = 0;
    for_rows([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        const uint32_t e_prev = (r == 0) ? S.before : (uint32_t)__builtin_amdgcn_readlane((int)S.q[r ? r - 1 : 0].w, 63);
        const uint32_t e_next = (r == k6Rows - 1) ? S.after : (uint32_t)__builtin_amdgcn_readlane((int)S.q[r + 1 < k6Rows ? r + 1 : r].x, 0);
        const uint32_t xp = from_prev_lane(S.q[r].w, e_prev);
        const uint32_t xn = from_next_lane(S.q[r].x, e_next);
        const bool f = chunk_flag(xp, S.q[r].x, S.q[r].y, S.q[r].z, S.q[r].w, xn);
        const uint64_t fmask = __ballot(f);
        if (fmask != 0) {            /* rare: stash the mask, leave the chunk's surroundings for its element thread */
            write_lane_c<r>(fm_lo, (uint32_t)fmask);
            write_lane_c<r>(fm_hi, (uint32_t)(fmask >> 32));
            const uint32_t e_prev_z = (r == 0) ? S.before2 : (uint32_t)__builtin_amdgcn_readlane((int)S.q[r ? r - 1 : 0].z, 63);
            const uint32_t xpp = from_prev_lane(S.q[r].z, e_prev_z);
            const uint32_t slot = wslot + lanes_below(fmask);
            if (f && slot < (uint32_t)k6DepCap) {
                Deposit d;
                d.xpp = xpp; d.xp = xp; d.x0 = S.q[r].x; d.x1 = S.q[r].y; d.x2 = S.q[r].z; d.x3 = S.q[r].w; d.xn = xn;
                d.chunk = (uint32_t)(64 * r + lane);
                ws.dep[slot] = d;
            }
            wslot += (uint32_t)__builtin_popcountll(fmask);
        }
    });
    if (last_tile && (n & 15ull) != 0 && n > wbase && n < wend) {
        /* the chunk cut by the stream end is always an element */
        const uint32_t cut = (uint32_t)(n - wbase) >> 4;
        const int cr = (int)(cut >> 6), cl = (int)(cut & 63u);
        const uint64_t have = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)fm_hi, cr) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)fm_lo, cr);
        if (!((have >> cl) & 1ull)) {
            /* not flagged by its bytes: nothing was deposited for it; it is the wavefront's last element and its thread reads the stream itself */
            if (lane == 0 && wslot < (uint32_t)k6DepCap) ws.dep[wslot].chunk = 0xFFFFFFFFu;
            if (lane == cr) { if (cl < 32) fm_lo |= 1u << cl; else fm_hi |= 1u << (cl - 32); }
        }
    }
    S.fm_lo = fm_lo; S.fm_hi = fm_hi;
    {
        const uint32_t cnt = (lane < k6Rows) ? (uint32_t)__builtin_popcount(fm_lo) + (uint32_t)__builtin_popcount(fm_hi) : 0u;
        const uint32_t inc = wave_incl_scan32(cnt, lane);
        S.local_pre = inc - cnt;
        S.rowmask = __ballot(cnt != 0u);
        S.nelem = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    }
    /* the segment's chunk list, in stream order */
    for (uint64_t rm = S.rowmask; rm != 0ull; rm &= rm - 1ull) {
        const int r = __builtin_ctzll(rm);
        const uint32_t rp = (uint32_t)__builtin_amdgcn_readlane((int)S.local_pre, r);
        const uint64_t f = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)S.fm_hi, r) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)S.fm_lo, r);
        if ((f >> lane) & 1ull) ws.list[rp + lanes_below(f)] = (uint16_t)(64 * r + lane);
    }
    if (lane == 0) {
        sl.nel[S.gen & 1u][wv] = S.nelem;
        (void)__hip_atomic_fetch_add(&sl.arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

/* ---- the tile's elements, by its element wavefront ------------------------------------------------- */
struct TileElems6 { uint32_t wb1, wb2, wb3, nflag; };
__device__ __forceinline__ TileElems6 tile_elems(const Slot6& sl, uint32_t gen)
{
    TileElems6 t;
    const uint32_t* nel = sl.nel[gen & 1u];
    const uint32_t n0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)nel[0]), n1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)nel[1]);
    const uint32_t n2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)nel[2]), n3 = (uint32_t)__builtin_amdgcn_readfirstlane((int)nel[3]);
    t.wb1 = n0; t.wb2 = n0 + n1; t.wb3 = t.wb2 + n2; t.nflag = t.wb3 + n3;
    return t;
}

/* Elements [i0, i0 + 64) of the tile, one per lane: bytes from the deposit the flagging lane left (or from the stream), the
 * exact rules, the gap in front.  prev_end = end of the element in front of element i0 (stream offset), carried from batch to batch. */
__device__ __forceinline__ TileAgg tile_batch(Elem& el, const WaveSet6* ws /* [k6Waves] */, const TileElems6& te, uint32_t i0, int lane,
                                              const TileGeo6& g, uint64_t n, uint64_t& prev_end)
{
    const uint32_t i = i0 + (uint32_t)lane;
    const bool have = i < te.nflag;
    uint32_t c = 0, ew = 0, ej = 0;
    bool have_dep = false;
    if (have) {
        ew = (i >= te.wb1 ? 1u : 0u) + (i >= te.wb2 ? 1u : 0u) + (i >= te.wb3 ? 1u : 0u);
        ej = i - (ew == 0u ? 0u : ew == 1u ? te.wb1 : ew == 2u ? te.wb2 : te.wb3);
        const uint32_t cl = ws[ew].list[ej];
        c = (uint32_t)k6WaveChunks * ew + cl;
        have_dep = ej < (uint32_t)k6DepCap && ws[ew].dep[ej < (uint32_t)k6DepCap ? ej : 0u].chunk == cl;
    }
    const uint32_t c_prev = (uint32_t)__shfl_up((int)c, 1, 64);
    const uint64_t my_prev_end = lane == 0 ? prev_end : g.base + 16ull * ((uint64_t)c_prev + 1u);
    TileAgg ea = agg_identity();
    el.gap = 0; el.chunk = 0;
    if (have) {
        if (have_dep) {        /* field by field: a copy of the whole struct ends up in scratch memory */
            const Deposit& d = ws[ew].dep[ej];
            el.v.xpp = d.xpp; el.v.xp = d.xp; el.v.x0 = d.x0; el.v.x1 = d.x1; el.v.x2 = d.x2; el.v.x3 = d.x3; el.v.xn = d.xn;
            el.v.stream = g.src; el.v.g0 = g.base + 16ull * c; el.v.n = n;
        } else {
            elem_load(el.v, g.src, g.base + 16ull * c, n, g.last_tile);
        }
        elem_walk(el.v, el.m, el.s, el.cls);
        el.gap = span_bytes(my_prev_end, el.v.g0, n);
        el.chunk = c;
        ea = elem_agg(el.gap, el.s);
    }
    const uint32_t cnt = te.nflag - i0 < 64u ? te.nflag - i0 : 64u;
    const uint32_t c_last = (uint32_t)__shfl((int)c, (int)(cnt - 1u), 64);
    prev_end = g.base + 16ull * ((uint64_t)c_last + 1u);
    return ea;
}

/* all four flag passes are in: the tile's aggregate goes out */
__device__ __forceinline__ bool tile_aggregate_publish(const Set6& S, Slot6& sl, const WaveSet6* ws, uint32_t* abort_all,
                                                       const uint8_t* __restrict__ stream, const uint8_t* __restrict__ tail, uint64_t n, uint64_t num_tiles,
                                                       unsigned long long* __restrict__ desc, int lane)
{
    if (!lds_wait_ge(&sl.arrived, (uint32_t)k6Waves * S.gen, abort_all)) return false;
    const TileGeo6 g = tile_geo(S.tile, stream, tail, num_tiles);
    const TileElems6 te = tile_elems(sl, S.gen);
    const uint32_t npass = (te.nflag + (uint32_t)k4ElemPass - 1u) / (uint32_t)k4ElemPass;
    TileAgg acc = agg_identity();
    uint64_t prev_end = g.base;
#pragma unroll 1
    for (uint32_t p = 0; p < npass; ++p) {
        Elem el;
        TileAgg ea = tile_batch(el, ws, te, p * (uint32_t)k4ElemPass, lane, g, n, prev_end);
        ea = wave_scan_combine(ea, lane);
        acc = combine(acc, agg_readlane(ea, 63));
    }
    const TileAgg tagg = combine(acc, gap_agg(span_bytes(prev_end, g.tile_end, n)));
    look_back_publish(desc, (uint64_t)S.tile, tagg, lane);
    if (lane == 0) { sl.agg_cnt = tagg.cnt; sl.agg_known = tagg.known; sl.agg_sig = tagg.sig; sl.agg_last = tagg.last; }
    return true;
}

/* The element wavefront resolves the tile's prefix: takes the ticket of the set's next tile (its round trip runs under the
 * look-back), folds the tiles in front, leaves the tile's exclusive prefix in LDS. */
__device__ __forceinline__ bool tile_resolve(const Set6& S, Slot6& sl, uint32_t* abort_all, unsigned long long* __restrict__ desc,
                                             RunHeader* __restrict__ hdr, uint8_t* rbsp, uint64_t rbsp_cap, uint64_t num_tiles, int lane)
{
    uint32_t tnext = 0;
    if (lane == 0) tnext = atomicAdd(&hdr->ticket, 1u);
    TileAgg tagg;
    tagg.cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)sl.agg_cnt); tagg.known = (uint32_t)__builtin_amdgcn_readfirstlane((int)sl.agg_known);
    tagg.sig = (uint32_t)__builtin_amdgcn_readfirstlane((int)sl.agg_sig); tagg.last = (uint32_t)__builtin_amdgcn_readfirstlane((int)sl.agg_last);
    Prefix ex;
    uint32_t it, stl;
    const bool ok = look_back_resolve(desc, (uint64_t)S.tile, tagg, hdr, lane, ex, it, stl);
    ex = prefix_uniform4(ex);
    const uint32_t tile_kept = tagg.known + (ex.inside ? tagg.sig : 0u);
    const bool can = rbsp != nullptr && ex.kept + tile_kept <= rbsp_cap;
    if (lane == 0) {
        if (!ok) {
            lds_store(abort_all, 1u);
        } else {
            if (rbsp != nullptr && !can) atomicMax(&hdr->error, (uint32_t)(-HBS_E_CAPACITY));
            if ((uint64_t)S.tile == num_tiles - 1) {
                const Prefix incl = fold(ex, tagg);
                hdr->final_kept = incl.kept; hdr->final_nals = incl.nals; hdr->final_inside = incl.inside;
            }
        }
        sl.ex_kept = ex.kept; sl.ex_nals = ex.nals; sl.ex_inside = ex.inside;
        sl.ok = !ok ? 0u : (rbsp != nullptr && !can) ? 2u : 1u;
        sl.seg[0] = seg_pack(-1, 0u, ex.inside != 0u);
        sl.tile = tnext;
        lds_store(&sl.tile_gen, S.gen + 1u);
    }
    return ok;
}

/* ---- prefix, the elements' second half, copy -------------------------------------------------------- */
/* what the element wavefront carries from one batch of elements to the next */
struct EmitState6 { TileAgg accb; uint64_t prev_end; };

/* one batch of the tile's elements, second half: index entries, their own kept bytes, the segment words of the batch */
__device__ __forceinline__ void tile_emit_batch(const Set6& S, Slot6& sl, const WaveSet6* ws, const TileElems6& te, uint32_t p, EmitState6& st,
                                                const TileGeo6& g, uint64_t n, const Prefix& excl, bool can_store, uint8_t* out, const EmitTarget& tgt, int lane)
{
    const uint32_t pbase = p * (uint32_t)k4ElemPass;
    if (te.nflag != 0u) {
        Elem el;
        TileAgg ea = tile_batch(el, ws, te, pbase, lane, g, n, st.prev_end);
        ea = wave_scan_combine(ea, lane);
        TileAgg up = agg_shfl_up(ea, 1);
        if (lane == 0) up = agg_identity();
        const TileAgg e = combine(st.accb, up);
        st.accb = combine(st.accb, agg_readlane(ea, 63));
        if (pbase + (uint32_t)lane < te.nflag) elem_emit(el, e, excl, can_store, out, tgt, &sl.seg[lane + 1]);
    }
    if (lane == 0) lds_store(&sl.stamp, 1024u * S.gen + p + 1u);
}

/* The element wavefront of the set's tile: prefix, first batch.  Runs BEFORE this wavefront waits for anybody else's
 * tile, so that the serial work of the workgroup's two tiles proceeds on two wavefronts at once. */
__device__ __forceinline__ bool set_resolve_emit(const Set6& S, Slot6& sl, const WaveSet6* ws, uint32_t* abort_all, EmitState6& st,
                                                 const uint8_t* __restrict__ stream, const uint8_t* __restrict__ tail, uint64_t n, uint64_t num_tiles,
                                                 uint8_t* __restrict__ rbsp, uint64_t rbsp_cap, unsigned long long* __restrict__ desc,
                                                 RunHeader* __restrict__ hdr, const EmitTarget& tgt, int lane)
{
    if (!tile_resolve(S, sl, abort_all, desc, hdr, rbsp, rbsp_cap, num_tiles, lane)) {
        if (lane == 0) lds_store(&sl.stamp, 1024u * S.gen + 1023u);      /* wake the others: they see abort_all / ok == 0 */
        return false;
    }
    Prefix excl;
    {
        Prefix ex;
        ex.kept = sl.ex_kept; ex.nals = sl.ex_nals; ex.inside = sl.ex_inside;
        excl = prefix_uniform4(ex);
    }
    const bool can_store = rbsp != nullptr && (uint32_t)__builtin_amdgcn_readfirstlane((int)sl.ok) == 1u;
    const TileGeo6 g = tile_geo(S.tile, stream, tail, num_tiles);
    const TileElems6 te = tile_elems(sl, S.gen);
    st.accb = agg_identity();
    st.prev_end = g.base;
    tile_emit_batch(S, sl, ws, te, 0u, st, g, n, excl, can_store, rbsp + excl.kept, tgt, lane);
    return true;
}

/* everybody: copy what each batch serves; the element wavefront emits the batches behind the first in between.
 * false: the workgroup gives up (a look-back timed out somewhere) */
__device__ __forceinline__ bool set_copy(Set6& S, bool mine, Slot6& sl, const WaveSet6* ws, uint32_t* abort_all, EmitState6& st,
                                         const uint8_t* __restrict__ stream, const uint8_t* __restrict__ tail, uint64_t n, uint64_t num_tiles,
                                         uint8_t* __restrict__ rbsp, const EmitTarget& tgt, int wv, int lane HBS6_TL_ARG)
{
    /* the first batch's segment words are in: so are the prefix and everybody's element counts */
    if (!mine && !lds_wait_ge(&sl.stamp, 1024u * S.gen + 1u, abort_all)) return false;
    HBS6_MARK(4 + 4 * tl_set)
    const uint32_t okv = (uint32_t)__builtin_amdgcn_readfirstlane((int)sl.ok);
    if (okv == 0u) return false;
    Prefix excl;
    {
        Prefix ex;
        ex.kept = sl.ex_kept; ex.nals = sl.ex_nals; ex.inside = sl.ex_inside;
        excl = prefix_uniform4(ex);
    }
    const bool can_store = rbsp != nullptr && okv == 1u;
    uint8_t* const out = rbsp + excl.kept;
    const TileGeo6 g = tile_geo(S.tile, stream, tail, num_tiles);
    const TileElems6 te = tile_elems(sl, S.gen);
    const uint32_t npass = (te.nflag + (uint32_t)k4ElemPass - 1u) / (uint32_t)k4ElemPass;
    const uint32_t np = npass ? npass : 1u;
    const uint32_t wave_base = (wv == 0) ? 0u : (wv == 1) ? te.wb1 : (wv == 2) ? te.wb2 : te.wb3;
    const uint32_t whole = (uint32_t)(span_bytes(g.base, g.tile_end, n) >> 4);      /* chunks of the tile that are complete */
    const uint32_t copied0 = (mine && np > 1u) ? (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_load(&sl.copied)) : 0u;
#pragma unroll 1
    for (uint32_t p = 0; p < np; ++p) {
        const uint32_t pbase = p * (uint32_t)k4ElemPass;
        if (p != 0u) {
            if (mine) {
                /* everybody has copied what the previous batch served: its segment words may go */
                if (!lds_wait_ge(&sl.copied, copied0 + (uint32_t)k6Waves * p, abort_all)) return false;
                tile_emit_batch(S, sl, ws, te, p, st, g, n, excl, can_store, out, tgt, lane);
            } else if (!lds_wait_ge(&sl.stamp, 1024u * S.gen + p + 1u, abort_all)) {
                return false;
            }
        }
        if (can_store) {
            /* lane j: segment word j of this batch (j = 0..63), word 64 apart: a row without elements needs one
             * word, picked with a readlane instead of an LDS round trip */
            const uint32_t segv = sl.seg[lane];
            const uint32_t seg64 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sl.seg[k4ElemPass]);
            for_rows([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const uint32_t cc = (uint32_t)(64 * (k6Rows * wv + r) + lane);
                const uint32_t rowpre = wave_base + (uint32_t)__builtin_amdgcn_readlane((int)S.local_pre, r);
                if (!((S.rowmask >> r) & 1ull)) {          /* no element in this row: one k, one word for all lanes */
                    const uint32_t k = rowpre;
                    const bool served = p == 0u ? k <= (uint32_t)k4ElemPass : (k > pbase && k <= pbase + (uint32_t)k4ElemPass);
                    if (served) {
                        const uint32_t j = k - pbase;
                        const uint32_t w = (j == (uint32_t)k4ElemPass) ? seg64 : (uint32_t)__builtin_amdgcn_readlane((int)segv, (int)(j & 63u));
                        if (seg_inside(w) && cc < whole) arena_store16(out + (int64_t)seg_bias(w) + 16u * cc, S.q[r]);
                    }
                } else {
                    const uint64_t f = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)S.fm_hi, r) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)S.fm_lo, r);
                    const uint32_t k = rowpre + lanes_below(f);
                    const bool served = p == 0u ? k <= (uint32_t)k4ElemPass : (k > pbase && k <= pbase + (uint32_t)k4ElemPass);
                    if (!((f >> lane) & 1ull) && served && cc < whole) {
                        const uint32_t w = sl.seg[k - pbase];
                        if (seg_inside(w)) arena_store16(out + (int64_t)(seg_bias(w) + (int32_t)(16u * cc)), S.q[r]);
                    }
                }
            });
        }
        if (np > 1u && lane == 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      /* my reads of the segment words are done */
            (void)__hip_atomic_fetch_add(&sl.copied, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    return true;
}

/* the set's next tile: its ticket was taken by whoever resolved this generation */
__device__ __forceinline__ bool set_next(Set6& S, Slot6& sl, uint32_t* abort_all, uint64_t num_tiles)
{
    if (!lds_wait_ge(&sl.tile_gen, S.gen + 1u, abort_all)) return false;
    const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)sl.tile);
    S.gen += 1u;
    S.tile = ((uint64_t)t < num_tiles) ? t : kNoTile;
    return true;
}

__global__ __launch_bounds__(k6Threads, 2)
void k_scan_extract6(const uint8_t* __restrict__ stream, uint64_t n, uint64_t num_tiles,
                     hbs_nal_entry* __restrict__ index, uint64_t index_cap,
                     uint8_t* __restrict__ rbsp, uint64_t rbsp_cap,
                     unsigned long long* __restrict__ desc, RunHeader* __restrict__ hdr, const uint8_t* __restrict__ tail,
                     int gate)
{
    if (gate == kGateIfSparse && probe_dense_dev(hdr)) return;
    __shared__ Lds6 l;
    const int tid0 = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    EmitTarget tgt;
    tgt.index = index; tgt.index_cap = index_cap; tgt.hdr = hdr;
    if (tid0 == 0) {
        l.abort_all = 0;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            l.slot[s].tile = atomicAdd(&hdr->ticket, 1u);
            l.slot[s].tile_gen = 1; l.slot[s].arrived = 0; l.slot[s].stamp = 0; l.slot[s].copied = 0; l.slot[s].ok = 1;
        }
    }
    __syncthreads();
    Set6 A, B;
    A.gen = 0; B.gen = 0;
    HBS6_ITER_DECL
    int lane = launder_lane(tid0) & 63;
    if (!set_next(A, l.slot[0], &l.abort_all, num_tiles)) return;
    if (!set_next(B, l.slot[1], &l.abort_all, num_tiles)) return;
    if (A.tile != kNoTile) set_load(A, stream, tail, n, num_tiles, wv, lane);
    if (B.tile != kNoTile) set_load(B, stream, tail, n, num_tiles, wv, lane);

    while (A.tile != kNoTile || B.tile != kNoTile) {
        lane = launder_lane(tid0) & 63;
        HBS6_MARK(0)
        __builtin_amdgcn_s_setprio(2);        /* until the aggregates are out, others wait for this */
        if (A.tile != kNoTile) {
            set_flags(A, l.ws[0][wv], l.slot[0], n, num_tiles, wv, lane);
            HBS6_MARK(1)
            if (element_wave(0, A.gen) == wv && !tile_aggregate_publish(A, l.slot[0], l.ws[0], &l.abort_all, stream, tail, n, num_tiles, desc, lane)) return;
        }
        HBS6_MARK(2)
        if (B.tile != kNoTile) {
            set_flags(B, l.ws[1][wv], l.slot[1], n, num_tiles, wv, lane);
            if (element_wave(1, B.gen) == wv && !tile_aggregate_publish(B, l.slot[1], l.ws[1], &l.abort_all, stream, tail, n, num_tiles, desc, lane)) return;
        }
        lane = launder_lane(tid0) & 63;
        HBS6_MARK(3)
        /* the serial work of the two tiles, each on its own wavefront, before anybody waits for anybody */
        const bool mineA = A.tile != kNoTile && element_wave(0, A.gen) == wv, mineB = B.tile != kNoTile && element_wave(1, B.gen) == wv;
        EmitState6 st;
        st.accb = agg_identity(); st.prev_end = 0;
        if (mineA && !set_resolve_emit(A, l.slot[0], l.ws[0], &l.abort_all, st, stream, tail, n, num_tiles, rbsp, rbsp_cap, desc, hdr, tgt, lane)) return;
        if (mineB && !set_resolve_emit(B, l.slot[1], l.ws[1], &l.abort_all, st, stream, tail, n, num_tiles, rbsp, rbsp_cap, desc, hdr, tgt, lane)) return;
        __builtin_amdgcn_s_setprio(0);
        HBS6_MARK(5)
        if (A.tile != kNoTile) {
            if (!set_copy(A, mineA, l.slot[0], l.ws[0], &l.abort_all, st, stream, tail, n, num_tiles, rbsp, tgt, wv, lane HBS6_TL_PASS(0))) return;
            HBS6_MARK(6)
            if (!set_next(A, l.slot[0], &l.abort_all, num_tiles)) return;
            if (A.tile != kNoTile) set_load(A, stream, tail, n, num_tiles, wv, lane);
        }
        HBS6_MARK(7)
        if (B.tile != kNoTile) {
            if (!set_copy(B, mineB, l.slot[1], l.ws[1], &l.abort_all, st, stream, tail, n, num_tiles, rbsp, tgt, wv, lane HBS6_TL_PASS(1))) return;
            HBS6_MARK(10)
            if (!set_next(B, l.slot[1], &l.abort_all, num_tiles)) return;
            if (B.tile != kNoTile) set_load(B, stream, tail, n, num_tiles, wv, lane);
        }
        HBS6_MARK(11)
        HBS6_ITER_NEXT
    }
}

#ifdef HBS_PHASE_TIMING
extern "C" int hbs_debug_timeline6(unsigned long long* host_out /* [8][4][48][12] */)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_timeline6), sizeof(unsigned long long) * k6TlWgs * k6Waves * k6TlIters * k6TlMarks);
}
#endif

int scan6_tile_bytes() { return k6TileBytes; }
int scan6_tail_bytes() { return k6TailBytes; }

int scan6_grid_blocks(int device, int* blocks_per_cu_out)
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return -1;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_scan_extract6, k6Threads, 0) != hipSuccess) return -1;
    if (per_cu < 1) per_cu = 1;
    if (blocks_per_cu_out) *blocks_per_cu_out = per_cu;
    return prop.multiProcessorCount * per_cu;
}

void launch_scan_extract6_kernel(const ScanArgs& a, uint64_t num_tiles, int gate, hipStream_t st)
{
    /* every workgroup takes two tiles before it does anything else */
    uint64_t grid = (uint64_t)a.grid_blocks6;
    if (grid > (num_tiles + 1) / 2) grid = (num_tiles + 1) / 2;
    if (grid < 1) grid = 1;
    k_scan_extract6<<<dim3((unsigned)grid), dim3(k6Threads), 0, st>>>(
        a.stream, a.n, num_tiles, a.index, a.index_cap, a.rbsp, a.rbsp_cap, a.desc, a.hdr, a.tail, gate);
}

} // namespace hbs
