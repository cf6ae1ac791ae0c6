/*
 * hbs_scan4.hip -- K12, event-sparse form: fused start-code scan + NAL index +
 * RBSP extraction whose per-byte work is one conservative test and one copy.
 *
 * Same contract, descriptors and tile algebra as hbs_scan.hip (reference loop
 * find_nal_unit + nal_to_rbsp, h264_nal.c:38-76 / :147-200, driven as in
 * hevc_analyze.c:135-177).  What differs is who does the exact work:
 *
 *   1. A tile is 64 KiB; wavefront w holds its 8 KiB as 8 rows of 1 KiB in
 *      VGPRs (hbs_wave.h).  Per 16-byte chunk, chunk_flag() (hbs_sparse.h)
 *      decides that no pattern 00 00 {<=3} can touch it; the row's ballot is its
 *      flag mask.  Flagged chunks -- a start code per NAL, a few emulation
 *      prevention bytes, a few false alarms: ~15 of 4096 -- are listed in LDS
 *      in stream order.
 *   2. Thread i takes the i-th listed chunk ("element"): exact window logic of
 *      hbs_tile.h on its six dwords, re-read from L2.  A wave scan with
 *      combine() over (gap, chunk) elements gives the tile aggregate; normally
 *      only wavefront 0 has any element.
 *   3. Wavefront 0 alone runs the decoupled look-back, 256 predecessors per
 *      step (4 per lane); the others wait at a barrier.
 *   4. With the carried state known the elements emit index entries, write
 *      their own kept bytes, and leave one segment word each; every other chunk
 *      finds the word of the nearest element in front of it (row prefix +
 *      mbcnt of the flag mask) and, if inside a NAL, is one byte-aligned
 *      16-byte store straight from its registers.
 *
 * Tiles are handed out by an atomic ticket in arrival order, so a workgroup
 * only ever waits for tiles that are already being worked on: no co-residency
 * requirement, and a slow workgroup delays its successors, not a whole round.
 * Streams dense in zero pairs (every chunk an element) stay exact: elements are
 * processed kThreads at a time.
 */
#include <hip/hip_runtime.h>
#include "hbs_wave.h"
#include "hbs_sparse.h"
#include "hbs_scan.h"

namespace hbs {

#ifdef HBS_PHASE_TIMING
__device__ unsigned long long g_phase_cycles4[1024][8];
#define HBS4_T_DECL unsigned long long t_prev = __builtin_amdgcn_s_memtime(), t_acc[8] = {0,0,0,0,0,0,0,0};
#define HBS4_T_MARK(i) { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); t_acc[i] += t_now - t_prev; t_prev = t_now; }
#define HBS4_T_COUNT(i, v) { t_acc[i] += (v); }
#define HBS4_T_FLUSH if (threadIdx.x == 0 && blockIdx.x < 1024) { for (int i = 0; i < 8; ++i) g_phase_cycles4[blockIdx.x][i] = t_acc[i]; }
__device__ uint32_t g_dbg4[4096];
#define HBS4_DBG(code) code
#else
#define HBS4_DBG(code)
#define HBS4_T_DECL
#define HBS4_T_MARK(i)
#define HBS4_T_COUNT(i, v)
#define HBS4_T_FLUSH
#endif

static_assert(k4Rows == 32, "the row lists below name every row register");
#define HBS_ROWS(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30) X(31)
#define HBS_ROW_PAIRS(X) X(0,1) X(1,2) X(2,3) X(3,4) X(4,5) X(5,6) X(6,7) X(7,8) X(8,9) X(9,10) X(10,11) X(11,12) X(12,13) X(13,14) X(14,15) X(15,16) X(16,17) X(17,18) X(18,19) X(19,20) X(20,21) X(21,22) X(22,23) X(23,24) X(24,25) X(25,26) X(26,27) X(27,28) X(28,29) X(29,30) X(30,31)   /* (row, next row) */
#define HBS_ROW_TRIPLES(X) X(0,1,2) X(1,2,3) X(2,3,4) X(3,4,5) X(4,5,6) X(5,6,7) X(6,7,8) X(7,8,9) X(8,9,10) X(9,10,11) X(10,11,12) X(11,12,13) X(12,13,14) X(13,14,15) X(14,15,16) X(15,16,17) X(16,17,18) X(17,18,19) X(18,19,20) X(19,20,21) X(20,21,22) X(21,22,23) X(22,23,24) X(23,24,25) X(24,25,26) X(25,26,27) X(26,27,28) X(27,28,29) X(28,29,30) X(29,30,31)   /* (previous row, row, next row), inner rows */
#define HBS_ROWS_BUT_LAST(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24) X(25) X(26) X(27) X(28) X(29) X(30)

/* one wavefront's segment: k4Rows rows of 1 KiB in named registers + the dwords just outside */
struct RowRegs {
#define HBS_DECL(r) u32x4 q##r;
    HBS_ROWS(HBS_DECL)
#undef HBS_DECL
    uint32_t before;          /* dword in front of the segment (0xFFFFFFFF before the stream) */
    uint32_t before2;         /* the dword in front of that one                               */
    uint32_t after;           /* dword behind it (0xFF bytes past the end of the stream)      */
};

/* Row r of the register image, r wave-uniform: a scalar switch picks the registers, so that the
 * per-row phases are one rolled loop body each (unrolled over 16 named rows the compiler hoists
 * every row's addresses and constants out of the tile loop and runs out of registers). */
__device__ __forceinline__ u32x4 get_row(const RowRegs& R, int r)
{
    switch (r) {
#define HBS_CASE(i) case i: return R.q##i;
    HBS_ROWS_BUT_LAST(HBS_CASE)
#undef HBS_CASE
    default: return R.q31;
    }
}
__device__ __forceinline__ uint32_t row_first_dword(const RowRegs& R, int r)
{
    switch (r) {
#define HBS_CASE(i) case i: return R.q##i.x;
    HBS_ROWS_BUT_LAST(HBS_CASE)
#undef HBS_CASE
    default: return R.q31.x;
    }
}

/* All of a wavefront's rows, unguarded: the last tile of a stream is read from a padded copy
 * (k_prepare_tail4), so every address below exists.  `src` is the stream or that copy. */
__device__ __forceinline__ void fetch_row_regs(RowRegs& R, const uint8_t* __restrict__ src, uint64_t seg, int lane)
{
    const u32x4* p = reinterpret_cast<const u32x4*>(src + seg) + lane;
#define HBS_LD(r) R.q##r = p[r * 64];
    HBS_ROWS(HBS_LD)
#undef HBS_LD
}

/* what a flagged lane leaves for the thread that will handle its chunk as an element */
constexpr int kDepCap = 64;
struct Deposit { uint32_t xpp, xp, x0, x1, x2, x3, xn, chunk; };

/* an element between the two phases (single-pass tiles): its bytes, marks, summary, and where
 * it stands in the tile */
struct ElemState {
    uint32_t xpp, xp, x0, x1, x2, x3, xn;
    uint32_t pa, pb, pc;                   /* ElemPacked                                      */
    uint32_t gap;                          /* bytes of the gap in front of it                 */
    uint32_t chunk;
    TileAgg e;                             /* aggregate of the tile in front of that gap      */
};

struct Lds4 {
    uint32_t row_cnt[k4TileRows];          /* flagged chunks per row                                          */
    uint16_t list[k4ChunksPerTile];        /* flagged chunks of the tile, in stream order  */
    uint32_t seg[k4ElemPass + 1];          /* segment words: [0] tile start, [i+1] element i of the pass */
    ElemState est[k4ElemPass];
    Deposit dep[k4Waves][kDepCap];         /* bytes of the first elements of each wavefront, left by the flag pass */
    TileAgg wtot[k4Waves];                 /* per-wavefront element aggregates of a pass   */
    unsigned long long ex_kept, ex_nals;   /* the tile's exclusive prefix, from wavefront 0 */
    uint32_t ex_inside, ex_ok;
    uint32_t ticket;
};

__device__ __forceinline__ uint32_t lanes_below(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__device__ __forceinline__ TileAgg agg_shfl_up(const TileAgg& a, int d)
{
    TileAgg t;
    t.cnt = __shfl_up(a.cnt, d, 64); t.known = __shfl_up(a.known, d, 64);
    t.sig = __shfl_up(a.sig, d, 64); t.last = __shfl_up(a.last, d, 64);
    return t;
}

/* inclusive scan with combine(): lane l <- elements of lanes 0..l in order */
__device__ __forceinline__ TileAgg wave_scan_combine(TileAgg a, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const TileAgg t = agg_shfl_up(a, d);
        if (lane >= d) a = combine(t, a);
    }
    return a;
}

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int l)
{
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32) |
           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
}

/*
 * Decoupled look-back by ONE wavefront, 256 predecessors per step: lane l reads the
 * descriptors of tiles win_hi - 4l - j (j = 0 nearest).  A lane folds the tiles in front of
 * its nearest prefix; the lanes up to the first one that holds a prefix are folded by
 * window_fold3().  Returns false on timeout/abort.  Called by every lane of wavefront 0.
 */
__device__ __forceinline__ bool look_back4(unsigned long long* desc, uint64_t tile, const TileAgg& mine,
                                           RunHeader* hdr, int lane, Prefix& excl, uint32_t& dbg_iters, uint32_t& dbg_stalls)
{
    dbg_iters = 0; dbg_stalls = 0;
    bool ok = true;
    excl.kept = 0; excl.nals = 0; excl.inside = 0;
    if (tile != 0) {
        if (lane == 0) {
            st_desc3(&desc[2 * tile], pack_agg0(mine));
            st_desc3(&desc[2 * tile + 1], pack_agg1(mine));
        }
        TileAgg acc = agg_identity();                 /* tiles between the window and `tile` */
        int64_t win_hi = (int64_t)tile - 1;
        uint32_t spins = 0;
        for (;;) {
            ++dbg_iters;
            uint64_t w0[4], w1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t t = win_hi - (4 * lane + j);
                w0[j] = kDescPrefix; w1[j] = kDescPrefix;          /* virtual tile -1: empty prefix */
                if (t >= 0) {
                    w0[j] = ld_desc3(&desc[2 * t]);
                    w1[j] = ld_desc3(&desc[2 * t + 1]);
                }
            }
            int jp = 4;                     /* my nearest tile that already has its prefix */
            bool lane_ok = true;            /* every tile in front of it has its aggregate  */
            uint64_t pw0 = 0, pw1 = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t s0 = (uint32_t)(w0[j] & 3u), s1 = (uint32_t)(w1[j] & 3u);
                const bool ready = (s0 == s1) && (s0 != kDescEmpty);
                if (jp == 4) {
                    if (!ready) lane_ok = false;
                    else if (s0 == kDescPrefix) { jp = j; pw0 = w0[j]; pw1 = w1[j]; }
                }
            }
            TileAgg la = agg_identity();    /* earliest first: j = 3 is the earliest tile */
#pragma unroll
            for (int j = 3; j >= 0; --j)
                if (j < jp) la = combine(la, unpack_agg(w0[j], w1[j]));
            const uint64_t m_pre = __ballot(jp < 4 && lane_ok);
            const uint64_t m_ok = __ballot(lane_ok);
            const int lstar = m_pre ? (int)__builtin_ctzll(m_pre) : 64;
            const uint64_t need = (lstar >= 63) ? ~0ull : ((2ull << lstar) - 1ull);
            if ((m_ok & need) != need) {
                ++dbg_stalls;
                bool aborted = false;
                if ((spins & 63u) == 63u)
                    aborted = __hip_atomic_load(&hdr->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
                if (++spins > (1u << 20) || aborted) { ok = false; break; }
                continue;               /* the round trip of the next poll is delay enough */
            }
            const TileAgg win = window_fold3(la, lstar < 64 ? lstar + 1 : 64, lane);
            const TileAgg total = combine(win, acc);
            if (lstar < 64) {
                const Prefix p = unpack_pre(readlane_u64(pw0, lstar), readlane_u64(pw1, lstar));
                excl = fold(p, total);
                break;
            }
            acc = total;
            win_hi -= 256;
        }
    }
    if (lane == 0) {
        if (ok) {
            const Prefix incl = fold(excl, mine);
            st_desc3(&desc[2 * tile], pack_pre0(incl));
            st_desc3(&desc[2 * tile + 1], pack_pre1(incl));
        } else {
            __hip_atomic_store(&hdr->abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicMax(&hdr->error, (uint32_t)(-HBS_E_TIMEOUT));
        }
    }
    return ok;
}

__device__ __forceinline__ Prefix prefix_uniform4(const Prefix& p)
{
    Prefix r;
    r.kept = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(p.kept >> 32)) << 32) |
             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)p.kept);
    r.nals = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(p.nals >> 32)) << 32) |
             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)p.nals);
    r.inside = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.inside);
    return r;
}

/* the seven dwords around chunk g0 of the stream, for an element */
__device__ __forceinline__ void elem_load(ElemView& v, const uint8_t* __restrict__ stream, uint64_t g0, uint64_t n, bool padded)
{
    if (g0 >= 8 && (padded || g0 + 20 <= n)) {
        const uint32_t* p = reinterpret_cast<const uint32_t*>(stream + g0);
        const u32x4 q = *reinterpret_cast<const u32x4*>(p);
        v.xpp = p[-2]; v.xp = p[-1]; v.xn = p[4];
        v.x0 = q.x; v.x1 = q.y; v.x2 = q.z; v.x3 = q.w;
    } else {
        const u32x4 q = load_chunk_guarded(stream, g0, n);
        v.xpp = load_dword_guarded(stream, (int64_t)g0 - 8, n);
        v.xp = load_dword_guarded(stream, (int64_t)g0 - 4, n);
        v.xn = load_dword_guarded(stream, (int64_t)g0 + 16, n);
        v.x0 = q.x; v.x1 = q.y; v.x2 = q.z; v.x3 = q.w;
    }
    v.stream = stream; v.g0 = g0; v.n = n;
}

/* lane `l` of v <- the wave-uniform value s.  The s_nop covers gfx950's wait states between a
 * VALU instruction that writes an SGPR (the v_cmp of a ballot) and a VALU instruction reading
 * it, which the compiler cannot insert across an asm statement. */
#define write_lane(v, s, l) asm volatile("s_nop 1\n\tv_writelane_b32 %0, %1, " #l : "+v"(v) : "s"(s))

/* flag mask of one row; e_prev / e_next = the dwords just outside the row */
__device__ __forceinline__ uint64_t flag_row(const u32x4& q, uint32_t e_prev, uint32_t e_next, bool edge_tile,
                                             uint64_t g, uint64_t n, bool& mine)
{
    const uint32_t xp = from_prev_lane(q.w, e_prev);
    const uint32_t xn = from_next_lane(q.x, e_next);
    bool f = chunk_flag(xp, q.x, q.y, q.z, q.w, xn);
    if (edge_tile) f = f || (g < n && n < g + 16);          /* the chunk cut by the stream end is always an element */
    mine = f;
    return __ballot(f);
}

__global__ __launch_bounds__(k4Threads, 2)
void k_scan_extract4(const uint8_t* __restrict__ stream, uint64_t n, uint64_t num_tiles,
                     hbs_nal_entry* __restrict__ index, uint64_t index_cap,
                     uint8_t* __restrict__ rbsp, uint64_t rbsp_cap,
                     unsigned long long* __restrict__ desc, RunHeader* __restrict__ hdr, const uint8_t* __restrict__ tail)
{
    __shared__ Lds4 l;
    const int tid0 = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    EmitTarget tgt;
    tgt.index = index; tgt.index_cap = index_cap; tgt.hdr = hdr;
    if (tid0 == 0) l.ticket = atomicAdd(&hdr->ticket, 1u);
    __syncthreads();
    HBS4_T_DECL

    for (;;) {
        int tid = launder_lane(tid0);
        int lane = tid & 63;
        const uint64_t tile = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)l.ticket);
        if (tile >= num_tiles) break;
        const uint64_t base = tile * (uint64_t)k4TileBytes;
        const uint64_t tile_end = base + (uint64_t)k4TileBytes;
        const uint64_t wseg = base + (uint64_t)(wv * k4WaveBytes);
        const bool edge_tile = tile_end + 4 > n;          /* some chunk of the tile may be cut by the stream end */

        /* the last tile comes from its padded copy: tail[kTailLead + i] = stream[base + i] */
        const uint8_t* const src = (tile == num_tiles - 1)
            ? reinterpret_cast<const uint8_t*>(reinterpret_cast<uintptr_t>(tail) + (uintptr_t)k4TailLead - (uintptr_t)base) : stream;
        /* until the tile's aggregate is out, this workgroup is what its successors wait for */
        __builtin_amdgcn_s_setprio(3);
        RowRegs R;
        fetch_row_regs(R, src, wseg, lane);
        R.before = (wseg >= 4) ? *reinterpret_cast<const uint32_t*>(src + wseg - 4) : 0xFFFFFFFFu;
        R.before2 = (wseg >= 8) ? *reinterpret_cast<const uint32_t*>(src + wseg - 8) : 0xFFFFFFFFu;
        R.after = (tile == num_tiles - 1 || wv != k4Waves - 1) ? *reinterpret_cast<const uint32_t*>(src + wseg + k4WaveBytes)
                                                               : load_dword_guarded(stream, (int64_t)(wseg + k4WaveBytes), n);
        HBS4_T_MARK(0)

        /* ---- 1. flag masks of my rows --------------------------------------------------- */
        /* Straight-line over the named rows; a row's 64-bit mask is stashed in lane r of
         * fm_lo/fm_hi (v_writelane), so nothing per-row lives in SGPRs or LDS. */
        uint32_t myf = 0;                  /* bit r: my chunk of row r is an element */
        uint32_t fm_lo = 0, fm_hi = 0;     /* lane r: flag mask of my row r          */
        uint32_t wslot = 0;                /* elements of this wavefront so far      */
        {
            uint32_t e_prev = R.before;
#define HBS_FLAG_BODY(r, e_prev_z_expr, e_next_expr) { \
                const uint32_t xp = from_prev_lane(R.q##r.w, e_prev); \
                const uint32_t xn = from_next_lane(R.q##r.x, (e_next_expr)); \
                const bool f = chunk_flag(xp, R.q##r.x, R.q##r.y, R.q##r.z, R.q##r.w, xn); \
                const uint64_t fmask = __ballot(f); \
                myf |= f ? (1u << r) : 0u; \
                write_lane(fm_lo, (uint32_t)fmask, r); \
                write_lane(fm_hi, (uint32_t)(fmask >> 32), r); \
                if (fmask != 0) {        /* rare: leave the chunk's surroundings for its element thread */ \
                    const uint32_t xpp = from_prev_lane(R.q##r.z, (e_prev_z_expr)); \
                    const uint32_t slot = wslot + lanes_below(fmask); \
                    if (f && slot < (uint32_t)kDepCap) { \
                        Deposit d; \
                        d.xpp = xpp; d.xp = xp; d.x0 = R.q##r.x; d.x1 = R.q##r.y; d.x2 = R.q##r.z; d.x3 = R.q##r.w; d.xn = xn; \
                        d.chunk = (uint32_t)(64 * (k4Rows * wv + r) + lane); \
                        l.dep[wv][slot] = d; \
                    } \
                    wslot += (uint32_t)__builtin_popcountll(fmask); \
                } \
                e_prev = (uint32_t)__builtin_amdgcn_readlane((int)R.q##r.w, 63); }
#define HBS_FLAG(rp, r, rn) HBS_FLAG_BODY(r, (uint32_t)__builtin_amdgcn_readlane((int)R.q##rp.z, 63), (uint32_t)__builtin_amdgcn_readlane((int)R.q##rn.x, 0))
            HBS_FLAG_BODY(0, R.before2, (uint32_t)__builtin_amdgcn_readlane((int)R.q1.x, 0))
            HBS_ROW_TRIPLES(HBS_FLAG)
            HBS_FLAG_BODY(31, (uint32_t)__builtin_amdgcn_readlane((int)R.q30.z, 63), R.after)
#undef HBS_FLAG
#undef HBS_FLAG_BODY
        }
        if (edge_tile && (n & 15ull) != 0 && n > wseg && n < wseg + (uint64_t)k4WaveBytes) {
            /* the chunk cut by the stream end is always an element */
            const uint32_t cut = (uint32_t)(n - wseg) >> 4;            /* its chunk number in my segment */
            const int cr = (int)(cut >> 6), cl = (int)(cut & 63u);
            if (lane == cl && !((myf >> cr) & 1u)) {
                /* not flagged by its bytes: nothing was deposited for it; it is the wavefront's last
                 * element, and its thread must read the stream itself */
                myf |= 1u << cr;
                if (wslot < (uint32_t)kDepCap) l.dep[wv][wslot].chunk = 0xFFFFFFFFu;
            }
            if (lane == cr) { if (cl < 32) fm_lo |= 1u << cl; else fm_hi |= 1u << (cl - 32); }
        }
        if (lane < k4Rows) l.row_cnt[k4Rows * wv + lane] = (uint32_t)__builtin_popcount(fm_lo) + (uint32_t)__builtin_popcount(fm_hi);
        uint32_t rowmask = (uint32_t)__ballot(lane < k4Rows && (fm_lo | fm_hi) != 0u);   /* my rows that hold an element */
        __syncthreads();
        tid = launder_lane(tid0); lane = tid & 63;
        /* lane j: flagged chunks of the tile in front of rows 2j and 2j+1 */
        uint32_t pre_even, pre_odd, nflag;
        {
            const uint32_t c0 = l.row_cnt[2 * lane], c1 = l.row_cnt[2 * lane + 1];
            const uint32_t inc = wave_incl_scan32(c0 + c1, lane);
            pre_even = inc - (c0 + c1);
            pre_odd = pre_even + c0;
            nflag = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        }
        /* elements in front of wavefronts 1..3 (their first rows are even: lane 16 w of pre_even) */
        const uint32_t wb1 = (uint32_t)__builtin_amdgcn_readlane((int)pre_even, 1 * k4Rows / 2);
        const uint32_t wb2 = (uint32_t)__builtin_amdgcn_readlane((int)pre_even, 2 * k4Rows / 2);
        const uint32_t wb3 = (uint32_t)__builtin_amdgcn_readlane((int)pre_even, 3 * k4Rows / 2);
        static_assert(k4Waves == 4 && k4Rows % 2 == 0, "wave bases are read from pre_even");
        /* readlanes: only in wave-uniform control flow */
#define HBS_ROW_PRE(r) ((uint32_t)__builtin_amdgcn_readlane((int)(((r) & 1) ? pre_odd : pre_even), (k4Rows * wv + (r)) >> 1))
#define HBS_ROW_FM(r) (((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)fm_hi, (r)) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)fm_lo, (r)))
        for (uint32_t rm = rowmask; rm != 0u; rm &= rm - 1u) {
            const int r = __builtin_ctz(rm);
            const uint32_t rp = HBS_ROW_PRE(r);
            const uint64_t f = HBS_ROW_FM(r);
            if ((myf >> r) & 1u)
                l.list[rp + lanes_below(f)] = (uint16_t)(64 * (k4Rows * wv + r) + lane);
        }
        __syncthreads();
        HBS4_DBG(if (tile == 0) { if (tid == 0) { g_dbg4[0] = nflag; } if (tid < 128) g_dbg4[16 + tid] = l.row_cnt[tid]; for (int i = tid; i < 1024; i += k4Threads) g_dbg4[256 + i] = l.list[i]; g_dbg4[2048 + tid] = myf; })
        HBS4_T_MARK(1)

        /* ---- 2..4: phase 0 = aggregate + look-back, phase 1 = emit + copy ----------------- */
        const uint32_t npass = (nflag + (uint32_t)k4ElemPass - 1u) / (uint32_t)k4ElemPass;
        const bool parked = npass == 1u;           /* elements survive the look-back in LDS */
        TileAgg tagg = agg_identity();
        Prefix excl;
        excl.kept = 0; excl.nals = 0; excl.inside = 0;
        bool can_store = false;
        uint8_t* out = rbsp;
        uint32_t next_ticket = 0;
#pragma unroll 1
        for (int phase = 0; phase < 2; ++phase) {
            TileAgg acc = agg_identity();
            const uint32_t np = (phase == 1 && npass == 0u) ? 1u : npass;
#pragma unroll 1
            for (uint32_t p = 0; p < np; ++p) {
                tid = launder_lane(tid0); lane = tid & 63;
                const uint32_t pbase = p * (uint32_t)k4ElemPass;
                const uint32_t i = pbase + (uint32_t)tid;
                const bool wave_has = pbase + 64u * (uint32_t)wv < nflag;
                const bool active = i < nflag;
                ElemView v; ChunkMarks m; BlockSum s;
                TileAgg e = agg_identity();        /* tile aggregate in front of my element's gap */
                uint32_t gap = 0, c = 0;
                v.stream = stream; v.n = n;
                if (phase == 0 || !parked) {
                    TileAgg ea = agg_identity();
                    if (wave_has) {
                        if (active) {
                            c = l.list[i];
                            const uint64_t prev_end = (i > 0) ? base + 16ull * ((uint32_t)l.list[i - 1] + 1u) : base;
                            /* which wavefront flagged it, and as its how-manieth element */
                            const uint32_t ew = (i >= wb1 ? 1u : 0u) + (i >= wb2 ? 1u : 0u) + (i >= wb3 ? 1u : 0u);
                            const uint32_t ej = i - (ew == 0u ? 0u : ew == 1u ? wb1 : ew == 2u ? wb2 : wb3);
                            Deposit d;
                            d.chunk = 0xFFFFFFFFu;
                            if (ej < (uint32_t)kDepCap) d = l.dep[ew][ej];
                            if (d.chunk == c) {
                                v.xpp = d.xpp; v.xp = d.xp; v.x0 = d.x0; v.x1 = d.x1; v.x2 = d.x2; v.x3 = d.x3; v.xn = d.xn;
                                v.stream = src; v.g0 = base + 16ull * c; v.n = n;
                            } else {
                                elem_load(v, src, base + 16ull * c, n, tile == num_tiles - 1);
                            }
                            elem_walk(v, m, s);
                            gap = span_bytes(prev_end, v.g0, n);
                            ea = elem_agg(gap, s);
                        }
                        ea = wave_scan_combine(ea, lane);
                    }
                    if (lane == 63) l.wtot[wv] = ea;
                    __syncthreads();
                    TileAgg before = acc;
#pragma unroll
                    for (int w = 0; w < k4Waves; ++w) {
                        const TileAgg a = l.wtot[w];
                        if (w < wv) before = combine(before, a);
                        acc = combine(acc, a);
                    }
                    if (wave_has) {
                        TileAgg up = agg_shfl_up(ea, 1);
                        if (lane == 0) up = agg_identity();
                        e = combine(before, up);
                    }
                    if (phase == 0) {
                        if (parked && active) {
                            ElemState st;
                            const ElemPacked pk = elem_pack(m, s);
                            st.xpp = v.xpp; st.xp = v.xp; st.x0 = v.x0; st.x1 = v.x1; st.x2 = v.x2; st.x3 = v.x3; st.xn = v.xn;
                            st.pa = pk.a; st.pb = pk.b; st.pc = pk.c; st.gap = gap; st.chunk = c; st.e = e;
                            l.est[tid] = st;
                        }
                        if (npass > 1u) __syncthreads();
                        continue;
                    }
                } else if (active) {
                    const ElemState st = l.est[tid];
                    ElemPacked pk;
                    pk.a = st.pa; pk.b = st.pb; pk.c = st.pc;
                    elem_unpack(pk, m, s);
                    v.xpp = st.xpp; v.xp = st.xp; v.x0 = st.x0; v.x1 = st.x1; v.x2 = st.x2; v.x3 = st.x3; v.xn = st.xn;
                    gap = st.gap; c = st.chunk; e = st.e;
                    v.g0 = base + 16ull * c;
                }
                if (active) {
                    const ElemStart st = elem_start(e, gap, excl.inside);
                    const uint32_t keep = (uint32_t)emit_block_t<kChunk, ElemView, uint32_t>(v, 0, v.g0, m, st.inside, excl.nals + e.cnt,
                                                                                  excl.kept + st.kept, tgt);
                    const uint32_t nk = (uint32_t)__builtin_popcount(keep);
                    if (can_store && keep != 0u) {
                        if (keep == 0xFFFFu) {
                            u32x4 q; q.x = v.x0; q.y = v.x1; q.z = v.x2; q.w = v.x3;
                            reinterpret_cast<Unaligned16_3*>(out + st.kept)->v = q;
                        } else {
                            uint64_t lo, hi;
                            const uint32_t cn = compact_chunk_regs(v.x0, v.x1, v.x2, v.x3, keep, lo, hi);
                            store_pieces(out + st.kept, lo, hi, cn);
                        }
                    }
                    const bool after = (s.last != kKindNone) ? (s.last == kKindStart) : st.inside;
                    l.seg[tid + 1] = seg_pack((int32_t)c, st.kept + nk, after);
                }
                __syncthreads();
                HBS4_T_MARK(4)

                /* unflagged chunk with k elements in front of it: served by the pass that holds
                 * element k-1 (k = 0: the tile start, pass 0) */
                tid = launder_lane(tid0); lane = tid & 63;
                /* the next tile is claimed as late as its round trip can still hide behind the copy:
                 * tiles are looked back in ticket order, and a ticket taken long before its tile
                 * is started makes every successor wait */
                if (p + 1 == np && tid == 0) next_ticket = atomicAdd(&hdr->ticket, 1u);
                if (can_store) {
                    const uint32_t cc0 = (uint32_t)(64 * k4Rows * wv + lane);
                    const uint32_t whole = (uint32_t)(span_bytes(base, tile_end, n) >> 4);   /* chunks of the tile that are complete */
                    /* straight-line over the named rows: k from the stashed mask (two readlanes
                     * and mbcnt), one LDS read, one store */
#define HBS_COPY(r) { \
                        const uint32_t k = HBS_ROW_PRE(r) + lanes_below(HBS_ROW_FM(r)); \
                        const uint32_t cc = cc0 + 64u * r; \
                        const bool served = p == 0u ? k <= (uint32_t)k4ElemPass : (k > pbase && k <= pbase + (uint32_t)k4ElemPass); \
                        if (!((myf >> r) & 1u) && served && cc < whole) { \
                            const uint32_t w = l.seg[k - pbase]; \
                            if (seg_inside(w)) reinterpret_cast<Unaligned16_3*>(out + (int64_t)(seg_bias(w) + (int32_t)(16u * cc)))->v = R.q##r; \
                        } }
                    HBS_ROWS(HBS_COPY)
#undef HBS_COPY
                }
                if (p + 1 == np && tid == 0) l.ticket = next_ticket;
                __syncthreads();
                HBS4_T_MARK(5)
            }
            if (phase == 1) break;

            const uint64_t last_end = (nflag > 0) ? base + 16ull * ((uint32_t)l.list[nflag - 1] + 1u) : base;
            tagg = combine(acc, gap_agg(span_bytes(last_end, tile_end, n)));
            HBS4_T_MARK(2)
            if (wv != 0) __builtin_amdgcn_s_setprio(0);

            /* ---- 3. look-back (wavefront 0) ---------------------------------------------- */
            if (wv == 0) {
                Prefix ex;
                uint32_t it, stl;
                const bool ok = look_back4(desc, tile, tagg, hdr, lane, ex, it, stl);
                HBS4_T_COUNT(7, ((unsigned long long)stl << 32) | it)
                __builtin_amdgcn_s_setprio(0);
                if (lane == 0) {
                    l.ex_kept = ex.kept; l.ex_nals = ex.nals; l.ex_inside = ex.inside; l.ex_ok = ok ? 1u : 0u;
                    l.seg[0] = seg_pack(-1, 0u, ex.inside != 0u);
                }
            }
            __syncthreads();
            if (l.ex_ok == 0u) return;
            {
                Prefix ex;
                ex.kept = l.ex_kept; ex.nals = l.ex_nals; ex.inside = l.ex_inside;
                excl = prefix_uniform4(ex);
            }
            HBS4_T_MARK(3)
            if (tid == 0 && tile == num_tiles - 1) {
                const Prefix incl = fold(excl, tagg);
                hdr->final_kept = incl.kept; hdr->final_nals = incl.nals; hdr->final_inside = incl.inside;
            }
            const uint32_t tile_kept = tagg.known + (excl.inside ? tagg.sig : 0u);
            can_store = rbsp != nullptr && excl.kept + tile_kept <= rbsp_cap;
            if (rbsp != nullptr && !can_store && tid == 0) atomicMax(&hdr->error, (uint32_t)(-HBS_E_CAPACITY));
            out = rbsp + excl.kept;
        }
#undef HBS_ROW_PRE
#undef HBS_ROW_FM
    }
    HBS4_T_FLUSH
}

#ifdef HBS_PHASE_TIMING
extern "C" int hbs_debug_dump4(uint32_t* host_out /* [4096] */)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dbg4), sizeof(uint32_t) * 4096);
}
extern "C" int hbs_debug_phase_cycles4(unsigned long long* host_out /* [1024][8] */)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_phase_cycles4), sizeof(unsigned long long) * 1024 * 8);
}
#endif

int scan4_tile_bytes() { return k4TileBytes; }
int scan4_tail_bytes() { return k4TailBytes; }

/* tail[k4TailLead + i] = stream[last_base + i] for i in [-k4TailLead, tile + pad), 0xFF where the
 * stream has no byte: the main kernel reads the stream's last tile from here, unguarded */
__global__ void k_prepare_tail4(const uint8_t* __restrict__ stream, uint64_t n, uint8_t* __restrict__ tail)
{
    if (n == 0) return;
    const uint64_t last_base = ((n - 1) / (uint64_t)k4TileBytes) * (uint64_t)k4TileBytes;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < (uint32_t)k4TailBytes; i += gridDim.x * blockDim.x) {
        const int64_t q = (int64_t)last_base + (int64_t)i - k4TailLead;
        tail[i] = (q >= 0 && (uint64_t)q < n) ? stream[q] : (uint8_t)0xFF;
    }
}

void launch_scan4_prepare_tail(const ScanArgs& a, hipStream_t st)
{
    if (a.n) k_prepare_tail4<<<dim3(32), dim3(256), 0, st>>>(a.stream, a.n, a.tail);
}

int scan4_grid_blocks(int device, int* blocks_per_cu_out)
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return -1;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_scan_extract4, k4Threads, 0) != hipSuccess) return -1;
    if (per_cu < 1) per_cu = 1;
    if (blocks_per_cu_out) *blocks_per_cu_out = per_cu;
    return prop.multiProcessorCount * per_cu;
}

void launch_scan_extract4_kernel(const ScanArgs& a, uint64_t num_tiles, hipStream_t st)
{
    uint64_t grid = (uint64_t)a.grid_blocks;
    if (grid > num_tiles) grid = num_tiles;
    k_scan_extract4<<<dim3((unsigned)grid), dim3(k4Threads), 0, st>>>(
        a.stream, a.n, num_tiles, a.index, a.index_cap, a.rbsp, a.rbsp_cap, a.desc, a.hdr, a.tail);
}

} // namespace hbs
