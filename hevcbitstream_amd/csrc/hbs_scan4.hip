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
#else
#define HBS4_T_DECL
#define HBS4_T_MARK(i)
#define HBS4_T_COUNT(i, v)
#define HBS4_T_FLUSH
#endif

constexpr int k4TileRows = k4Waves * k3Rows;                /* rows of 1 KiB per tile */

struct Lds4 {
    unsigned long long fm[k4TileRows];     /* flag mask of each row (bit l: chunk l of the row is an element) */
    uint32_t row_cnt[64];                  /* flagged chunks per row (entries >= k4TileRows stay 0)          */
    uint16_t list[k4ChunksPerTile];        /* flagged chunks of the tile, in stream order  */
    uint32_t seg[k4ElemPass + 1];          /* segment words: [0] tile start, [i+1] element i of the pass */
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
                __builtin_amdgcn_s_sleep(2);
                continue;
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

/* the six dwords around chunk g0 of the stream, for an element */
__device__ __forceinline__ void elem_load(RegView& v, const uint8_t* __restrict__ stream, uint64_t g0, uint64_t n)
{
    const u32x4 q = load_chunk_guarded(stream, g0, n);
    v.xp = load_dword_guarded(stream, (int64_t)g0 - 4, n);
    v.xn = load_dword_guarded(stream, (int64_t)g0 + 16, n);
    v.x0 = q.x; v.x1 = q.y; v.x2 = q.z; v.x3 = q.w;
    v.stream = stream; v.g0 = g0; v.n = n;
}

__global__ __launch_bounds__(k4Threads, 4)
void k_scan_extract4(const uint8_t* __restrict__ stream, uint64_t n, uint64_t num_tiles,
                     hbs_nal_entry* __restrict__ index, uint64_t index_cap,
                     uint8_t* __restrict__ rbsp, uint64_t rbsp_cap,
                     unsigned long long* __restrict__ desc, RunHeader* __restrict__ hdr)
{
    __shared__ Lds4 l;
    const int tid0 = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    EmitTarget tgt;
    tgt.index = index; tgt.index_cap = index_cap; tgt.hdr = hdr;
    if (tid0 < 64) l.row_cnt[tid0] = 0;
    HBS4_T_DECL

    for (;;) {
        int tid = launder_lane(tid0);
        int lane = tid & 63;
        if (tid == 0) l.ticket = atomicAdd(&hdr->ticket, 1u);
        __syncthreads();
        const uint64_t tile = (uint64_t)l.ticket;
        if (tile >= num_tiles) break;
        const uint64_t base = tile * (uint64_t)k4TileBytes;
        const uint64_t tile_end = base + (uint64_t)k4TileBytes;
        const uint64_t wseg = base + (uint64_t)(wv * k3WaveBytes);
        const bool edge_tile = tile_end + 4 > n;          /* some chunk of the tile may be cut by the stream end */

        Rows R;
        fetch_rows(R, stream, wseg, n, lane);
        HBS4_T_MARK(0)

        /* ---- 1. flag masks of my 8 rows ------------------------------------------------- */
        uint32_t myf = 0;                  /* bit r: my chunk of row r is an element */
        {
            uint32_t e_prev = R.before;
            const uint32_t next_x_0 = R.q1.x, next_x_1 = R.q2.x, next_x_2 = R.q3.x, next_x_3 = R.q4.x;
            const uint32_t next_x_4 = R.q5.x, next_x_5 = R.q6.x, next_x_6 = R.q7.x, next_x_7 = 0u;
#define HBS_FLAG(r) { \
                const u32x4 q = R.q##r; \
                const uint32_t e_next = (r == k3Rows - 1) ? R.after : (uint32_t)__builtin_amdgcn_readlane((int)next_x_##r, 0); \
                const uint32_t xp = from_prev_lane(q.w, e_prev); \
                const uint32_t xn = from_next_lane(q.x, e_next); \
                bool f = chunk_flag(xp, q.x, q.y, q.z, q.w, xn); \
                if (edge_tile) { \
                    const uint64_t g = wseg + (uint64_t)(r * k3RowBytes + 16 * lane); \
                    f = f || (g < n && n < g + 16); \
                } \
                const uint64_t fmask = __ballot(f); \
                myf |= f ? (1u << r) : 0u; \
                if (lane == 0) { l.fm[k3Rows * wv + r] = fmask; l.row_cnt[k3Rows * wv + r] = (uint32_t)__builtin_popcountll(fmask); } \
                e_prev = (uint32_t)__builtin_amdgcn_readlane((int)q.w, 63); }
            HBS_REP8(HBS_FLAG)
#undef HBS_FLAG
            (void)next_x_7;
        }
        __syncthreads();
        tid = launder_lane(tid0); lane = tid & 63;
        /* lane j: flagged chunks of the tile in front of row j */
        uint32_t row_pre;
        uint32_t nflag;
        {
            const uint32_t cnt = l.row_cnt[lane];
            const uint32_t inc = wave_incl_scan32(cnt, lane);
            row_pre = inc - cnt;
            nflag = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        }
        if (__ballot(myf != 0u) != 0ull) {
#pragma unroll
            for (int r = 0; r < k3Rows; ++r) {
                const uint32_t rp = (uint32_t)__builtin_amdgcn_readlane((int)row_pre, k3Rows * wv + r);
                if ((myf >> r) & 1u)
                    l.list[rp + lanes_below(l.fm[k3Rows * wv + r])] = (uint16_t)(64 * (k3Rows * wv + r) + lane);
            }
        }
        __syncthreads();
        HBS4_T_MARK(1)

        /* ---- 2..4: phase 0 = aggregate + look-back, phase 1 = emit + copy ----------------- */
        const uint32_t npass = (nflag + (uint32_t)k4ElemPass - 1u) / (uint32_t)k4ElemPass;
        TileAgg tagg = agg_identity();
        Prefix excl;
        excl.kept = 0; excl.nals = 0; excl.inside = 0;
        bool can_store = false;
        uint8_t* out = rbsp;
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
                TileAgg ea = agg_identity();
                RegView v; BlockMarks m; BlockSum s;
                uint32_t gap = 0, c = 0;
                if (wave_has) {
                    if (i < nflag) {
                        c = l.list[i];
                        const uint64_t prev_end = (i > 0) ? base + 16ull * ((uint32_t)l.list[i - 1] + 1u) : base;
                        elem_load(v, stream, base + 16ull * c, n);
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
                if (phase == 0) {
                    if (npass > 1u) __syncthreads();
                    continue;
                }
                if (wave_has) {
                    TileAgg up = agg_shfl_up(ea, 1);
                    if (lane == 0) up = agg_identity();
                    if (i < nflag) {
                        const TileAgg e = combine(before, up);
                        const ElemStart st = elem_start(e, gap, excl.inside);
                        const uint32_t keep = (uint32_t)emit_block_t<kChunk, RegView>(v, 0, v.g0, m, st.inside, excl.nals + e.cnt,
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
                }
                __syncthreads();
                HBS4_T_MARK(4)

                /* unflagged chunk with k elements in front of it: served by the pass that holds
                 * element k-1 (k = 0: the tile start, pass 0) */
                tid = launder_lane(tid0); lane = tid & 63;
                if (can_store) {
#define HBS_COPY(r) { \
                        const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)row_pre, k3Rows * wv + r) + lanes_below(l.fm[k3Rows * wv + r]); \
                        const uint32_t kp = k ? (k - 1u) / (uint32_t)k4ElemPass : 0u; \
                        const uint32_t cc = (uint32_t)(64 * (k3Rows * wv + r) + lane); \
                        const bool plain = !((myf >> r) & 1u) && kp == p && (!edge_tile || base + 16ull * cc + 16ull <= n); \
                        if (plain) { \
                            const uint32_t w = l.seg[k - pbase]; \
                            if (seg_inside(w)) reinterpret_cast<Unaligned16_3*>(out + (int64_t)(seg_bias(w) + (int32_t)(16u * cc)))->v = R.q##r; \
                        } }
                    HBS_REP8(HBS_COPY)
#undef HBS_COPY
                }
                __syncthreads();
                HBS4_T_MARK(5)
            }
            if (phase == 1) break;

            const uint64_t last_end = (nflag > 0) ? base + 16ull * ((uint32_t)l.list[nflag - 1] + 1u) : base;
            tagg = combine(acc, gap_agg(span_bytes(last_end, tile_end, n)));
            HBS4_T_MARK(2)

            /* ---- 3. look-back (wavefront 0) ---------------------------------------------- */
            if (wv == 0) {
                Prefix ex;
                uint32_t it, stl;
                const bool ok = look_back4(desc, tile, tagg, hdr, lane, ex, it, stl);
                HBS4_T_COUNT(7, ((unsigned long long)stl << 32) | it)
                if (lane == 0) {
                    l.ex_kept = ex.kept; l.ex_nals = ex.nals; l.ex_inside = ex.inside; l.ex_ok = ok ? 1u : 0u;
                    l.seg[0] = seg_pack(-1, 0u, ex.inside != 0u);
                }
            }
            __syncthreads();
            if (l.ex_ok == 0u) return;
            excl.kept = l.ex_kept; excl.nals = l.ex_nals; excl.inside = l.ex_inside;
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
    }
    HBS4_T_FLUSH
}

#ifdef HBS_PHASE_TIMING
extern "C" int hbs_debug_phase_cycles4(unsigned long long* host_out /* [1024][8] */)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_phase_cycles4), sizeof(unsigned long long) * 1024 * 8);
}
#endif

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
        a.stream, a.n, num_tiles, a.index, a.index_cap, a.rbsp, a.rbsp_cap, a.desc, a.hdr);
}

} // namespace hbs
