/*
 * hbs_scan3.hip -- K12, register-resident form: fused start-code scan + NAL index
 * + RBSP extraction with the tile held in VGPRs instead of an LDS image.
 *
 * Same contract and same tile algebra as hbs_scan.hip (reference loop
 * find_nal_unit + nal_to_rbsp, h264_nal.c:38-76 / :147-200 driven as in
 * hevc_analyze.c:135-177), different mapping to the machine:
 *
 *   - a workgroup is 8 wavefronts; a tile is 64 KiB; wavefront w owns the
 *     CONTIGUOUS 8 KiB [8w, 8w+8) KiB of it as 8 rows of 1 KiB, one
 *     global_load_dwordx4 per row (lane l = 16 bytes at 16*l: coalesced).  The
 *     next tile's rows are loaded into each row's registers as soon as the emit
 *     pass has written that row out;
 *   - a lane sees the dword in front of / behind its chunk through DPP
 *     wave_shr/wave_shl (row edges: v_readlane of the neighbouring row; segment
 *     edges: one scalar-like load);
 *   - almost every row has no two adjacent zero bytes at all (5 VALU ops per
 *     dword decide that): its summary is a constant and its bytes are copied
 *     with one byte-aligned 16-byte store per lane.  Rows that do have some get
 *     the exact per-chunk classification (hbs_chunk.h) and a wave scan;
 *   - state runs row to row in SCALAR registers; the only workgroup exchange is
 *     the 8 wave aggregates (one barrier) and the decoupled look-back.
 *
 * No LDS image, no staging pass, no per-thread search; LDS holds ~1 KiB.
 */
#include <hip/hip_runtime.h>
#include "hbs_wave.h"
#include "hbs_scan.h"

namespace hbs {

#ifdef HBS_PHASE_TIMING
__device__ unsigned long long g_phase_cycles3[1024][8];
#define HBS3_T_DECL unsigned long long t_prev = __builtin_amdgcn_s_memtime(), t_acc[8] = {0,0,0,0,0,0,0,0};
#define HBS3_T_MARK(i) { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); t_acc[i] += t_now - t_prev; t_prev = t_now; }
#define HBS3_T_COUNT(i, v) { t_acc[i] += (v); }
#define HBS3_T_FLUSH if (threadIdx.x == 0 && blockIdx.x < 1024) { for (int i = 0; i < 8; ++i) g_phase_cycles3[blockIdx.x][i] = t_acc[i]; }
#else
#define HBS3_T_DECL
#define HBS3_T_MARK(i)
#define HBS3_T_COUNT(i, v)
#define HBS3_T_FLUSH
#endif

struct WaveSlot3 {
    uint32_t status, abort;
    TileAgg win;
    uint64_t pre_kept, pre_nals;
    uint32_t pre_inside, pad;
};

struct Lds3 {
    TileAgg wave_agg[k3Waves];
    WaveSlot3 lb[2][4];
};

/* what a lane knows about its chunk inside a non-trivial row, after the row scan */
struct LanePos {
    uint32_t in_state;     /* 0 outside, 1 inside, 2 = state at the row's first byte */
    uint32_t known, sig;   /* kept bytes of the row in front of the chunk            */
    uint32_t cnt;          /* NAL starts of the row in front of the chunk            */
};

/*
 * Classify one row.  Returns false for a plain row (no pattern can end in it,
 * fully inside the stream): its aggregate is {0, 0, 1024, none}.  Otherwise
 * fills the per-lane marks/summary, the row aggregate and the lane positions.
 */
__device__ __forceinline__ bool row_classify(const u32x4& q, uint32_t edge_prev, uint32_t edge_next,
                                             const uint8_t* __restrict__ stream, uint64_t row_g0, uint64_t n, int lane,
                                             RegView& v, BlockMarks& m, TileAgg& agg, LanePos& pos)
{
    const uint32_t xp = from_prev_lane(q.w, edge_prev);
    const uint32_t xn = from_next_lane(q.x, edge_next);
    const bool maybe = chunk_maybe_pattern(xp, q.x, q.y, q.z, q.w, xn);
    const bool whole = row_g0 + k3RowBytes <= n;
    if (__ballot(maybe) == 0 && whole) return false;

    v.xp = xp; v.x0 = q.x; v.x1 = q.y; v.x2 = q.z; v.x3 = q.w; v.xn = xn;
    v.stream = stream; v.g0 = row_g0 + 16ull * (uint32_t)lane; v.n = n;
    BlockSum s;
    if (maybe || !whole) {
        const uint32_t pats = maybe ? chunk_patterns(xp, q.x, q.y, q.z, q.w, xn) : 0u;
        walk_block_t<kChunk, RegView>(v, 0, v.g0, n, pats & 0xFFFFu, pats >> 16, m, s);
    } else {
        m.cand = 0xFFFFull; m.ev = m.ev_start = m.err = 0;
        s.cnt = 0; s.known = 0; s.carry = 16; s.last = kKindNone;
    }

    /* row scan (the block_scan of hbs_scan.hip, one wavefront wide) */
    const uint64_t m_ev = __ballot(s.last != kKindNone);
    if (m_ev == 0) {
        const uint32_t inc = wave_incl_scan32(s.carry, lane);
        pos.in_state = 2u; pos.known = 0; pos.sig = inc - s.carry; pos.cnt = 0;
        agg.cnt = 0; agg.known = 0; agg.sig = __builtin_amdgcn_readlane(inc, 63); agg.last = kKindNone;
    } else {
        const uint64_t m_st = __ballot(s.last == kKindStart);
        const uint64_t lower = m_ev & ((1ull << lane) - 1ull);
        uint32_t st = 2u;
        if (lower != 0) st = (uint32_t)((m_st >> (63 - __builtin_clzll(lower))) & 1ull);
        const uint32_t known = s.known + (st == 1u ? s.carry : 0u);
        const uint32_t sig = (st == 2u) ? s.carry : 0u;
        const uint32_t packed = known | (sig << 11) | (s.cnt << 22);           /* <= 1024, <= 1024, <= 341 per row */
        const uint32_t inc = wave_incl_scan32(packed, lane);
        const uint32_t ex = inc - packed;
        pos.in_state = st; pos.known = ex & 0x7FFu; pos.sig = (ex >> 11) & 0x7FFu; pos.cnt = ex >> 22;
        const uint32_t tot = __builtin_amdgcn_readlane(inc, 63);
        agg.known = tot & 0x7FFu; agg.sig = (tot >> 11) & 0x7FFu; agg.cnt = tot >> 22;
        agg.last = ((m_st >> (63 - __builtin_clzll(m_ev))) & 1ull) ? kKindStart : kKindStop;
    }
    return true;
}

__device__ __forceinline__ bool look_back3(Lds3& l, unsigned long long* desc, uint64_t tile, const TileAgg& mine,
                                           RunHeader* hdr, int tid, Prefix& excl)
{
    const int lane = tid & 63, wv = tid >> 6;
    bool ok = true;
    if (tile == 0) {
        excl.kept = 0; excl.nals = 0; excl.inside = 0;
    } else {
        if (tid == 0) {
            st_desc3(&desc[2 * tile], pack_agg0(mine));
            st_desc3(&desc[2 * tile + 1], pack_agg1(mine));
        }
        TileAgg acc = {0u, 0u, 0u, kKindNone};
        int64_t win_hi = (int64_t)tile - 1;
        uint32_t spins = 0, par = 0;
        for (;;) {
            if (wv < 4) {
                const int64_t t = win_hi - (64 * wv + lane);
                uint64_t w0 = kDescPrefix, w1 = kDescPrefix;
                if (t >= 0) {
                    w0 = ld_desc3(&desc[2 * t]);
                    w1 = ld_desc3(&desc[2 * t + 1]);
                }
                const uint32_t s0 = (uint32_t)(w0 & 3u), s1 = (uint32_t)(w1 & 3u);
                const bool ready = (s0 == s1) && (s0 != kDescEmpty);
                const bool is_pre = ready && (s0 == kDescPrefix);
                const uint64_t m_pre = __ballot(is_pre);
                const uint64_t m_ready = __ballot(ready);
                const int lstar = m_pre ? (int)__builtin_ctzll(m_pre) : 64;
                const uint64_t need = (lstar >= 64) ? ~0ull : ((1ull << lstar) - 1ull);
                const bool win_ok = (m_ready & need) == need;
                const TileAgg win = window_fold3(unpack_agg(w0, w1), lstar, lane);
                const Prefix p = unpack_pre(w0, w1);
                if (lane == (lstar & 63)) {
                    WaveSlot3& sl = l.lb[par][wv];
                    sl.status = win_ok ? (lstar < 64 ? 2u : 1u) : 0u;
                    sl.abort = (wv == 0 && (spins & 63u) == 63u) ? __hip_atomic_load(&hdr->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                    sl.win = win;
                    sl.pre_kept = p.kept; sl.pre_nals = p.nals; sl.pre_inside = p.inside;
                }
            }
            __syncthreads();
            bool done = false, stall = false;
            TileAgg a2 = acc;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                if (done || stall) continue;
                const WaveSlot3& sl = l.lb[par][w];
                if (sl.status == 0u) { stall = true; continue; }
                a2 = combine(sl.win, a2);
                if (sl.status == 2u) {
                    Prefix qv;
                    qv.kept = sl.pre_kept; qv.nals = sl.pre_nals; qv.inside = sl.pre_inside;
                    excl = fold(qv, a2);
                    done = true;
                }
            }
            const bool aborted = l.lb[par][0].abort != 0u;
            par ^= 1u;
            if (done) break;
            if (stall) {
                if (++spins > (1u << 20) || aborted) { ok = false; break; }
                __builtin_amdgcn_s_sleep(1);
                continue;
            }
            acc = a2;
            win_hi -= 256;
        }
    }
    if (tid == 0) {
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

/* Row r of the register set, r wave-uniform.  The rows stay in their own registers (a
 * rotating register file would have every v_mov wait for the prefetch that is in flight into
 * the row it moves); a scalar switch picks the one to read or to reload. */
__device__ __forceinline__ u32x4 get_row(const Rows& R, int r)
{
    switch (r) {
    case 0: return R.q0; case 1: return R.q1; case 2: return R.q2; case 3: return R.q3;
    case 4: return R.q4; case 5: return R.q5; case 6: return R.q6; default: return R.q7;
    }
}
__device__ __forceinline__ uint32_t get_row_x(const Rows& R, int r)
{
    switch (r) {
    case 0: return R.q0.x; case 1: return R.q1.x; case 2: return R.q2.x; case 3: return R.q3.x;
    case 4: return R.q4.x; case 5: return R.q5.x; case 6: return R.q6.x; default: return R.q7.x;
    }
}
__device__ __forceinline__ void load_row(Rows& R, int r, const u32x4* p)
{
    switch (r) {
    case 0: R.q0 = *p; break; case 1: R.q1 = *p; break; case 2: R.q2 = *p; break; case 3: R.q3 = *p; break;
    case 4: R.q4 = *p; break; case 5: R.q5 = *p; break; case 6: R.q6 = *p; break; default: R.q7 = *p; break;
    }
}
__device__ __forceinline__ void set_row(Rows& R, int r, const u32x4& v)
{
    switch (r) {
    case 0: R.q0 = v; break; case 1: R.q1 = v; break; case 2: R.q2 = v; break; case 3: R.q3 = v; break;
    case 4: R.q4 = v; break; case 5: R.q5 = v; break; case 6: R.q6 = v; break; default: R.q7 = v; break;
    }
}

__device__ __forceinline__ Prefix prefix_uniform(const Prefix& p)
{
    Prefix r;
    r.kept = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(p.kept >> 32)) << 32) |
             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)p.kept);
    r.nals = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(p.nals >> 32)) << 32) |
             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)p.nals);
    r.inside = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.inside);
    return r;
}

__device__ __forceinline__ TileAgg agg_uniform(const TileAgg& a)
{
    TileAgg r;
    r.cnt = __builtin_amdgcn_readfirstlane(a.cnt);
    r.known = __builtin_amdgcn_readfirstlane(a.known);
    r.sig = __builtin_amdgcn_readfirstlane(a.sig);
    r.last = __builtin_amdgcn_readfirstlane(a.last);
    return r;
}

__global__ __launch_bounds__(k3Threads, 4)
void k_scan_extract3(const uint8_t* __restrict__ stream, uint64_t n, uint64_t num_tiles,
                     hbs_nal_entry* __restrict__ index, uint64_t index_cap,
                     uint8_t* __restrict__ rbsp, uint64_t rbsp_cap,
                     unsigned long long* __restrict__ desc, RunHeader* __restrict__ hdr)
{
    __shared__ Lds3 l;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    EmitTarget tgt;
    tgt.index = index; tgt.index_cap = index_cap; tgt.hdr = hdr;

    Rows cur;
    uint64_t tile = blockIdx.x;
    if (tile < num_tiles) fetch_rows(cur, stream, tile * (uint64_t)k3TileBytes + (uint64_t)(wv * k3WaveBytes), n, lane);
    HBS3_T_DECL

    for (; tile < num_tiles; tile += gridDim.x) {
        const uint64_t seg = tile * (uint64_t)k3TileBytes + (uint64_t)(wv * k3WaveBytes);

        /* ---- pass A: row aggregates -> wave aggregate (scalars) ---------------------------- */
        /* One loop body serves all rows (a scalar switch reads row r): eight times less code
         * and fewer live registers than unrolling over the named rows. */
        uint32_t slow = 0;                            /* bit r: row r needs the per-chunk path */
        TileAgg wagg = {0u, 0u, 0u, kKindNone};
        {
            uint32_t e_prev = cur.before;
#pragma unroll 1
            for (int r = 0; r < k3Rows; ++r) {
                const u32x4 q = get_row(cur, r);
                const uint32_t e_next = (r == k3Rows - 1) ? cur.after : (uint32_t)__builtin_amdgcn_readlane((int)get_row_x(cur, r + 1), 0);
                const uint64_t g0 = seg + (uint64_t)(r * k3RowBytes);
                if (g0 < n) {
                    RegView v; BlockMarks m; LanePos pos; TileAgg ra;
                    if (row_classify(q, e_prev, e_next, stream, g0, n, lane, v, m, ra, pos)) { slow |= 1u << r; wagg = combine(wagg, agg_uniform(ra)); }
                    else { const TileAgg pl = {0u, 0u, (uint32_t)k3RowBytes, kKindNone}; wagg = combine(wagg, pl); }
                }
                e_prev = (uint32_t)__builtin_amdgcn_readlane((int)q.w, 63);
            }
        }
        HBS3_T_MARK(0)
        if (lane == 0) l.wave_agg[wv] = wagg;
        __syncthreads();
        TileAgg before = {0u, 0u, 0u, kKindNone}, tagg = {0u, 0u, 0u, kKindNone};
#pragma unroll
        for (int w = 0; w < k3Waves; ++w) {
            const TileAgg a = l.wave_agg[w];
            if (w < wv) before = combine(before, a);
            tagg = combine(tagg, a);
        }
        HBS3_T_MARK(1)

        Prefix excl;
        if (!look_back3(l, desc, tile, tagg, hdr, tid, excl)) return;
        HBS3_T_MARK(2)
        if (tid == 0 && tile == num_tiles - 1) {
            const Prefix incl = fold(excl, tagg);
            hdr->final_kept = incl.kept; hdr->final_nals = incl.nals; hdr->final_inside = incl.inside;
        }
        /* The next tile's rows are loaded INTO the registers of the rows this pass has just
         * written out (one load per loop iteration, see below): a prefetch that overlaps the
         * emit pass without a second register set. */
        const bool more = tile + gridDim.x < num_tiles;
        const uint64_t next_seg = (tile + gridDim.x) * (uint64_t)k3TileBytes + (uint64_t)(wv * k3WaveBytes);

        /* ---- pass B: emit + copy, rows in order, running prefix in scalars ------------------- */
        const uint32_t tile_kept = tagg.known + (excl.inside ? tagg.sig : 0u);
        const bool can_store = rbsp != nullptr && excl.kept + tile_kept <= rbsp_cap;
        if (rbsp != nullptr && !can_store && tid == 0) atomicMax(&hdr->error, (uint32_t)(-HBS_E_CAPACITY));
        Prefix run = prefix_uniform(fold(excl, before));  /* state at this wavefront's first byte */
        {
            uint32_t e_prev = cur.before;
#pragma unroll 1
            for (int r = 0; r < k3Rows; ++r) {
                const uint64_t g0 = seg + (uint64_t)(r * k3RowBytes);
                const u32x4 q = get_row(cur, r);
                const uint32_t e_next = (r == k3Rows - 1) ? cur.after : (uint32_t)__builtin_amdgcn_readlane((int)get_row_x(cur, r + 1), 0);
                if (g0 < n) {
                    if (!((slow >> r) & 1u)) {
                        if (run.inside) {
                            if (can_store) reinterpret_cast<Unaligned16_3*>(rbsp + run.kept + 16ull * (uint32_t)lane)->v = q;
                            run.kept += k3RowBytes;
                        }
                    } else {
                        RegView v; BlockMarks m; LanePos pos; TileAgg ra;
                        (void)row_classify(q, e_prev, e_next, stream, g0, n, lane, v, m, ra, pos);
                        const bool inside = (pos.in_state == 1u) || (pos.in_state == 2u && run.inside);
                        const uint64_t dst = run.kept + pos.known + (run.inside ? pos.sig : 0u);
                        const uint32_t keep = (uint32_t)emit_block_t<kChunk, RegView>(v, 0, v.g0, m, inside, run.nals + pos.cnt, dst, tgt);
                        if (can_store && keep != 0u) {
                            if (keep == 0xFFFFu) {
                                reinterpret_cast<Unaligned16_3*>(rbsp + dst)->v = q;
                            } else {
                                uint64_t lo, hi;
                                const uint32_t c = compact_chunk_regs(q.x, q.y, q.z, q.w, keep, lo, hi);
                                store_pieces(rbsp + dst, lo, hi, c);
                            }
                        }
                        run = prefix_uniform(fold(run, agg_uniform(ra)));
                    }
                }
                e_prev = (uint32_t)__builtin_amdgcn_readlane((int)q.w, 63);
            }
        }
        /* next tile: all rows at once, with static register names (inside the rolled loop the
         * compiler cannot tell which row a pending load targets and waits for all of them) */
        if (more) fetch_rows(cur, stream, next_seg, n, lane);
        HBS3_T_MARK(3)
        __syncthreads();                                  /* wave_agg / lb slots are reused by the next tile */
        HBS3_T_MARK(4)
    }
    HBS3_T_FLUSH
}

#ifdef HBS_PHASE_TIMING
extern "C" int hbs_debug_phase_cycles3(unsigned long long* host_out /* [1024][8] */)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_phase_cycles3), sizeof(unsigned long long) * 1024 * 8);
}
#endif

int scan3_grid_blocks(int device, int* blocks_per_cu_out)
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return -1;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_scan_extract3, k3Threads, 0) != hipSuccess) return -1;
    /* all workgroups must be co-resident (tile i waits on tiles < i): stay within what the
     * occupancy query grants, and at 2 per CU (16 wavefronts) which is what 128 VGPRs allow */
    if (per_cu > 2) per_cu = 2;
    if (per_cu < 1) per_cu = 1;
    if (blocks_per_cu_out) *blocks_per_cu_out = per_cu;
    return prop.multiProcessorCount * per_cu;
}

void launch_scan_extract3_kernel(const ScanArgs& a, uint64_t num_tiles, hipStream_t st)
{
    uint64_t grid = (uint64_t)a.grid_blocks3;
    if (grid > num_tiles) grid = num_tiles;
    k_scan_extract3<<<dim3((unsigned)grid), dim3(k3Threads), 0, st>>>(
        a.stream, a.n, num_tiles, a.index, a.index_cap, a.rbsp, a.rbsp_cap, a.desc, a.hdr);
}

} // namespace hbs
