/*
 * hbs_scan.hip -- K12: fused start-code scan + NAL index + RBSP extraction for
 * gfx950 (MI355X), one pass over the stream.
 *
 * Replaces the reference's sequential loop
 *     while (find_nal_unit(p, sz, &s, &e) > 0) { nal_to_rbsp(p+s, ...); p += e; }
 * (hevc_analyze.c:135-177 driving h264_nal.c:38-76 and :147-200) with a
 * chained scan:
 *
 *   - persistent workgroups (256 threads = 4 wave64), tile i -> workgroup
 *     i mod grid; a tile is 16 KiB staged in LDS with coalesced 16-byte loads;
 *   - each thread classifies 64 contiguous bytes (hbs_tile.h), a workgroup
 *     scan turns that into a tile aggregate {NAL starts, kept bytes as a
 *     function of the carried inside/outside state, state after the tile};
 *   - decoupled look-back over per-tile descriptors (two self-validating
 *     8-byte words, agent-scope relaxed atomics: the data is the flag, so no
 *     fences) yields the exclusive prefix {arena offset, NAL ordinal, state};
 *   - the tile's kept bytes are gathered from LDS into 16-byte words aligned in
 *     the arena and stored coalesced; index entries are scattered by the
 *     threads that own the start / end events.
 *
 * HBM traffic: stream read once (1 B/B), RBSP written once (~1 B/B), index
 * 32 B/NAL, descriptors 16 B per 16 KiB tile.  No MFMA: byte scan, HBM-bound.
 */
#include <hip/hip_runtime.h>
#include "hbs_tile.h"
#include "hbs_scan.h"

namespace hbs {

struct TileLds {
    alignas(16) uint8_t raw[kHalo + kTileBytes + kHalo];  /* raw[kHalo+i] = S[tile_base+i] */
    uint64_t keep[kThreads];
    uint32_t rank[kThreads + 1];
    uint32_t wave_last[4];
    uint32_t wave_cnt[4];
    uint32_t wave_known[4];
    uint32_t wave_sig[4];
    /* exclusive prefix of this tile, broadcast by wave 0 */
    uint64_t ex_kept;
    uint64_t ex_nals;
    uint32_t ex_inside;
    uint32_t abort;
};

/* 16 stream bytes at offset g (may straddle or exceed n): 0xFF outside [0,n) */
__device__ __forceinline__ uint4 load16_guarded(const uint8_t* __restrict__ s, int64_t g, uint64_t n)
{
    if (g >= 0 && (uint64_t)g + 16 <= n) return *reinterpret_cast<const uint4*>(s + g);
    uint32_t w[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    if (g + 16 > 0 && g < (int64_t)n) {
        for (int b = 0; b < 16; ++b) {
            const int64_t q = g + b;
            if (q >= 0 && (uint64_t)q < n) {
                w[b >> 2] &= ~(0xFFu << (8 * (b & 3)));
                w[b >> 2] |= (uint32_t)s[q] << (8 * (b & 3));
            }
        }
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ void stage_tile(TileLds& l, const uint8_t* __restrict__ s, uint64_t tile_base, uint64_t n, int tid)
{
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        v[u] = load16_guarded(s, (int64_t)(tile_base + 16ull * (uint32_t)(u * kThreads + tid)), n);
#pragma unroll
    for (int u = 0; u < 4; ++u)
        *reinterpret_cast<uint4*>(&l.raw[kHalo + 16 * (u * kThreads + tid)]) = v[u];
    if (tid == 0)
        *reinterpret_cast<uint4*>(&l.raw[0]) = load16_guarded(s, (int64_t)tile_base - 16, n);
    if (tid == 64)
        *reinterpret_cast<uint4*>(&l.raw[kHalo + kTileBytes]) = load16_guarded(s, (int64_t)(tile_base + kTileBytes), n);
}

/* inclusive wave scan by DPP-free shuffles (6 steps) */
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

enum : uint32_t { kInOutside = 0, kInInside = 1, kInCarry = 2 };

struct ThreadPrefix {
    uint32_t in_state;   /* kIn*: state at the first byte of the block     */
    uint32_t cnt;        /* NAL starts before the block (tile-relative)    */
    uint32_t known;      /* kept bytes before the block, state-independent */
    uint32_t sig;        /* kept bytes before the block if tile carry-in is inside */
};

/* workgroup scan of the block summaries -> per-thread exclusive prefix + tile aggregate */
__device__ __forceinline__ ThreadPrefix block_scan(TileLds& l, const BlockSum& s, int tid, TileAgg& agg)
{
    const int lane = tid & 63, wv = tid >> 6;
    const uint64_t m_ev = __ballot(s.last != kKindNone);
    const uint64_t m_st = __ballot(s.last == kKindStart);
    const uint64_t lower = m_ev & ((1ull << lane) - 1ull);

    uint32_t in_state = kInCarry;
    if (lower != 0) in_state = (uint32_t)((m_st >> (63 - __builtin_clzll(lower))) & 1ull);
    if (lane == 0) {
        uint32_t wl = kKindNone;
        if (m_ev != 0) wl = ((m_st >> (63 - __builtin_clzll(m_ev))) & 1ull) ? kKindStart : kKindStop;
        l.wave_last[wv] = wl;
    }
    __syncthreads();

    uint32_t carry_kind = kKindNone;
    for (int w = wv - 1; w >= 0; --w) {
        const uint32_t k = l.wave_last[w];
        if (k != kKindNone) { carry_kind = k; break; }
    }
    if (in_state == kInCarry && carry_kind != kKindNone)
        in_state = (carry_kind == kKindStart) ? kInInside : kInOutside;

    const uint32_t known = s.known + (in_state == kInInside ? s.carry : 0u);
    const uint32_t sig = (in_state == kInCarry) ? s.carry : 0u;

    /* per-wave sums fit 13 bits each: scan known|sig packed, cnt separately */
    const uint32_t packed = known | (sig << 16);
    const uint32_t ip = wave_incl_scan(packed, lane);
    const uint32_t ic = wave_incl_scan(s.cnt, lane);
    if (lane == 63) {
        l.wave_known[wv] = ip & 0xFFFFu;
        l.wave_sig[wv] = ip >> 16;
        l.wave_cnt[wv] = ic;
    }
    __syncthreads();

    ThreadPrefix p;
    p.in_state = in_state;
    p.known = (ip & 0xFFFFu) - known;
    p.sig = (ip >> 16) - sig;
    p.cnt = ic - s.cnt;
    uint32_t tk = 0, ts = 0, tc = 0, tl = kKindNone;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w < wv) { p.known += l.wave_known[w]; p.sig += l.wave_sig[w]; p.cnt += l.wave_cnt[w]; }
        tk += l.wave_known[w]; ts += l.wave_sig[w]; tc += l.wave_cnt[w];
        if (l.wave_last[w] != kKindNone) tl = l.wave_last[w];
    }
    agg.known = tk; agg.sig = ts; agg.cnt = tc; agg.last = tl;
    return p;
}

__device__ __forceinline__ uint64_t ld_desc(const unsigned long long* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_desc(unsigned long long* p, uint64_t v)
{
    __hip_atomic_store(p, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ TileAgg readlane_agg(const TileAgg& a, int l)
{
    TileAgg r;
    r.cnt = __builtin_amdgcn_readlane(a.cnt, l);
    r.known = __builtin_amdgcn_readlane(a.known, l);
    r.sig = __builtin_amdgcn_readlane(a.sig, l);
    r.last = __builtin_amdgcn_readlane(a.last, l);
    return r;
}

/*
 * Decoupled look-back, executed by wave 0.  desc[2*t], desc[2*t+1] are the two
 * words of tile t.  Returns the exclusive prefix of `tile`; false on timeout.
 */
__device__ __forceinline__ bool look_back(unsigned long long* desc, uint64_t tile, const TileAgg& mine,
                                          RunHeader* hdr, int lane, Prefix& excl)
{
    if (tile == 0) {
        excl.kept = 0; excl.nals = 0; excl.inside = 0;
    } else {
        if (lane == 0) {
            st_desc(&desc[2 * tile], pack_agg0(mine));
            st_desc(&desc[2 * tile + 1], pack_agg1(mine));
        }
        TileAgg acc = {0u, 0u, 0u, kKindNone};       /* tiles between the window and `tile` */
        int64_t win_hi = (int64_t)tile - 1;           /* nearest tile of the current window  */
        uint32_t spins = 0;
        for (;;) {
            const int64_t t = win_hi - lane;           /* lane l looks at tile win_hi - l     */
            uint64_t w0 = kDescPrefix, w1 = kDescPrefix;   /* virtual tile -1: empty prefix   */
            if (t >= 0) {
                w0 = ld_desc(&desc[2 * t]);
                w1 = ld_desc(&desc[2 * t + 1]);
            }
            const uint32_t s0 = (uint32_t)(w0 & 3u), s1 = (uint32_t)(w1 & 3u);
            const bool ready = (s0 == s1) && (s0 != kDescEmpty);
            const bool is_pre = ready && (s0 == kDescPrefix);
            const uint64_t m_pre = __ballot(is_pre);
            const uint64_t m_ready = __ballot(ready);
            const int lstar = m_pre ? (int)__builtin_ctzll(m_pre) : 64;      /* nearest prefix */
            const uint64_t need = (lstar >= 64) ? ~0ull : ((1ull << lstar) - 1ull);
            if ((m_ready & need) != need) {                                    /* a nearer tile is not ready */
                if (++spins > (1u << 22) || __hip_atomic_load(&hdr->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    if (lane == 0) {
                        __hip_atomic_store(&hdr->abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        atomicMax(&hdr->error, (uint32_t)(-HBS_E_TIMEOUT));
                    }
                    return false;
                }
                __builtin_amdgcn_s_sleep(2);
                continue;
            }
            const TileAgg a = unpack_agg(w0, w1);
            TileAgg win = {0u, 0u, 0u, kKindNone};
            const int top = (lstar >= 64) ? 63 : lstar - 1;
            for (int l = top; l >= 0; --l) win = combine(win, readlane_agg(a, l));   /* earliest first */
            acc = combine(win, acc);
            if (lstar < 64) {
                Prefix p = unpack_pre(w0, w1);
                p.kept = ((uint64_t)__builtin_amdgcn_readlane((uint32_t)(p.kept >> 32), lstar) << 32) |
                         __builtin_amdgcn_readlane((uint32_t)p.kept, lstar);
                p.nals = ((uint64_t)__builtin_amdgcn_readlane((uint32_t)(p.nals >> 32), lstar) << 32) |
                         __builtin_amdgcn_readlane((uint32_t)p.nals, lstar);
                p.inside = __builtin_amdgcn_readlane(p.inside, lstar);
                excl = fold(p, acc);
                break;
            }
            win_hi -= 64;
        }
    }
    if (lane == 0) {
        const Prefix incl = fold(excl, mine);
        st_desc(&desc[2 * tile], pack_pre0(incl));
        st_desc(&desc[2 * tile + 1], pack_pre1(incl));
    }
    return true;
}

__device__ __forceinline__ void store_word(uint8_t* dst, const GatherOut& g)
{
    if (g.lo == 0 && g.hi == 16) {
        *reinterpret_cast<uint4*>(dst) = make_uint4(g.w[0], g.w[1], g.w[2], g.w[3]);
    } else {
        for (uint32_t o = g.lo; o < g.hi; ++o) dst[o] = (uint8_t)(g.w[o >> 2] >> (8u * (o & 3u)));
    }
}

__global__ __launch_bounds__(kThreads)
void k_scan_extract(const uint8_t* __restrict__ stream, uint64_t n, uint64_t num_tiles,
                    hbs_nal_entry* __restrict__ index, uint64_t index_cap,
                    uint8_t* __restrict__ rbsp, uint64_t rbsp_cap,
                    unsigned long long* __restrict__ desc, RunHeader* __restrict__ hdr)
{
    __shared__ TileLds l;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    EmitTarget tgt;
    tgt.index = index; tgt.index_cap = index_cap; tgt.hdr = hdr;

    for (uint64_t tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
        const uint64_t tile_base = tile * (uint64_t)kTileBytes;
        stage_tile(l, stream, tile_base, n, tid);
        __syncthreads();

        const uint8_t* blk = &l.raw[kHalo + kBlockBytes * tid];
        const uint64_t g0 = tile_base + (uint64_t)(kBlockBytes * tid);
        BlockMarks marks;
        BlockSum sum;
        classify_block(blk, g0, n, marks, sum);

        TileAgg agg;
        const ThreadPrefix tp = block_scan(l, sum, tid, agg);

        if (wv == 0) {
            Prefix excl;
            const bool ok = look_back(desc, tile, agg, hdr, lane, excl);
            if (lane == 0) {
                l.ex_kept = excl.kept; l.ex_nals = excl.nals; l.ex_inside = excl.inside;
                l.abort = ok ? 0u : 1u;
                if (ok && tile == num_tiles - 1) {
                    const Prefix incl = fold(excl, agg);
                    hdr->final_kept = incl.kept; hdr->final_nals = incl.nals; hdr->final_inside = incl.inside;
                }
            }
        }
        __syncthreads();
        if (l.abort) return;

        const uint64_t ex_kept = l.ex_kept, ex_nals = l.ex_nals;
        const bool ex_inside = l.ex_inside != 0;
        const bool inside = (tp.in_state == kInInside) || (tp.in_state == kInCarry && ex_inside);
        const uint32_t rank0 = tp.known + (ex_inside ? tp.sig : 0u);
        const uint32_t tile_kept = agg.known + (ex_inside ? agg.sig : 0u);

        const uint64_t keep = emit_block(blk, g0, marks, inside, ex_nals + tp.cnt, ex_kept + rank0, tgt);
        l.keep[tid] = keep;
        l.rank[tid] = rank0;
        if (tid == 0) l.rank[kThreads] = tile_kept;
        __syncthreads();

        if (rbsp != nullptr && tile_kept != 0) {
            if (ex_kept + tile_kept <= rbsp_cap) {
                const uint32_t ob = (uint32_t)(ex_kept & 15ull);
                const uint32_t nwords = (ob + tile_kept + 15u) >> 4;
                uint8_t* out = rbsp + (ex_kept - ob);
                for (uint32_t wi = tid; wi < nwords; wi += kThreads) {
                    const GatherOut g = gather_word(&l.raw[kHalo], l.rank, l.keep, wi, ob, tile_kept);
                    store_word(out + 16ull * wi, g);
                }
            } else if (tid == 0) {
                atomicMax(&hdr->error, (uint32_t)(-HBS_E_CAPACITY));
            }
        }
        __syncthreads();
    }
}

__global__ void k_init_header(RunHeader* hdr)
{
    hdr->final_kept = 0; hdr->final_nals = 0; hdr->final_inside = 0;
    hdr->error = 0; hdr->first_empty = ~0ull; hdr->abort_flag = 0; hdr->pad = 0;
}

__global__ void k_tail_fixup(const uint8_t* __restrict__ stream, uint64_t n,
                             hbs_nal_entry* index, uint64_t index_cap,
                             uint8_t* rbsp, uint64_t rbsp_cap, RunHeader* hdr, hbs_summary* sum)
{
    uint8_t tail[8];
    for (int i = 0; i < 8; ++i) {
        const int64_t q = (int64_t)n - 8 + i;
        tail[i] = (q >= 0) ? stream[q] : (uint8_t)0xFF;
    }
    tail_fixup(hdr, index, index_cap, rbsp, rbsp_cap, tail, n, sum);
}

__global__ void k_fill_rbsp_len(const RunHeader* hdr, hbs_nal_entry* index, uint64_t index_cap)
{
    const uint64_t found = hdr->final_nals;
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < found; k += (uint64_t)gridDim.x * blockDim.x)
        fill_rbsp_len(hdr, index, index_cap, k);
}

/* ---- host side ------------------------------------------------------------ */

int scan_grid_blocks(int device, int* blocks_per_cu_out)
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return -1;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_scan_extract, kThreads, 0) != hipSuccess) return -1;
    /* All workgroups must be co-resident (tile i waits on tiles < i).  The
     * occupancy query can over-report by one block per CU on ROCm 7.2
     * (MI355X_MICROARCH.md, Residency), so stay one below it and at most 8. */
    if (per_cu > 1) per_cu -= 1;
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    if (blocks_per_cu_out) *blocks_per_cu_out = per_cu;
    return prop.multiProcessorCount * per_cu;
}

hipError_t launch_scan_extract(const ScanArgs& a, hipStream_t st)
{
    hipError_t e;
    k_init_header<<<1, 1, 0, st>>>(a.hdr);
    if (a.index_cap) {
        e = hipMemsetAsync(a.index, 0, a.index_cap * sizeof(hbs_nal_entry), st);
        if (e != hipSuccess) return e;
    }
    const uint64_t num_tiles = (a.n + kTileBytes - 1) / kTileBytes;
    if (num_tiles) {
        e = hipMemsetAsync(a.desc, 0, num_tiles * 16, st);
        if (e != hipSuccess) return e;
        uint64_t grid = (uint64_t)a.grid_blocks;
        if (grid > num_tiles) grid = num_tiles;
        k_scan_extract<<<dim3((unsigned)grid), dim3(kThreads), 0, st>>>(
            a.stream, a.n, num_tiles, a.index, a.index_cap, a.rbsp, a.rbsp_cap, a.desc, a.hdr);
    }
    k_tail_fixup<<<1, 1, 0, st>>>(a.stream, a.n, a.index, a.index_cap, a.rbsp, a.rbsp_cap, a.hdr, a.summary);
    if (a.index_cap)
        k_fill_rbsp_len<<<256, 256, 0, st>>>(a.hdr, a.index, a.index_cap);
    return hipGetLastError();
}

} // namespace hbs
