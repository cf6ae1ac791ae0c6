/*
 * hbs_emit.hip -- K3: RBSP arena + NAL index -> Annex-B stream (emulation
 * prevention insertion and start codes), and the device generator of the
 * synthetic stream S(seed, n_nals, mode).
 *
 * Replaces rbsp_to_nal() per NAL (reference h264_nal.c:92-132, called from
 * write_hevc_nal_unit, hevc_stream.c:1324-1327) plus the start-code bytes the
 * reference's callers put in front of each NAL.
 *
 * A 03 is only ever inserted behind two zero bytes, so chunk_flag() (hbs_sparse.h:
 * no two adjacent zeros start in bytes [-2, 16) of the chunk) clears nearly every
 * 16-byte chunk for a plain copy; flagged chunks run the byte-exact rbsp_to_nal
 * rules of hbs_emit.h.  Single pass (RBSP read once, stream written once), two ways:
 * k3_tiles cuts the ARENA into 192 KiB tiles as K12 cuts a stream (when the index's
 * NALs lie back to back; see "arena tiles" below), k3_fused cuts the work by NAL
 * (any index; see the comment in front of it).  k3_count / scan / k3_emit is the
 * older three-step version (RBSP read twice): zero-heavy arenas, picked by a probe.
 */
#include <hip/hip_runtime.h>
#include "hbs_wave.h"
#include <utility>
#include "hbs_sparse.h"
#include "hbs_elems.h"
#include "hbs_emit.h"
#include "hbs_emit_launch.h"

#ifndef HBS_K3F_COPY_DEPTH
#define HBS_K3F_COPY_DEPTH -1     /* k3_fused: stores / loads of a wavefront in flight (-1: no limit).  Round 3 tried 3 / 8 and 3 / 4,
                                     the recipe that took k3_tiles from 6.9 to 6.1 ms: nothing here (8.7-8.9 ms either way, scripts/experiments/k3f_ab.sh) */
#endif
#ifndef HBS_K3F_LOAD_DEPTH
#define HBS_K3F_LOAD_DEPTH -1
#endif

namespace hbs {

/* Which of the two ways?  The single-pass kernel (k3_fused) sends every row that holds a flagged chunk through
 * a rolled, byte-exact loop -- twice; past one flagged chunk in kEmitDenseOneIn (zero-heavy data) the older
 * count / scan / emit kernels, whose cost does not depend on the data that much, are faster (measured,
 * scripts/emit_density.py: 807 against 684 GB/s at 1 % zeros, 323 against 163 at 10 %).  A probe samples the
 * arena and both ways are enqueued; the one it rules out returns at once. */
constexpr uint32_t kEmitDenseOneIn = 800;
__device__ __forceinline__ bool emit_probe_dense(const uint32_t* probe) { return (uint64_t)probe[1] * kEmitDenseOneIn > (uint64_t)probe[0]; }
/* The arena-tile kernel walks a tile's elements 64 at a time without looking at the rows between them, so it keeps its pace up
 * to the density at which tiles pass its element limit (kTDenseLimit of 12288 chunks): round 3, with the probe counting exact
 * patterns, a zero-heavy arena (2-3 % of the chunks) is its case, not the three steps'. */
constexpr uint32_t kEmitTilesDenseOneIn = 20;
__device__ __forceinline__ bool emit_probe_dense_tiles(const uint32_t* probe) { return (uint64_t)probe[1] * kEmitTilesDenseOneIn > (uint64_t)probe[0]; }

/* the arena-tile kernel (k3_tiles, below) takes the sparse case when k3t_check found the index eligible: tflag[0] = a
 * violation was seen, tflag[1] = the global conditions hold */
__device__ __forceinline__ bool tile_path_on(const uint32_t* tflag) { return tflag && tflag[1] == 1u && tflag[0] == 0u && tflag[3] == 0u; }
/* (tflag[2] was "the tile kernel gave up on a tile dense in elements": since round 4 it walks such a tile by rows, nothing sets it) */
__device__ __forceinline__ bool tile_path_done(const uint32_t* tflag) { return tile_path_on(tflag); }

/* the three steps are for what the probe calls dense -- unless the tile kernel, in front of them, has done the call */
__device__ __forceinline__ bool three_steps_run(const uint32_t* probe, const uint32_t* tflag)
{
    return !probe || (emit_probe_dense(probe) && !(!emit_probe_dense_tiles(probe) && tile_path_done(tflag)));
}

/* tflag[3] (k3t_check, in front of everything that follows the index into the arena): an entry of the index lies outside the
 * caller's RBSP buffer.  The call then ends with HBS_E_ARG and nothing is read through the index. */
__device__ __forceinline__ bool index_bad(const uint32_t* vflag) { return vflag[3] != 0u; }

/* tflag[4] (k3t_check): the index is NOT one stretch of the arena with gaps below 16 bytes.  While it is -- and the arena tiles
 * have not done the call -- arenas of tiny NALs go through the group kernel (hbs_emit_groups.h) instead of a lane per NAL. */
__device__ __forceinline__ bool group_path_on(const uint32_t* tflag);
enum : int { kWhenAlways = 0, kWhenSparse = 1, kWhenDense = 2, kWhenEither = 3, kWhenNoTiles = 4, kWhenGroups = 5 };
/* the helper kernels of the two fall-back chains: those of the kernel by NALs (its item list) run when the data is sparse and
 * the tile kernel, in front of them since round 3, has not done the call; those of the three steps when these run */
__device__ __forceinline__ bool emit_skip(const uint32_t* probe, int when, const uint32_t* tflag)
{
    if (when == kWhenAlways) return false;
    if (when == kWhenSparse) return (probe && emit_probe_dense(probe)) || tile_path_done(tflag);
    if (when == kWhenEither) return emit_skip(probe, kWhenSparse, tflag) && !three_steps_run(probe, tflag);
    if (when == kWhenGroups) return !group_path_on(tflag);        /* the scan between the group kernel's two passes */
    if (when == kWhenNoTiles) return tile_path_on(tflag) || group_path_on(tflag);   /* small NALs: a lane per NAL unless the arena tiles or the group kernel do the call */
    return !three_steps_run(probe, tflag);
}
/* kWhenEither (the automatic mode, probe != nullptr: one scan serves whichever chain runs): where the scan's total goes */
__device__ __forceinline__ unsigned long long* scan_total_of(unsigned long long* total, unsigned long long* total_dense,
                                                             const uint32_t* probe, int when, const uint32_t* tflag)
{
    return (when == kWhenEither && three_steps_run(probe, tflag)) ? total_dense : total;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__device__ __forceinline__ uint32_t wave_excl_scan_u32(uint32_t v, int lane, uint32_t& total)
{
    uint32_t x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(x, d, 64);
        if (lane >= d) x += t;
    }
    total = __shfl(x, 63, 64);
    return x - v;
}

/* the density probe: kProbeBlocks workgroups, each a window of the arena (role of the workgroups behind the first kCheckBlocks
 * of k3t_check: one launch for the two things every call needs before anything else) */
constexpr unsigned kCheckBlocks = 1024, kProbeBlocks = 64;
__device__ __forceinline__ void probe_window(const uint8_t* __restrict__ rbsp, uint64_t bytes, uint32_t* __restrict__ probe, unsigned block)
{
    struct __attribute__((packed, aligned(1))) U16 { u32x4 v; };
    const uint64_t stride = (bytes / kProbeBlocks) & ~15ull;
    const uint64_t base = (uint64_t)block * stride;
    uint32_t chunks = 0, flagged = 0;
    for (int k = 0; k < 4; ++k) {
        const uint64_t off = base + (uint64_t)(k * 256 + (int)threadIdx.x) * 16u;
        const bool in = off + 16 <= bytes;
        bool f = false;
        if (in) {
            const u32x4 q = reinterpret_cast<const U16*>(rbsp + off)->v;
            f = chunk_flag(0xFFFFFFFFu, q.x, q.y, q.z, q.w, 0xFFFFFFFFu) && chunk_pattern_any_dev(0xFFFFFFFFu, q.x, q.y, q.z, q.w, 0xFFFFFFFFu);
        }
        chunks += (uint32_t)__builtin_popcountll(__ballot(in));
        flagged += (uint32_t)__builtin_popcountll(__ballot(f));
    }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&probe[0], chunks); atomicAdd(&probe[1], flagged); }
}

__device__ __forceinline__ uint64_t gap_of(const hbs_nal_entry* __restrict__ idx, uint64_t k, int gap_mode)
{
    if (gap_mode == 1) return synth_gap(k);
    const uint64_t prev_end = k ? idx[k - 1].end : 0ull;
    return idx[k].start - prev_end;
}

/* 16 bytes of a NAL's RBSP at offset off from its first byte; 0xFF behind its end */
__device__ __forceinline__ u32x4 load_nal_chunk(const uint8_t* __restrict__ rbsp, uint64_t begin, uint32_t len, uint32_t off)
{
    struct __attribute__((packed, aligned(1))) U16 { u32x4 v; };
    if (off + 16u <= len) return reinterpret_cast<const U16*>(rbsp + begin + off)->v;
    uint32_t w0 = 0xFFFFFFFFu, w1 = 0xFFFFFFFFu, w2 = 0xFFFFFFFFu, w3 = 0xFFFFFFFFu;
#pragma unroll 1
    for (uint32_t b = 0; b < 16; ++b) {
        if (off + b < len) {
            const uint32_t m = ~(0xFFu << (8u * (b & 3u)));
            const uint32_t x = (uint32_t)rbsp[begin + off + b] << (8u * (b & 3u));
            if ((b >> 2) == 0) w0 = (w0 & m) | x;
            else if ((b >> 2) == 1) w1 = (w1 & m) | x;
            else if ((b >> 2) == 2) w2 = (w2 & m) | x;
            else w3 = (w3 & m) | x;
        }
    }
    u32x4 v;
    v.x = w0; v.y = w1; v.z = w2; v.w = w3;
    return v;
}

/* A flagged chunk, exactly: its bytes (one 16-byte load), the count it is entered with (the dword in front; a run of four
 * or more zeros is followed back in memory), the 03s it gets.  off is a multiple of 16 below len. */
/* lead_count() (hbs_emit.h: the count a byte is entered with = the run of zeros in front of it inside its NAL), sixteen bytes a
 * step: byte by byte, every element inside a long run of zeros walked the run back through a chain of dependent loads -- a tile
 * with 16 KiB of zeros in it took milliseconds (round 4: scripts/emit_paths.py with HBS_EMIT_MIXED=2) */
__device__ __forceinline__ uint32_t lead_count_dev(const uint8_t* __restrict__ rbsp, uint64_t nal_begin, uint64_t p)
{
    struct __attribute__((packed, aligned(1))) U16 { u32x4 v; };
    uint64_t z = 0;
#pragma unroll 1
    while (p - z >= nal_begin + 16u) {
        const u32x4 q = reinterpret_cast<const U16*>(rbsp + (p - z - 16u))->v;
        if ((q.x | q.y | q.z | q.w) == 0u) { z += 16u; continue; }
        const uint32_t tz = q.w ? (uint32_t)__builtin_clz(q.w) >> 3 : q.z ? 4u + ((uint32_t)__builtin_clz(q.z) >> 3)
                          : q.y ? 8u + ((uint32_t)__builtin_clz(q.y) >> 3) : 12u + ((uint32_t)__builtin_clz(q.x) >> 3);
        z += tz;
        return z == 0 ? 0u : ((z & 1ull) ? 1u : 2u);
    }
    while (p - z > nal_begin && rbsp[p - 1 - z] == 0) ++z;
    return z == 0 ? 0u : ((z & 1ull) ? 1u : 2u);
}

struct ExactChunk { u32x4 q; uint32_t nb, mask; };
__device__ __forceinline__ ExactChunk exact_chunk(const uint8_t* __restrict__ rbsp, uint64_t begin, uint32_t len, uint32_t off)
{
    struct __attribute__((packed, aligned(1))) U4 { uint32_t v; };
    ExactChunk e;
    e.q = load_nal_chunk(rbsp, begin, len, off);
    e.nb = len - off < 16u ? len - off : 16u;
    uint32_t count = 0;                                   /* a NAL starts with count = 0 */
    if (off != 0) {
        count = lead_count4(reinterpret_cast<const U4*>(rbsp + begin + off - 4u)->v);
        if (count == kLeadUnknown) count = lead_count_dev(rbsp, begin, begin + off);
    }
    e.mask = insert_mask16(e.q.x, e.q.y, e.q.z, e.q.w, e.nb, count);
    return e;
}
__device__ __forceinline__ void store_exact(uint8_t* dst, const ExactChunk& e)
{
    struct __attribute__((packed, aligned(1))) U16 { u32x4 v; };
    if (e.mask == 0u && e.nb == 16u) reinterpret_cast<U16*>(dst)->v = e.q;
    else (void)emit_chunk16(dst, e.q.x, e.q.y, e.q.z, e.q.w, e.nb, e.mask);
}

/* the same from registers that already hold the chunk and the dword in front of it (the kernels that walk a NAL row by row) */
__device__ __forceinline__ ExactChunk exact_from_regs(const u32x4& q, uint32_t xp, const uint8_t* __restrict__ rbsp, uint64_t begin, uint32_t len, uint32_t off)
{
    ExactChunk e;
    e.q = q;
    e.nb = len - off < 16u ? len - off : 16u;
    uint32_t count = 0;
    if (off != 0) {
        count = lead_count4(xp);
        if (count == kLeadUnknown) count = lead_count_dev(rbsp, begin, begin + off);
    }
    e.mask = insert_mask16(q.x, q.y, q.z, q.w, e.nb, count);
    return e;
}

/* One row of a NAL: which of its chunks may need a 03 (conservative), from the row's registers. */
struct RowFlags {
    bool mine;             /* my chunk */
    uint64_t mask;         /* the row's  */
    uint32_t xp;           /* the four bytes in front of my chunk */
};
__device__ __forceinline__ RowFlags row_flags(const u32x4& q, uint32_t e_prev, uint32_t e_next, uint32_t off, uint32_t len)
{
    const uint32_t xp = from_prev_lane(q.w, e_prev);
    const uint32_t xn = from_next_lane(q.x, e_next);
    RowFlags r;
    r.xp = xp;
    r.mine = off < len && chunk_flag(xp, q.x, q.y, q.z, q.w, xn);
    r.mask = __ballot(r.mine);
    if (__builtin_popcountll(r.mask) > kExactFlagMin) {      /* many zero pairs in this KiB: which are followed by a byte <= 3? */
        r.mine = r.mine && chunk_pattern_any_dev(xp, q.x, q.y, q.z, q.w, xn);
        r.mask = __ballot(r.mine);
    }
    return r;
}

/* bytes rbsp_to_nal inserts into one NAL; a wavefront's work, the result wave-uniform */
__device__ __forceinline__ uint32_t count_nal(const uint8_t* __restrict__ rbsp, uint64_t begin, uint32_t len, int lane)
{
    const uint32_t nrows = (len + 1023u) / 1024u;
    uint32_t ins = 0;                                     /* wave-uniform */
    uint32_t e_prev = 0xFFFFFFFFu;                        /* a NAL starts with count = 0 */
    u32x4 qn = load_nal_chunk(rbsp, begin, len, 16u * (uint32_t)lane);
    for (uint32_t r = 0; r < nrows; ++r) {
        const u32x4 q = qn;
        const uint32_t off = 1024u * r + 16u * (uint32_t)lane;
        qn = load_nal_chunk(rbsp, begin, len, off + 1024u);
        const RowFlags f = row_flags(q, e_prev, (uint32_t)__builtin_amdgcn_readlane((int)qn.x, 0), off, len);
        if (f.mask != 0) {
            uint32_t c = 0;
            if (f.mine) c = (uint32_t)__builtin_popcount(exact_from_regs(q, f.xp, rbsp, begin, len, off).mask);
            ins += wave_sum_u32(c);
        }
        e_prev = (uint32_t)__builtin_amdgcn_readlane((int)q.w, 63);
    }
    return ins;
}

__global__ __launch_bounds__(256)
void k3_count(const uint8_t* __restrict__ rbsp, const hbs_nal_entry* __restrict__ idx, uint64_t n, int gap_mode,
              unsigned long long* __restrict__ nal_total, const uint32_t* __restrict__ probe, const uint32_t* __restrict__ vflag)
{
    if (!three_steps_run(probe, vflag) || index_bad(vflag)) return;
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t k = wave; k < n; k += nwaves) {
        const uint64_t begin = idx[k].rbsp_off;
        const uint32_t len = idx[k].rbsp_len;
        const uint32_t ins = count_nal(rbsp, begin, len, lane);
        if (lane == 0) nal_total[k] = gap_of(idx, k, gap_mode) + len + ins;
    }
}

/* ---- exclusive scan of v[0..n) into out[0..n), total to *total -----------------------------
 * three small launches: per-block sums of contiguous slices, a scan of the 1024 sums, and the
 * slices again with their offsets (coalesced 256-wide chunks, running carry). */
constexpr int kScanBlocks = 1024;

__device__ __forceinline__ unsigned long long wave_incl_scan_u64(unsigned long long x, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned long long t = __shfl_up(x, d, 64);
        if (lane >= d) x += t;
    }
    return x;
}

/* exclusive scan across the 256 threads of a workgroup; *total = sum (same in every thread) */
__device__ __forceinline__ unsigned long long block_excl_scan_u64(unsigned long long x, unsigned long long* wsum /* [4] LDS */,
                                                                  unsigned long long& total)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long inc = wave_incl_scan_u64(x, lane);
    __syncthreads();                       /* wsum may still be read from the previous round */
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    unsigned long long before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { if (w < wv) before += wsum[w]; tot += wsum[w]; }
    total = tot;
    return before + inc - x;
}

__global__ __launch_bounds__(256)
void k_scan_reduce(const unsigned long long* __restrict__ v, uint64_t n, unsigned long long* __restrict__ part,
                   const uint32_t* __restrict__ probe, int when, const uint32_t* __restrict__ tflag)
{
    if (emit_skip(probe, when, tflag)) return;
    __shared__ unsigned long long wsum[4];
    const uint64_t per = (n + kScanBlocks - 1) / kScanBlocks;
    const uint64_t lo = blockIdx.x * per, hi = (lo + per < n) ? lo + per : n;
    unsigned long long s = 0;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += 256) s += v[i];
    unsigned long long tot;
    (void)block_excl_scan_u64(s, wsum, tot);
    if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

__global__ __launch_bounds__(kScanBlocks)
void k_scan_parts(unsigned long long* __restrict__ part, unsigned long long* __restrict__ total_sparse, unsigned long long* __restrict__ total_dense,
                  const uint32_t* __restrict__ probe, int when, const uint32_t* __restrict__ tflag)
{
    if (emit_skip(probe, when, tflag)) return;
    unsigned long long* const total = scan_total_of(total_sparse, total_dense, probe, when, tflag);
    __shared__ unsigned long long sh[kScanBlocks];
    const int tid = threadIdx.x;
    const unsigned long long s = part[tid];
    sh[tid] = s;
    __syncthreads();
    for (int d = 1; d < kScanBlocks; d <<= 1) {
        const unsigned long long t = (tid >= d) ? sh[tid - d] : 0ull;
        __syncthreads();
        sh[tid] += t;
        __syncthreads();
    }
    part[tid] = sh[tid] - s;
    if (tid == kScanBlocks - 1) *total = sh[tid];
}

__global__ __launch_bounds__(256)
void k_scan_apply(const unsigned long long* __restrict__ v, unsigned long long* __restrict__ out, uint64_t n,
                  const unsigned long long* __restrict__ part, const uint32_t* __restrict__ probe, int when, const uint32_t* __restrict__ tflag)
{
    if (emit_skip(probe, when, tflag)) return;
    __shared__ unsigned long long wsum[4];
    const uint64_t per = (n + kScanBlocks - 1) / kScanBlocks;
    const uint64_t lo = blockIdx.x * per, hi = (lo + per < n) ? lo + per : n;
    unsigned long long carry = part[blockIdx.x];
    for (uint64_t base = lo; base < hi; base += 256) {
        const uint64_t i = base + threadIdx.x;
        const unsigned long long x = (i < hi) ? v[i] : 0ull;
        unsigned long long tot;
        const unsigned long long ex = block_excl_scan_u64(x, wsum, tot);
        if (i < hi) out[i] = carry + ex;
        carry += tot;
    }
}

/* the same scan by ONE workgroup, for up to kScanOneMax values: a thread sums its slice, the 1024 sums are scanned in LDS, the
 * thread writes its slice's prefixes.  One launch instead of three -- and an emit call enqueues two scans of which at most one
 * does anything (round 3: a 1 GiB call was 17 launches around one that matters). */
constexpr uint64_t kScanOneMax = 1ull << 18;      /* (2^21 was tried in round 4 to save two empty launches on the bench arena: one workgroup over 1.7 M values takes 4 ms when it runs) */
__global__ __launch_bounds__(1024)
void k_scan_one(const unsigned long long* __restrict__ v, unsigned long long* __restrict__ out, uint64_t n,
                unsigned long long* __restrict__ total_sparse, unsigned long long* __restrict__ total_dense,
                const uint32_t* __restrict__ probe, int when, const uint32_t* __restrict__ tflag)
{
    if (emit_skip(probe, when, tflag)) return;
    unsigned long long* const total = scan_total_of(total_sparse, total_dense, probe, when, tflag);
    __shared__ unsigned long long sh[1024];
    const int tid = threadIdx.x;
    const uint64_t per = (n + 1023u) / 1024u;
    const uint64_t lo = (uint64_t)tid * per < n ? (uint64_t)tid * per : n, hi = lo + per < n ? lo + per : n;
    unsigned long long s = 0;
    for (uint64_t i = lo; i < hi; ++i) s += v[i];
    sh[tid] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const unsigned long long t = (tid >= d) ? sh[tid - d] : 0ull;
        __syncthreads();
        sh[tid] += t;
        __syncthreads();
    }
    unsigned long long carry = sh[tid] - s;
    for (uint64_t i = lo; i < hi; ++i) { const unsigned long long x = v[i]; out[i] = carry; carry += x; }
    if (tid == 1023) *total = sh[1023];
}

static void launch_scan_u64(const unsigned long long* v, unsigned long long* out, uint64_t n, unsigned long long* total,
                            unsigned long long* part /* kScanBlocks entries */, hipStream_t st,
                            const uint32_t* probe = nullptr, int when = kWhenAlways, const uint32_t* tflag = nullptr,
                            unsigned long long* total_dense = nullptr /* kWhenEither: the total's place when the three steps run */)
{
    if (n <= kScanOneMax) {
        k_scan_one<<<1, 1024, 0, st>>>(v, out, n, total, total_dense, probe, when, tflag);
        return;
    }
    k_scan_reduce<<<kScanBlocks, 256, 0, st>>>(v, n, part, probe, when, tflag);
    k_scan_parts<<<1, kScanBlocks, 0, st>>>(part, total, total_dense, probe, when, tflag);
    k_scan_apply<<<kScanBlocks, 256, 0, st>>>(v, out, n, part, probe, when, tflag);
}

/* one NAL written by one wavefront: gap bytes at `base`, then the NAL with its 03s; `total` = gap + len + inserted */
__device__ __forceinline__ void emit_nal(const uint8_t* __restrict__ rbsp, const hbs_nal_entry* __restrict__ idx, uint64_t k,
                                         uint64_t gap, uint64_t base, uint64_t total,
                                         uint8_t* __restrict__ out, uint64_t out_cap, hbs_nal_entry* __restrict__ idx_out,
                                         uint32_t* __restrict__ err, int lane)
{
    struct __attribute__((packed, aligned(1))) U16 { u32x4 v; };
    const uint64_t begin = idx[k].rbsp_off;
    const uint32_t len = idx[k].rbsp_len;
    const uint32_t nrows = (len + 1023u) / 1024u;
    const uint64_t nal_start = base + gap;
    const uint64_t nal_end = base + total;
    const bool fits = nal_end <= out_cap;
    if (fits) {
        uint32_t ins = 0;                                 /* bytes inserted so far, wave-uniform */
        uint32_t e_prev = 0xFFFFFFFFu;
        u32x4 qn = load_nal_chunk(rbsp, begin, len, 16u * (uint32_t)lane);
        for (uint32_t r = 0; r < nrows; ++r) {
            const u32x4 q = qn;
            const uint32_t off = 1024u * r + 16u * (uint32_t)lane;
            qn = load_nal_chunk(rbsp, begin, len, off + 1024u);
            const RowFlags f = row_flags(q, e_prev, (uint32_t)__builtin_amdgcn_readlane((int)qn.x, 0), off, len);
            uint8_t* dst = out + nal_start + off + ins;
            uint32_t c = 0;
            if (f.mask != 0) {
                ExactChunk ec;
                ec.mask = 0; ec.nb = 0;
                if (f.mine) { ec = exact_from_regs(q, f.xp, rbsp, begin, len, off); c = (uint32_t)__builtin_popcount(ec.mask); }
                uint32_t tot;
                dst += wave_excl_scan_u32(c, lane, tot);
                ins += tot;
                if (f.mine) store_exact(dst, ec);
            }
            if (!f.mine) {
                if (off + 16u <= len) {
                    reinterpret_cast<U16*>(dst)->v = q;
                } else if (off < len) {                    /* the NAL's last, partial chunk */
                    const uint32_t nb = len - off;
#pragma unroll 1
                    for (uint32_t b = 0; b < nb; ++b) {
                        const uint32_t w = (b >> 2) == 0 ? q.x : (b >> 2) == 1 ? q.y : (b >> 2) == 2 ? q.z : q.w;
                        dst[b] = (uint8_t)(w >> (8u * (b & 3u)));
                    }
                }
            }
            e_prev = (uint32_t)__builtin_amdgcn_readlane((int)q.w, 63);
        }
    }
    if (lane == 0) {
        if (fits) {
            for (uint64_t i = base; i + 1 < nal_start; ++i) out[i] = 0;      /* zero_byte / leading zeros */
            if (gap) out[nal_start - 1] = 1;
        }
        if (idx_out) {
            hbs_nal_entry e;
            e.start = nal_start; e.end = nal_end;
            e.rbsp_off = begin; e.rbsp_len = len; e.status = 0;
            idx_out[k] = e;
        }
        if (!fits) atomicMax(err, (uint32_t)(-HBS_E_CAPACITY));
    }
}

__global__ __launch_bounds__(256)
void k3_emit(const uint8_t* __restrict__ rbsp, const hbs_nal_entry* __restrict__ idx, uint64_t n, int gap_mode,
             const unsigned long long* __restrict__ nal_total, const unsigned long long* __restrict__ out_off,
             uint8_t* __restrict__ out, uint64_t out_cap, hbs_nal_entry* __restrict__ idx_out, uint32_t* __restrict__ err,
             const uint32_t* __restrict__ probe, const uint32_t* __restrict__ vflag)
{
    if (!three_steps_run(probe, vflag) || index_bad(vflag)) return;
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t k = wave; k < n; k += nwaves)
        emit_nal(rbsp, idx, k, gap_of(idx, k, gap_mode), out_off[k], nal_total[k], out, out_cap, idx_out, err, lane);
}

/* ---- arenas of tiny NALs: a LANE per NAL (round 4) ---------------------------------------------------
 * NALs below ~200 bytes -- one slice per CTU row at a low bitrate -- have more than 1024 starts per 192 KiB, so the arena
 * tiles do not apply, and the kernels by NALs give every NAL a wavefront (or a 12 KiB slot): a 64-byte NAL used one lane in
 * sixteen (0.008 of the HBM peak at 64 bytes, 0.04 at 384).  Here every lane walks its own NAL byte by byte, rbsp_to_nal
 * as it is written (h264_nal.c:92-132): pass 1 the bytes that go in, an exclusive scan of the NALs' output sizes (the
 * three steps' own), pass 2 the bytes -- read as unaligned dwords, written as unaligned dwords from a small register
 * buffer.  Picked on the HOST (mean NAL size = rbsp_bytes / n below kTinyMeanBytes): no device-side gate, no probe. */
constexpr uint64_t kTilesMinMeanBytes = 224;     /* ... from this mean up the arena tiles are tried first (878 starts per tile on average; 1024 is their limit) */
constexpr uint64_t kTinyMeanBytes = 448;         /* below this mean the host takes this route (arena tiles first from kTilesMinMeanBytes up, see launch_emit_annexb) */

/* a lane's NAL, 16 bytes a load (a dword a load fetched every 128-byte line thirty-two times: 64 lanes x 32 wavefronts of
 * lines do not stay in a 16 KiB L1) */
template <class F>
__device__ __forceinline__ void tiny_bytes(const uint8_t* __restrict__ arena, uint64_t arena_bytes, uint64_t begin, uint32_t len, F&& f)
{
    struct __attribute__((packed, aligned(1))) U16 { u32x4 v; };
    /* 64 bytes a step, the four loads issued together and the next step's in flight (one load in flight per lane: a NAL of 384
     * bytes was 24 memory round trips, 7.5 ms over 2 GiB; four: 4.4 ms).  A chunk may reach past the NAL's end -- the bytes
     * behind it are not looked at -- but not past the arena's. */
    auto load4 = [&](u32x4* q, uint32_t i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint64_t at = begin + i + 16u * (uint32_t)k;
            q[k] = u32x4{0u, 0u, 0u, 0u};
            if (i + 16u * (uint32_t)k < len) {
                if (at + 16u <= arena_bytes) q[k] = reinterpret_cast<const U16*>(arena + at)->v;
                else q[k] = load_chunk_guarded(arena, at, arena_bytes);
            }
        }
    };
    u32x4 cur[4], nxt[4];
    load4(cur, 0u);
#pragma unroll 1
    for (uint32_t i = 0; i < len; i += 64u) {
        if (i + 64u < len) load4(nxt, i + 64u);
#pragma unroll 1
        for (int k = 0; k < 4; ++k) {
            const u32x4 c = k == 0 ? cur[0] : k == 1 ? cur[1] : k == 2 ? cur[2] : cur[3];
            const uint32_t at = i + 16u * (uint32_t)k;
            if (at >= len) break;
            const uint32_t nb = len - at < 16u ? len - at : 16u;
            uint32_t w[4] = {c.x, c.y, c.z, c.w};
            if (nb == 16u) {
#pragma unroll
                for (int d = 0; d < 4; ++d)
#pragma unroll
                    for (int b = 0; b < 4; ++b) { f(w[d] & 0xFFu); w[d] >>= 8; }
            } else {
#pragma unroll 1
                for (uint32_t b = 0; b < nb; ++b) {
                    f(w[0] & 0xFFu);
                    w[0] = (w[0] >> 8) | (w[1] << 24); w[1] = (w[1] >> 8) | (w[2] << 24); w[2] = (w[2] >> 8) | (w[3] << 24); w[3] >>= 8;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) cur[k] = nxt[k];
    }
}

__device__ __forceinline__ uint32_t tiny_count(const uint8_t* __restrict__ arena, uint64_t arena_bytes, uint64_t begin, uint32_t len)
{
    uint32_t ins = 0, count = 0;
    tiny_bytes(arena, arena_bytes, begin, len, [&](uint32_t v) {
        if (count == 2u && v <= 3u) { ++ins; count = 0u; }
        count = v == 0u ? count + 1u : 0u;
    });
    return ins;
}

__global__ __launch_bounds__(256)
void k3_count_tiny(const uint8_t* __restrict__ rbsp, uint64_t rbsp_bytes, const hbs_nal_entry* __restrict__ idx, uint64_t n, int gap_mode,
                   unsigned long long* __restrict__ nal_total, const uint32_t* __restrict__ vflag)
{
    if (index_bad(vflag) || tile_path_on(vflag) || group_path_on(vflag)) return;   /* (the arena tiles or the group kernel, in front, do the call when the index allows them) */
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t len = idx[k].rbsp_len;
        nal_total[k] = gap_of(idx, k, gap_mode) + len + tiny_count(rbsp, rbsp_bytes, idx[k].rbsp_off, len);
    }
}

/* bytes out through a 128-bit buffer: one (unaligned) 16-byte store whenever sixteen are there */
struct TinyOut {
    uint8_t* dst; uint64_t lo, hi; uint32_t have;
    __device__ __forceinline__ void put(uint32_t v)
    {
        struct __attribute__((packed, aligned(1))) U16 { u32x4 v; };
        if (have < 8u) lo |= (uint64_t)v << (8u * have); else hi |= (uint64_t)v << (8u * (have - 8u));
        if (++have == 16u) {
            u32x4 q; q.x = (uint32_t)lo; q.y = (uint32_t)(lo >> 32); q.z = (uint32_t)hi; q.w = (uint32_t)(hi >> 32);
            reinterpret_cast<U16*>(dst)->v = q;
            dst += 16; lo = hi = 0ull; have = 0u;
        }
    }
    __device__ __forceinline__ void flush()
    {
        for (uint32_t b = 0; b < have; ++b) dst[b] = (uint8_t)((b < 8u ? lo >> (8u * b) : hi >> (8u * (b - 8u))));
        dst += have; have = 0u; lo = hi = 0ull;
    }
};

__global__ __launch_bounds__(256)
void k3_emit_tiny(const uint8_t* __restrict__ rbsp, uint64_t rbsp_bytes, const hbs_nal_entry* __restrict__ idx, uint64_t n, int gap_mode,
                  const unsigned long long* __restrict__ nal_total, const unsigned long long* __restrict__ out_off,
                  uint8_t* __restrict__ out, uint64_t out_cap, hbs_nal_entry* __restrict__ idx_out, uint32_t* __restrict__ err,
                  const uint32_t* __restrict__ vflag)
{
    if (index_bad(vflag) || tile_path_on(vflag) || group_path_on(vflag)) return;
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t begin = idx[k].rbsp_off;
        const uint32_t len = idx[k].rbsp_len;
        const uint64_t gap = gap_of(idx, k, gap_mode), base = out_off[k], total = nal_total[k];
        const uint64_t nal_start = base + gap, nal_end = base + total;
        if (idx_out) {
            hbs_nal_entry e;
            e.start = nal_start; e.end = nal_end; e.rbsp_off = begin; e.rbsp_len = len; e.status = 0;
            idx_out[k] = e;
        }
        if (nal_end > out_cap) { atomicMax(err, (uint32_t)(-HBS_E_CAPACITY)); continue; }
        TinyOut o;
        o.dst = out + base; o.lo = o.hi = 0ull; o.have = 0u;
        for (uint64_t g = 0; g + 1 < gap; ++g) o.put(0u);                        /* zero_byte / leading zeros, then 01 */
        if (gap) o.put(1u);
        uint32_t count = 0;
        tiny_bytes(rbsp, rbsp_bytes, begin, len, [&](uint32_t v) {
            if (count == 2u && v <= 3u) { o.put(3u); count = 0u; }
            o.put(v);
            count = v == 0u ? count + 1u : 0u;
        });
        o.flush();
    }
}

bool emit_takes_tiny_path(uint64_t n, uint64_t rbsp_bytes, int two_pass, int tiles)
{
    /* automatic mode only (a pinned path stays pinned: the tests and the soak pin every path on every arena) */
    return two_pass < 0 && tiles == 1 && n > 256 && rbsp_bytes / n < kTinyMeanBytes;      /* (256: k3_small takes what is below) */
}

/* ---- a handful of small NALs: the whole call in one launch of one workgroup --------------------------
 * (the general path is a dozen launches: what a legacy rbsp_to_nal() of one parameter set or a short
 * batch pays for is their latency).  Wavefront w takes NALs w, w + 4, ...: sizes, a scan across the
 * workgroup, the bytes; thread 0 writes the summary. */
constexpr uint64_t kEmitSmallNals = 256, kEmitSmallBytes = 32768;

__global__ __launch_bounds__(256)
void k3_small(const uint8_t* __restrict__ rbsp, uint64_t rbsp_bytes, const hbs_nal_entry* __restrict__ idx, uint32_t n, int gap_mode,
              uint8_t* __restrict__ out, uint64_t out_cap, hbs_nal_entry* __restrict__ idx_out,
              uint32_t* __restrict__ err, hbs_summary* __restrict__ sum)
{
    __shared__ unsigned long long tot[kEmitSmallNals], off[kEmitSmallNals], wsum[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) *err = 0;
    /* every entry inside the caller's buffer, or nothing is read through the index (HBS_E_ARG) */
    const bool outside = threadIdx.x < n && (idx[threadIdx.x].rbsp_off > rbsp_bytes || idx[threadIdx.x].rbsp_len > rbsp_bytes - idx[threadIdx.x].rbsp_off);
    if (__syncthreads_or(outside ? 1 : 0)) {
        if (threadIdx.x == 0) {
            sum->nal_count = n; sum->nal_found = n; sum->rbsp_bytes = rbsp_bytes; sum->stream_bytes = 0;
            sum->stop_reason = -1; sum->error = HBS_E_ARG;
            sum->reserved[0] = sum->reserved[1] = sum->reserved[2] = 0;
        }
        return;
    }
    for (uint32_t k = (uint32_t)wv; k < n; k += 4) {
        const uint32_t ins = count_nal(rbsp, idx[k].rbsp_off, idx[k].rbsp_len, lane);
        if (lane == 0) tot[k] = gap_of(idx, k, gap_mode) + idx[k].rbsp_len + ins;
    }
    __syncthreads();
    unsigned long long total;
    const unsigned long long mine = threadIdx.x < n ? tot[threadIdx.x] : 0ull;
    const unsigned long long before = block_excl_scan_u64(mine, wsum, total);
    if (threadIdx.x < n) off[threadIdx.x] = before;
    __syncthreads();
    for (uint32_t k = (uint32_t)wv; k < n; k += 4)
        emit_nal(rbsp, idx, k, gap_of(idx, k, gap_mode), off[k], tot[k], out, out_cap, idx_out, err, lane);
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        sum->nal_count = n; sum->nal_found = n; sum->rbsp_bytes = rbsp_bytes;
        sum->stream_bytes = total;
        sum->stop_reason = n ? -1 : 0; sum->error = -(int32_t)atomicMax(err, 0u);       /* read where the atomics landed */
        sum->reserved[0] = sum->reserved[1] = sum->reserved[2] = 0;
    }
}

bool emit_takes_small_path(uint64_t n, uint64_t rbsp_bytes, int two_pass)
{
    return two_pass < 0 && n != 0 && n <= kEmitSmallNals && rbsp_bytes <= kEmitSmallBytes;
}

/* ---- single pass: count, look-back, emit ---------------------------------------------------
 * The unit of work is an ITEM: rows [12 s, 12 s + 12) of a NAL ("segment"; see "work items" below).
 * A workgroup takes a group of kEmitGroup consecutive items by ticket, kEmitSlots per wavefront
 * (fat wavefronts, 2 workgroups per CU, as in K12).  A wavefront loads the up to kEmitRows rows of
 * 1 KiB of each of its items into registers in one burst (lane l = 16 bytes at 16 l of each row)
 * and counts the bytes rbsp_to_nal would insert; wavefront 0 publishes the group's output size
 * and looks back over the groups in front of it (decoupled look-back: one 64-bit word per group,
 * value << 2 | status, status 1 = size of the group, 2 = size of everything up to and including
 * it; 256 groups per step); then every wavefront writes its items from the registers it still
 * holds.  HBM traffic: RBSP read once, stream written once.  Tickets are taken by running
 * workgroups only and a workgroup finishes its groups in ticket order, so every group a look-back
 * waits for is being worked on.  (Look-back units as small as one wavefront's share were tried:
 * with 2048 of them in flight the prefix frontier cannot advance fast enough, 3x slower.)
 *
 * Rows that chunk_flag() clears are copied straight from registers; a row with a flagged chunk
 * goes through a rolled loop that runs the byte-exact rules of hbs_emit.h on memory, so the
 * unrolled code stays small. */
#ifdef HBS_PHASE_TIMING
/* diagnostic build only (make diag, scripts/emit_phase.py): bit 0 skips the stores, bit 1 the look-back wait, bit 2 the count */
__device__ int g_k3_exp = 0;
extern "C" int hbs_debug_k3_exp(int v) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_k3_exp), &v, sizeof(int)); }
#define HBS_K3_EXP(bit) ((g_k3_exp >> (bit)) & 1)
__device__ unsigned long long g_phase_cycles_emit[1024][8];
extern "C" int hbs_debug_phase_cycles_emit(unsigned long long* host_out /* [1024][8] */)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_phase_cycles_emit), sizeof(unsigned long long) * 1024 * 8);
}
#define HBS3_T_DECL unsigned long long t_prev = __builtin_amdgcn_s_memtime(), t_acc[8] = {0,0,0,0,0,0,0,0};
#define HBS3_T_MARK(i) { __builtin_amdgcn_s_waitcnt(0); const unsigned long long t_now = __builtin_amdgcn_s_memtime(); t_acc[i] += t_now - t_prev; t_prev = t_now; }
#define HBS3_T_FLUSH if (threadIdx.x == 0 && blockIdx.x < 1024) { for (int i = 0; i < 8; ++i) g_phase_cycles_emit[blockIdx.x][i] = t_acc[i]; }
#else
#define HBS_K3_EXP(bit) 0
#define HBS3_T_DECL
#define HBS3_T_MARK(i)
#define HBS3_T_FLUSH
#endif

#ifndef HBS_EMIT_SLOTS
#define HBS_EMIT_SLOTS 3
#define HBS_EMIT_ROWS 12
#endif
#ifndef HBS_EMIT_WGS
#define HBS_EMIT_WGS 2                            /* workgroups per CU the register budget is set for */
#endif
constexpr int kEmitSlots = HBS_EMIT_SLOTS;        /* NALs a wavefront works on at a time */
constexpr int kEmitRows = HBS_EMIT_ROWS;          /* rows of 1 KiB per slot held in registers */
constexpr int kEmitGroup = 4 * kEmitSlots;        /* NALs per workgroup and ticket */
constexpr uint32_t kEmitSpinLimit = 1u << 26;

struct __attribute__((packed, aligned(1))) Chunk16 { u32x4 v; };
/* streaming accesses of the single pass: the arena is read once and the stream written once (hbs_wave.h has the measurements) */
#ifndef HBS_K3_NT
#define HBS_K3_NT 1
#endif
typedef const __attribute__((address_space(1))) u32x4_u1* global_u32x4_u1_ptr;
__device__ __forceinline__ u32x4 k3_load16(const uint8_t* p)
{
#if HBS_K3_NT
    return __builtin_nontemporal_load((global_u32x4_u1_ptr)(uintptr_t)p);
#else
    return reinterpret_cast<const Chunk16*>(p)->v;
#endif
}
__device__ __forceinline__ void k3_store16(uint8_t* p, u32x4 v)
{
#if HBS_K3_NT
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4_u1*>(p));
#else
    reinterpret_cast<Chunk16*>(p)->v = v;
#endif
}

/* rows [r0, r0 + kEmitRows) of a NAL.  Unpredicated loads: a lane whose chunk starts behind the
 * NAL's last chunk reads that last chunk again (and never uses it), and the last chunk may reach up
 * to 15 bytes past the NAL -- harmless (flags are conservative, exact code is bounded by len) as
 * long as the bytes exist: reads stay below `arena`. */
__device__ __forceinline__ void load_rows(u32x4 (&R)[kEmitRows], const uint8_t* __restrict__ rbsp, uint64_t arena,
                                          uint64_t begin, uint32_t len, uint32_t r0, int lane)
{
    if (len == 0) return;                                                                  /* wave-uniform */
    const uint32_t last_off = (len - 1u) & ~15u;
    const uint64_t room = arena - begin;                                                   /* bytes readable from the NAL's first byte on */
    const uint8_t* const base = rbsp + begin;
    if (room >= 16u) {
        const uint64_t safe = room - 16u;
        const uint32_t lim = safe < (uint64_t)last_off ? (uint32_t)safe : last_off;        /* wave-uniform */
#pragma unroll
        for (int r = 0; r < kEmitRows; ++r) {
            if (1024u * (r0 + (uint32_t)r) <= last_off) {                                  /* wave-uniform: the row exists */
                /* launder_lane: every row computes its offset where it needs it; kept across the slots
                 * and phases (they are all the same expression) they would fill the register file */
                const uint32_t off = 1024u * (r0 + (uint32_t)r) + 16u * (uint32_t)launder_lane(lane);
                R[r] = k3_load16(base + (off < lim ? off : lim));
#if HBS_K3F_LOAD_DEPTH >= 0
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(HBS_K3F_LOAD_DEPTH) : "memory");
#endif
            }
        }
    }
    if (begin + last_off + 16u > arena && (last_off >> 10) >= r0 && (last_off >> 10) < r0 + kEmitRows) {   /* wave-uniform, rare */
        const u32x4 t = load_nal_chunk(rbsp, begin, len, last_off);
#pragma unroll
        for (int r = 0; r < kEmitRows; ++r)
            if ((last_off >> 10) == r0 + (uint32_t)r && (uint32_t)lane == ((last_off >> 4) & 63u)) R[r] = t;
    }
}

/* flags of one batch of rows held in R: rowmask = rows with a flagged chunk (they take the exact
 * path; wave-uniform), myflags bit r = my chunk of row r is flagged */
__device__ __forceinline__ void flag_batch(const u32x4 (&R)[kEmitRows], uint32_t len, uint32_t r0, int lane,
                                           uint32_t& e_prev, uint32_t& rowmask, uint32_t& myflags)
{
    rowmask = 0; myflags = 0;
#pragma unroll
    for (int r = 0; r < kEmitRows; ++r) {
        const uint32_t row_lo = 1024u * (r0 + (uint32_t)r);
        if (row_lo < len) {                                    /* wave-uniform */
            const uint32_t off = row_lo + 16u * (uint32_t)launder_lane(lane);
            /* a 03 goes in front of byte i only if bytes i-2, i-1 are zero: the chunk behind does not matter */
            const uint32_t xp = from_prev_lane(R[r].w, e_prev);
            bool mine = off < len && chunk_flag(xp, R[r].x, R[r].y, R[r].z, R[r].w, 0xFFFFFFFFu);
            if (__builtin_popcountll(__ballot(mine)) > kExactFlagMin) mine = mine && chunk_pattern_any_dev(xp, R[r].x, R[r].y, R[r].z, R[r].w, 0xFFFFFFFFu);
            if (__ballot(mine) != 0) rowmask |= 1u << r;
            myflags |= (mine ? 1u : 0u) << r;
            e_prev = (uint32_t)__builtin_amdgcn_readlane((int)R[r].w, 63);
        }
#ifndef HBS_K3_FLAG_INTERLEAVE
#define HBS_K3_FLAG_INTERLEAVE 1
#endif
        if ((r % HBS_K3_FLAG_INTERLEAVE) == HBS_K3_FLAG_INTERLEAVE - 1) __builtin_amdgcn_sched_barrier(0);
    }
}

/* bytes rbsp_to_nal inserts into the batch: its flagged chunks, exactly, each fetched again (one 16-byte load) in a rolled
 * loop.  (Counting them inside flag_batch from the registers that are at hand -- no load at all -- put 36 copies of the walk
 * into the unrolled flag pass and ran 8 % slower: 9.2-9.6 against 8.6-8.9 ms on the 16 GiB arena.) */
__device__ __forceinline__ uint32_t count_batch(const uint8_t* __restrict__ rbsp, uint64_t begin, uint32_t len, uint32_t r0, int lane,
                                                uint32_t rowmask, uint32_t myflags)
{
    uint32_t ins = 0;
#pragma unroll 1
    for (uint32_t rm = rowmask; rm != 0; rm &= rm - 1u) {
        const uint32_t r = (uint32_t)__builtin_ctz(rm);
        const uint32_t off = 1024u * (r0 + r) + 16u * (uint32_t)lane;
        uint32_t c = 0;
        if ((myflags >> r) & 1u) c = (uint32_t)__builtin_popcount(exact_chunk(rbsp, begin, len, off).mask);
        ins += wave_sum_u32(c);
    }
    return ins;
}

__device__ __forceinline__ void store_bytes(uint8_t* dst, const u32x4& q, uint32_t nb)
{
#pragma unroll 1
    for (uint32_t i = 0; i < nb; ++i) {
        const uint32_t w = (i >> 2) == 0 ? q.x : (i >> 2) == 1 ? q.y : (i >> 2) == 2 ? q.z : q.w;
        dst[i] = (uint8_t)(w >> (8u * (i & 3u)));
    }
}

/* writes the batch at dst0 + (offset in the NAL) + (bytes inserted in front); ins = bytes inserted so far */
__device__ __forceinline__ void emit_batch(const u32x4 (&R)[kEmitRows], const uint8_t* __restrict__ rbsp, uint64_t begin, uint32_t len,
                                           uint32_t r0, int lane, uint32_t rowmask, uint32_t myflags, uint32_t& ins, uint8_t* dst0)
{
    uint32_t rowins = ins;                /* lane r: bytes inserted in front of row r */
#pragma unroll 1
    for (uint32_t rm = rowmask; rm != 0; rm &= rm - 1u) {      /* rows with flagged chunks: exact rules, from memory */
        const uint32_t r = (uint32_t)__builtin_ctz(rm);
        const uint32_t off = 1024u * (r0 + r) + 16u * (uint32_t)lane;
        const bool mine = ((myflags >> r) & 1u) != 0;
        uint32_t c = 0, tot;
        ExactChunk ec;
        ec.mask = 0; ec.nb = 0;
        if (mine) { ec = exact_chunk(rbsp, begin, len, off); c = (uint32_t)__builtin_popcount(ec.mask); }
        uint8_t* dst = dst0 + off + ins + wave_excl_scan_u32(c, lane, tot);
        if (mine) {
            store_exact(dst, ec);
        } else if (off < len) {
            const u32x4 q = load_nal_chunk(rbsp, begin, len, off);
            if (off + 16u <= len) reinterpret_cast<Chunk16*>(dst)->v = q;
            else store_bytes(dst, q, len - off);               /* the NAL's last, partial chunk */
        }
        if ((uint32_t)lane > r) rowins += tot;
        ins += tot;
    }
    const uint32_t last_off = len ? ((len - 1u) & ~15u) : 0u;
    const bool partial = (len & 15u) != 0;
#pragma unroll
    for (int r = 0; r < kEmitRows; ++r) {
        const uint32_t row_lo = 1024u * (r0 + (uint32_t)r);
        if (row_lo < len && !((rowmask >> r) & 1u)) {          /* wave-uniform: no flagged chunk in the row */
            const uint32_t off = row_lo + 16u * (uint32_t)launder_lane(lane);
            uint8_t* dst = dst0 + off + (uint32_t)__builtin_amdgcn_readlane((int)rowins, r);
            if (off + 16u <= len) k3_store16(dst, R[r]);
            if (partial && (last_off >> 10) == r0 + (uint32_t)r) {                    /* wave-uniform: the NAL's last row */
                if (off == last_off) store_bytes(dst, R[r], len - last_off);
            }
        }
#if HBS_K3F_COPY_DEPTH >= 0
        /* at most that many stores of a wavefront in flight: a short memory queue on the CU keeps the other workgroups'
         * look-back polls quick (hbs_scan4.hip, round 3) */
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(HBS_K3F_COPY_DEPTH) : "memory");
#endif
        __builtin_amdgcn_sched_barrier(0);                     /* one row at a time: interleaving rows only costs registers */
    }
}

/* bytes of the output in front of group g (wavefront 0, all lanes); 256 groups per step:
 * lane l looks at the groups at distance l, 64 + l, 128 + l, 192 + l */
/* a look-back word: status in bits 0-1 (0 nothing yet, 1 the tile's own bytes, 2 everything up to and including it), the bytes
 * above, and -- k3_tiles' dense tiles only -- a tag in bits 60-63 that says what count of zeros the tile leaves behind
 * (k3_dense_tile): the sums below ignore it */
constexpr int kDescTagShift = 60;
constexpr unsigned long long kDescValueMask = (1ull << kDescTagShift) - 1ull;
__device__ __forceinline__ unsigned long long k3_look_back(const unsigned long long* __restrict__ desc, uint64_t g, int lane, uint32_t* err)
{
    unsigned long long prefix = 0;
    uint64_t pos = g;                                         /* groups [0, pos) are still to be accounted for */
    while (pos > 0) {
        unsigned long long v[4] = {0, 0, 0, 0};
        uint32_t spins = 0;
        int stop_q = 4, stop_l = 0;                           /* nearest group with an inclusive prefix: quarter, lane */
        for (;;) {
            bool all = true;
            stop_q = 4;
#pragma unroll
            for (int q = 3; q >= 0; --q) {
                const uint64_t dist = (uint64_t)(64 * q + lane);
                const bool valid = dist < pos;
                if (valid && (v[q] & 3ull) == 0) v[q] = ld_desc3(desc + (pos - 1 - dist));
                const uint64_t full = __ballot(valid && (v[q] & 3ull) == 2ull);
                if (full) { stop_q = q; stop_l = (int)__builtin_ctzll(full); }
            }
            /* everything nearer than the stop (or the whole window) must be there */
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint64_t dist = (uint64_t)(64 * q + lane);
                const bool needed = dist < pos && (q < stop_q || (q == stop_q && lane <= stop_l));
                if (__ballot(needed && (v[q] & 3ull) == 0) != 0) all = false;
            }
            if (all) break;
            if (++spins > kEmitSpinLimit) { if (lane == 0) atomicMax(err, (uint32_t)(-HBS_E_TIMEOUT)); return prefix; }
            __builtin_amdgcn_s_sleep(1);
        }
        unsigned long long x = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint64_t dist = (uint64_t)(64 * q + lane);
            if (dist < pos && (q < stop_q || (q == stop_q && lane <= stop_l))) x += (v[q] & kDescValueMask) >> 2;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
        prefix += x;
        if (stop_q < 4) break;
        pos = pos > 256 ? pos - 256 : 0;
    }
    return prefix;
}

__device__ __forceinline__ uint64_t bcast64(uint64_t v, int src_lane)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src_lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src_lane);
    return ((uint64_t)hi << 32) | lo;
}

/* ---- work items --------------------------------------------------------------------------
 * A slot holds at most kEmitRows KiB, so the unit of work is a SEGMENT: rows [12 s, 12 s + 12) of a
 * NAL.  A NAL of any length is ceil(len / 12 KiB) consecutive items, spread over wavefronts and
 * workgroups like any other items; the look-back that places NALs also carries the bytes inserted
 * into the segments in front.  (A coded picture is routinely 50-500 KiB: without this one
 * wavefront would stream it alone while everything behind it waits for its size.)
 * When every NAL fits a slot -- items == NALs -- the item list is not built at all. */
constexpr uint32_t kEmitSegBytes = (uint32_t)kEmitRows * 1024u;
constexpr int kItemSegBits = 20;                             /* rbsp_len < 2^32: fewer than 2^19 segments */

__global__ void k3_seg_count(const hbs_nal_entry* __restrict__ idx, uint64_t n, unsigned long long* __restrict__ segs,
                             const uint32_t* __restrict__ probe, const uint32_t* __restrict__ tflag)
{
    if (emit_skip(probe, kWhenSparse, tflag)) return;
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t len = idx[k].rbsp_len;
        segs[k] = len <= kEmitSegBytes ? 1ull : (unsigned long long)((len + kEmitSegBytes - 1u) / kEmitSegBytes);
    }
}

/* the automatic mode's first step of BOTH fall-back chains in one launch (round 4: every launch that rules itself out on the device
 * still costs a few us of the call): segments per NAL for the kernel by NALs, or bytes per NAL for the three steps -- whichever
 * chain the probe and the tile kernel's eligibility leave to run, if any.  Both write nal_total; the chains exclude each other. */
__global__ __launch_bounds__(256)
void k3_sizes(const uint8_t* __restrict__ rbsp, const hbs_nal_entry* __restrict__ idx, uint64_t n, int gap_mode,
              unsigned long long* __restrict__ nal_total, const uint32_t* __restrict__ probe, const uint32_t* __restrict__ vflag)
{
    if (three_steps_run(probe, vflag)) {
        if (index_bad(vflag)) return;
        const int lane = threadIdx.x & 63;
        const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
        const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
        for (uint64_t k = wave; k < n; k += nwaves) {
            const uint64_t begin = idx[k].rbsp_off;
            const uint32_t len = idx[k].rbsp_len;
            const uint32_t ins = count_nal(rbsp, begin, len, lane);
            if (lane == 0) nal_total[k] = gap_of(idx, k, gap_mode) + len + ins;
        }
        return;
    }
    if (emit_skip(probe, kWhenSparse, vflag)) return;
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t len = idx[k].rbsp_len;
        nal_total[k] = len <= kEmitSegBytes ? 1ull : (unsigned long long)((len + kEmitSegBytes - 1u) / kEmitSegBytes);
    }
}

__global__ void k3_expand(const unsigned long long* __restrict__ segs, const unsigned long long* __restrict__ item_base, uint64_t n,
                          const unsigned long long* __restrict__ n_items, unsigned long long* __restrict__ items, uint64_t items_cap,
                          const uint32_t* __restrict__ probe, const uint32_t* __restrict__ tflag)
{
    if (emit_skip(probe, kWhenSparse, tflag)) return;
    if (*n_items == n || *n_items > items_cap) return;       /* identity: nothing to build; too many: the main kernel reports it */
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        const unsigned long long base = item_base[k], m = segs[k];
        for (unsigned long long sgm = 0; sgm < m; ++sgm) items[base + sgm] = (k << kItemSegBits) | sgm;
    }
}

/* what lane j knows about slot j of a wavefront's share of a group */
struct SlotEntry { uint64_t k, begin, gap; uint32_t len, r0; };
__device__ __forceinline__ SlotEntry fetch_entries(const hbs_nal_entry* __restrict__ idx, uint64_t n, int gap_mode,
                                                   const unsigned long long* __restrict__ items, uint64_t n_items,
                                                   uint64_t i0, int lane)
{
    SlotEntry e; e.k = ~0ull; e.begin = 0; e.gap = 0; e.len = 0; e.r0 = 0;
    const uint64_t i = i0 + (uint64_t)lane;
    if (lane < kEmitSlots && i < n_items) {
        uint64_t k = i, sgm = 0;
        if (n_items != n) { const unsigned long long it = items[i]; k = it >> kItemSegBits; sgm = it & ((1ull << kItemSegBits) - 1ull); }
        e.k = k; e.begin = idx[k].rbsp_off; e.len = idx[k].rbsp_len; e.r0 = (uint32_t)sgm * (uint32_t)kEmitRows;
        e.gap = sgm == 0 ? gap_of(idx, k, gap_mode) : 0ull;
    }
    return e;
}

struct Lds3 {
    unsigned long long tot[kEmitGroup];      /* bytes item j of the group takes in the output: gap + payload + inserted */
    unsigned long long off[kEmitGroup];      /* where it starts */
    uint32_t ticket;
};

__global__ __launch_bounds__(256, HBS_EMIT_WGS)
void k3_fused(const uint8_t* __restrict__ rbsp, uint64_t arena, const hbs_nal_entry* __restrict__ idx, uint64_t n, int gap_mode,
              const unsigned long long* __restrict__ items, const unsigned long long* __restrict__ n_items_ptr, uint64_t items_cap,
              unsigned long long* __restrict__ desc, uint32_t* __restrict__ ticket,
              uint8_t* __restrict__ out, uint64_t out_cap, hbs_nal_entry* __restrict__ idx_out,
              unsigned long long* __restrict__ total, uint32_t* __restrict__ err, const uint32_t* __restrict__ probe,
              const uint32_t* __restrict__ tflag, const uint32_t* __restrict__ vflag)
{
    if ((probe && emit_probe_dense(probe)) || tile_path_done(tflag) || index_bad(vflag)) return;      /* the tile kernel, in front of this one, did it */
    __shared__ Lds3 l;
    const int lane0 = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint64_t n_items = *n_items_ptr;
    if (n_items > items_cap) {                                /* only when the NALs add up to more than the output holds */
        if (blockIdx.x == 0 && threadIdx.x == 0) { atomicMax(err, (uint32_t)(-HBS_E_CAPACITY)); *total = 0; }
        return;
    }
    const uint64_t ngroups = (n_items + kEmitGroup - 1) / kEmitGroup;
    HBS3_T_DECL
    for (;;) {
        const int lane = launder_lane(lane0);                 /* keeps lane-constant values from being hoisted out of the loop (and spilled) */
        /* The ticket is taken as late as possible -- when the previous group's stores have been issued.
         * A workgroup that sits on a ticket it has not started yet is what the look-backs of everybody
         * behind it wait for: fetching the next ticket while the current group is written was 15 %
         * slower, fetching it before the look-back 50 %. */
        __syncthreads();                                      /* the previous group is done with l */
        if (threadIdx.x == 0) l.ticket = atomicAdd(ticket, 1u);
        __syncthreads();
        const uint64_t g = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)l.ticket);
        if (g >= ngroups) break;
        HBS3_T_MARK(0)
        /* until the group's size is out, this workgroup is what its successors wait for */
        __builtin_amdgcn_s_setprio(3);
        const SlotEntry ent = fetch_entries(idx, n, gap_mode, items, n_items, g * kEmitGroup + (uint64_t)(kEmitSlots * wv), lane);
        HBS3_T_MARK(1)

        /* 1. my items: load, count */
        u32x4 R[kEmitSlots][kEmitRows];
        uint64_t nal[kEmitSlots], begin[kEmitSlots], gap[kEmitSlots];
        uint32_t len[kEmitSlots], r0[kEmitSlots], ins[kEmitSlots], rowmask[kEmitSlots], myflags[kEmitSlots];
#pragma unroll
        for (int j = 0; j < kEmitSlots; ++j) {
            nal[j] = bcast64(ent.k, j); begin[j] = bcast64(ent.begin, j); gap[j] = bcast64(ent.gap, j);
            len[j] = (uint32_t)__builtin_amdgcn_readlane((int)ent.len, j);
            r0[j] = (uint32_t)__builtin_amdgcn_readlane((int)ent.r0, j);
            if (nal[j] != ~0ull) load_rows(R[j], rbsp, arena, begin[j], len[j], r0[j], lane);
        }
        HBS3_T_MARK(2)
#pragma unroll
        for (int j = 0; j < kEmitSlots; ++j) {
            ins[j] = 0; rowmask[j] = 0; myflags[j] = 0;
            unsigned long long tot = 0;
            if (nal[j] != ~0ull) {
                /* a NAL starts with count = 0; a later segment continues behind the dword in front of it */
                uint32_t e_prev = 0xFFFFFFFFu;
                if (r0[j] != 0) e_prev = reinterpret_cast<const Chunk16*>(rbsp + begin[j] + 1024ull * r0[j] - 16u)->v.w;
                if (!HBS_K3_EXP(2)) {
                    flag_batch(R[j], len[j], r0[j], lane, e_prev, rowmask[j], myflags[j]);
                    ins[j] = count_batch(rbsp, begin[j], len[j], r0[j], lane, rowmask[j], myflags[j]);
                }
                const uint32_t left = len[j] - 1024u * r0[j];         /* the first segment of an empty NAL: 0 */
                tot = gap[j] + (left < kEmitSegBytes ? left : kEmitSegBytes) + ins[j];
            }
            if (lane == 0) l.tot[kEmitSlots * wv + j] = tot;
        }
        HBS3_T_MARK(3)
        __syncthreads();
        HBS3_T_MARK(4)

        /* 2. where the group starts */
        if (wv == 0) {
            const unsigned long long x = lane < kEmitGroup ? l.tot[lane] : 0ull;
            const unsigned long long inc = wave_incl_scan_u64(x, lane);
            const unsigned long long agg = __shfl(inc, kEmitGroup - 1, 64);
            if (lane == 0 && g != 0) st_desc3(desc + g, (agg << 2) | 1ull);
            __builtin_amdgcn_s_setprio(0);
            const unsigned long long before = HBS_K3_EXP(1) ? g * 123000ull : k3_look_back(desc, g, lane, err);
            if (lane == 0) {
                st_desc3(desc + g, ((before + agg) << 2) | 2ull);
                if (g == ngroups - 1) *total = before + agg;
            }
            if (lane < kEmitGroup) l.off[lane] = before + inc - x;
        } else {
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
        HBS3_T_MARK(5)

        /* 3. the bytes, straight from the registers (no load in this code, so the stores are not made
         * to wait for one another) */
#pragma unroll
        for (int j = 0; j < kEmitSlots; ++j) {
            if (nal[j] != ~0ull) {                            /* wave-uniform */
                const uint64_t base = l.off[kEmitSlots * wv + j];
                const uint64_t first_byte = base + gap[j];            /* where this segment's first RBSP byte lands */
                const uint64_t item_end = base + l.tot[kEmitSlots * wv + j];
                const bool fits = item_end <= out_cap;
                if (fits && !HBS_K3_EXP(0)) {
                    uint32_t ins2 = 0;
                    /* emit_batch places byte `off` of the NAL at dst0 + off + (inserted since r0) */
                    emit_batch(R[j], rbsp, begin[j], len[j], r0[j], lane, rowmask[j], myflags[j], ins2, out + first_byte - 1024ull * r0[j]);
                }
                if (lane == 0) {
                    const bool first = r0[j] == 0, last = 1024ull * r0[j] + kEmitSegBytes >= len[j];
                    if (fits && first) {
                        for (uint64_t i = base; i + 1 < first_byte; ++i) out[i] = 0;      /* zero_byte / leading zeros */
                        if (first_byte != base) out[first_byte - 1] = 1;
                    }
                    if (idx_out) {
                        if (first) { idx_out[nal[j]].start = first_byte; idx_out[nal[j]].rbsp_off = begin[j]; idx_out[nal[j]].rbsp_len = len[j]; idx_out[nal[j]].status = 0; }
                        if (last) idx_out[nal[j]].end = item_end;
                    }
                    if (!fits) atomicMax(err, (uint32_t)(-HBS_E_CAPACITY));
                }
            }
        }
        HBS3_T_MARK(6)
    }
    HBS3_T_FLUSH
}

/* items never outnumber this: one per NAL plus one per started 12 KiB of payload */
uint64_t emit_items_bound(uint64_t n, uint64_t payload_bytes) { return n + payload_bytes / kEmitSegBytes + 1; }
uint64_t emit_desc_words(uint64_t items_cap) { return (items_cap + kEmitGroup - 1) / kEmitGroup + 1; }
uint64_t emit_dz_table_words() { return 32; }

int emit_grid_blocks(int device)
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return -1;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k3_fused, 256, 0) != hipSuccess || per_cu < 1) return -1;
    return prop.multiProcessorCount * per_cu;
}

/* (verdict_out: the context's own copy of the four verdict words -- the words themselves live in the shared workspace) */
__global__ void k3_summary(const unsigned long long* total, uint64_t n, uint64_t rbsp_bytes, const uint32_t* err, hbs_summary* sum,
                           const unsigned long long* total_dense = nullptr, const uint32_t* probe = nullptr, const uint32_t* tflag = nullptr,
                           const uint32_t* verdict_in = nullptr, uint32_t* verdict_out = nullptr)
{
    if (verdict_out) for (int i = 0; i < 4; ++i) verdict_out[i] = verdict_in[i];
    sum->nal_count = n; sum->nal_found = n; sum->rbsp_bytes = rbsp_bytes;
    sum->stream_bytes = (probe && three_steps_run(probe, tflag)) ? *total_dense : *total;
    sum->stop_reason = n ? -1 : 0; sum->error = -(int32_t)*err;
    sum->reserved[0] = sum->reserved[1] = sum->reserved[2] = 0;
}

__global__ void k3_summary_small(const unsigned long long* total_tiles, const unsigned long long* total_lanes, uint64_t n, uint64_t rbsp_bytes,
                                 const uint32_t* err, hbs_summary* sum, const uint32_t* tflag, uint32_t* verdict_out)
{
    if (verdict_out) for (int i = 0; i < 4; ++i) verdict_out[i] = tflag[i];
    sum->nal_count = n; sum->nal_found = n; sum->rbsp_bytes = rbsp_bytes;
    sum->stream_bytes = tile_path_on(tflag) ? *total_tiles : *total_lanes;
    sum->stop_reason = n ? -1 : 0; sum->error = -(int32_t)*err;
    sum->reserved[0] = sum->reserved[1] = sum->reserved[2] = 0;
}

/* ---- synthetic RBSP ---------------------------------------------------------- */

__global__ void k_synth_len(uint64_t seed, uint64_t n, unsigned long long* __restrict__ lens)
{
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x)
        lens[k] = synth_rbsp_len(seed, k);
}

__global__ __launch_bounds__(256)
void k_synth_fill(uint64_t seed, uint64_t n, int mode, const unsigned long long* __restrict__ lens,
                  const unsigned long long* __restrict__ offs, uint8_t* __restrict__ rbsp, uint64_t rbsp_cap,
                  hbs_nal_entry* __restrict__ idx, uint32_t* __restrict__ err)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    struct __attribute__((packed, aligned(1))) U8 { uint64_t v; };
    for (uint64_t k = wave; k < n; k += nwaves) {
        const uint32_t len = (uint32_t)lens[k];
        const uint64_t off = offs[k];
        if (off + len > rbsp_cap) { if (lane == 0) atomicMax(err, (uint32_t)(-HBS_E_CAPACITY)); continue; }
        const uint32_t nw = (len + 7) / 8;
        for (uint32_t w = lane; w < nw; w += 64) {
            const uint64_t x = synth_rbsp_word(seed, k, w, len, mode);
            uint8_t* p = rbsp + off + 8ull * w;
            if (8 * w + 8 <= len) reinterpret_cast<U8*>(p)->v = x;
            else for (uint32_t i = 0; 8 * w + i < len; ++i) p[i] = (uint8_t)(x >> (8 * i));
        }
        if (lane == 0) {
            hbs_nal_entry e;
            e.start = 0; e.end = 0; e.rbsp_off = off; e.rbsp_len = len; e.status = 0;
            idx[k] = e;
        }
    }
}

/* ---- host side --------------------------------------------------------------- */

/* ---- arena tiles: K3 with K12's shape -----------------------------------------------------------------------
 * k3_fused cuts the work by NAL: a slot holds one NAL's (segment's) rows, 83 % full on 8-12 KiB NALs, loads and stores both
 * byte-misaligned, 120 KiB per look-back.  When the NALs of the index lie back to back in the arena -- what hbs_index_extract
 * produces, and what hbs_write_headers + a scan produce -- the arena itself can be tiled the way K12 tiles a stream: 192 KiB of
 * ARENA per workgroup, 48 aligned rows of 1 KiB per wavefront in registers, and the few 16-byte chunks that need more than
 * a copy are "elements" walked exactly by wavefront 0: chunks chunk_flag() cannot clear (a 03 may have to go in) and chunks
 * in which a NAL begins (its start code goes in, the count of zeros restarts, its index entry is written).  A tile's
 * aggregate is one number -- the bytes inserted in it -- so the look-back is k3_fused's.  Every other chunk is one 16-byte
 * store at (its arena offset + bytes inserted in front of it).  Same bytes as the other paths (h264_nal.c:92-132 per NAL).
 *
 * Eligibility is decided on the device (k3t_check): NALs contiguous and in order, first byte 16-byte aligned, gaps below
 * 1 MiB, no 192 KiB window with more than 1024 NAL starts, workspace large enough; otherwise k3_fused runs as before. */
constexpr int kTRows = 48, kTWaves = 4, kTThreads = 64 * kTWaves;
constexpr uint32_t kTWaveBytes = (uint32_t)kTRows * 1024u, kTTileBytes = (uint32_t)kTWaves * kTWaveBytes;
constexpr int kTChunks = (int)(kTTileBytes / 16u);
constexpr int kTParkRows = 28;
constexpr uint32_t kTMaxStarts = 1024, kTMaxGap = 1u << 20;    /* NAL starts a tile may hold (512 until round 4: NALs of 256-448 bytes went a lane per NAL instead, at 0.12 of the peak) */
constexpr uint32_t kTFastStarts = 2u * (uint32_t)kTThreads;    /* ... of which the first two per thread are fetched inside the flag pass, the rest behind it */
constexpr uint64_t kTMinArena = 192ull << 20;      /* below, the kernel by NALs is as fast or faster (0.12 against 0.115 ms at 128 MiB, 0.173 against 0.186 at 256 MiB) */
constexpr int kTElemPass = 64;
constexpr int kTElemWaves = 2;                  /* a tile with several batches of elements: wavefront 1 parks rows too and takes every other batch (as hbs_scan4.hip) */
constexpr uint32_t kTDenseLimit = 1024;       /* a tile with more elements than this: wavefront 0 would walk them 64 at a time while every tile
                                                 behind waits (~5 us a batch) -- such a tile is walked by rows instead (k3_dense_tile) */
/* an entry of the tile's element list: chunk number | why it is one */
constexpr uint32_t kTListFlag = 0x8000u;      /* chunk_flag(): a 03 may have to go in                     */
constexpr uint32_t kTListStart = 0x4000u;     /* a NAL begins in it (or it is the arena's partial last chunk) */
constexpr uint32_t kTListChunk = 0x3FFFu;
static_assert(kTChunks <= (int)kTListChunk + 1, "a chunk number fits the list entry");


__global__ __launch_bounds__(256)
void k3t_check(const uint8_t* __restrict__ rbsp, uint64_t rbsp_bytes, const hbs_nal_entry* __restrict__ idx, uint64_t n, int gap_mode,
               uint64_t first_cap, uint64_t desc_words, int want_tiles, int pinned, uint32_t* __restrict__ tflag, uint32_t* __restrict__ err,
               uint32_t* __restrict__ probe, unsigned long long* __restrict__ first_k)
{
    if (blockIdx.x >= kCheckBlocks) {                /* launched only when a probe is wanted */
        probe_window(rbsp, rbsp_bytes, probe, blockIdx.x - kCheckBlocks);
        return;
    }
    bool bad = false, outside = false, apart = false;
    /* first_k[t] = the first NAL that begins at or behind arena tile t (k3t_first's table, filled on the way since round 4: one
     * launch and one pass over the index less).  The index is not trusted yet: tile numbers are clamped to the table, so a
     * corrupt index costs time, not memory -- and then nobody reads the table. */
    const uint64_t a0 = idx[0].rbsp_off;
    const bool fill = want_tiles && first_k != nullptr && first_cap != 0;
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < n; k += (uint64_t)kCheckBlocks * blockDim.x) {
        const uint64_t off = idx[k].rbsp_off;
        if (off > rbsp_bytes || idx[k].rbsp_len > rbsp_bytes - off) outside = true;      /* every kernel behind this one trusts the index */
        const uint64_t prev_off = k > 0 ? idx[k - 1].rbsp_off : 0ull;
        if (k > 0 && off != prev_off + idx[k - 1].rbsp_len) { bad = true; apart = true; }
        const uint64_t gap_k = gap_of(idx, k, gap_mode);
        if (gap_k >= (uint64_t)kTMaxGap) bad = true;
        if (gap_k >= 16u) apart = true;                            /* (kGMaxGap, hbs_emit_groups.h) */
        if (want_tiles && k + kTMaxStarts < n && idx[k + kTMaxStarts].rbsp_off - off < (uint64_t)kTTileBytes) bad = true;   /* (nobody asks when the tiles are not tried) */
        if (fill) {
            const uint64_t lo = k == 0 ? 0ull : (prev_off - a0) / kTTileBytes + 1ull;
            uint64_t hi = (off - a0) / kTTileBytes;
            if (hi > first_cap - 1) hi = first_cap - 1;
            for (uint64_t t = lo; t <= hi; ++t) first_k[t] = k;
        }
    }
    if (fill && blockIdx.x == 0 && threadIdx.x == 0) {             /* the tiles behind the last NAL's first byte */
        const uint64_t lo = (idx[n - 1].rbsp_off - a0) / kTTileBytes + 1ull;
        uint64_t hi = (idx[n - 1].rbsp_off + idx[n - 1].rbsp_len - a0) / kTTileBytes + 1ull;
        if (hi > first_cap - 1) hi = first_cap - 1;
        for (uint64_t t = lo; t <= hi; ++t) first_k[t] = n;
    }
    if (bad) atomicOr(&tflag[0], 1u);
    if (apart) atomicOr(&tflag[4], 1u);
    if (outside) { atomicOr(&tflag[3], 1u); atomicMax(err, (uint32_t)(-HBS_E_ARG)); }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const uint64_t arena_len = idx[n - 1].rbsp_off + idx[n - 1].rbsp_len - a0;
        const uint64_t ntiles = arena_len / kTTileBytes + 1;
        /* The tile kernel's row loads are unpredicated 16-byte loads anywhere in [a0, a0 + arena_len), clamped to the arena's
         * last 16 bytes: the arena must lie inside the caller's buffer (an index built by hand, or a corrupt one, may point
         * past it: such a call goes to the kernel by NALs, which checks every NAL against rbsp_bytes) and hold a whole chunk.
         * Subtractions only: none of the sums can wrap. */
        const uint64_t last_off = idx[n - 1].rbsp_off, last_len = idx[n - 1].rbsp_len;
        const bool inside = a0 <= last_off && last_off <= rbsp_bytes && last_len <= rbsp_bytes - last_off;
        /* An empty NAL at the very end of an arena of whole chunks begins in a chunk that holds no byte.  The tile loop keeps that
         * chunk as an element (the mask of the row that holds the arena's end, below), but the walk by rows of a dense tile takes
         * a chunk at or behind the end for nothing (dz_classify: active = x0 < arena_len) and left the start code out: found by
         * the long soak of round 6 (seed 66, iterations 47568 and 50920: streams of ~40-byte NALs, every chunk of a row with a
         * start in it, the output 3 / 4 bytes short).  One in sixteen of the streams that end in an empty NAL; they go by NALs.
         * (With a partial last chunk the NAL begins in it, and that chunk is an element either way.) */
        const bool phantom_start = last_len == 0u && (arena_len & 15u) == 0u;
        const bool ok = want_tiles && inside && arena_len >= 16u && !phantom_start &&
                        ((reinterpret_cast<uintptr_t>(rbsp) + a0) & 15u) == 0 && (pinned || arena_len >= kTMinArena) &&
                        ntiles + 1 <= first_cap && ntiles + 1 <= desc_words;
        tflag[1] = ok ? 1u : 0u;
    }
}

struct LdsT {
    unsigned long long rowbits[kTWaves * kTRows];    /* chunks in which a NAL begins (and the arena's last, partial chunk) */
    uint32_t starts[kTMaxStarts];                    /* where the tile's NALs begin, tile-relative, in order */
    uint32_t gaps[kTMaxStarts];                      /* bytes in front of each (zeros + 01)                  */
    uint32_t lens[kTMaxStarts];                      /* their rbsp_len (for the output index)                */
    uint16_t list[kTDenseLimit];                     /* the tile's elements, in order (a tile with more gives the call up) */
    uint32_t bsum[kTDenseLimit / kTElemPass];        /* tiles with several batches of elements: bytes inserted by each batch */
    uint32_t seg[kTDenseLimit + 1];                  /* bytes inserted in the tile up to and including element i; [0]: in front of the first (0) */
    u32x4 park[kTElemWaves][kTParkRows][64];         /* rows of the wavefronts that handle elements, meanwhile */
    uint32_t wave_tot[kTWaves];
    uint32_t dz_tot[kTWaves][3], dz_out[kTWaves][3]; /* dense tiles: bytes inserted in / count behind each wavefront's rows, per count it is entered with */
    uint32_t dz_in[kTWaves], dz_base[kTWaves];       /* ... the count each wavefront's rows are entered with, and the bytes inserted in front of them in the tile */
    unsigned long long before;                       /* bytes inserted in front of the tile                  */
    uint32_t ok;
    uint32_t ticket;
};

struct TileCtx {
    const uint8_t* arena;         /* first byte of the first NAL */
    uint64_t arena_len, tile_lo;  /* tile_lo: arena offset of the tile */
    uint64_t k_lo;                /* first NAL that begins in the tile */
    uint32_t m;                   /* how many do */
    uint64_t prev_begin;          /* arena offset at which the NAL in progress at the tile's first byte begins */
    uint64_t a0;                  /* idx[0].rbsp_off */
};

__device__ __forceinline__ uint32_t lower_bound_lds(const uint32_t* a, uint32_t m, uint32_t v)
{
    uint32_t lo = 0, hi = m;
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (a[mid] < v) lo = mid + 1; else hi = mid; }
    return lo;
}

/* n (0..16) bytes from the low end of lo:hi to dst, with at most four stores */
__device__ __forceinline__ void store_n(uint8_t* dst, uint64_t lo, uint64_t hi, uint32_t n)
{
    struct __attribute__((packed, aligned(1))) U8 { uint64_t v; };
    struct __attribute__((packed, aligned(1))) U4 { uint32_t v; };
    struct __attribute__((packed, aligned(1))) U2 { uint16_t v; };
    if (n & 16u) { reinterpret_cast<U8*>(dst)->v = lo; reinterpret_cast<U8*>(dst + 8)->v = hi; return; }
    if (n & 8u) { reinterpret_cast<U8*>(dst)->v = lo; dst += 8; lo = hi; }
    if (n & 4u) { reinterpret_cast<U4*>(dst)->v = (uint32_t)lo; dst += 4; lo >>= 32; }
    if (n & 2u) { reinterpret_cast<U2*>(dst)->v = (uint16_t)lo; dst += 2; lo >>= 16; }
    if (n & 1u) *dst = (uint8_t)lo;
}
/* lo:hi >>= 8 n bytes (n = 0..16) */
__device__ __forceinline__ void shift_bytes(uint64_t& lo, uint64_t& hi, uint32_t n)
{
    if (n >= 8u) { lo = n >= 16u ? 0ull : hi >> (8u * (n - 8u)); hi = 0ull; }
    else if (n) { lo = (lo >> (8u * n)) | (hi << (64u - 8u * n)); hi >>= 8u * n; }
}

/* One element: entry `ent` of the tile's list (chunk number + why).  Returns the bytes that go in (03s and the gaps of the NALs
 * that begin in it); kEmit: also writes the chunk with them at out + pos (when `store`) and the index entries of those NALs.
 * out = where the tile's byte 0 goes with nothing inserted in the tile, abs0 = its offset in the output, pos = this chunk's
 * offset from there. */
template <bool kEmit>
__device__ __forceinline__ uint32_t tile_element(const TileCtx& t, const LdsT& l, uint32_t ent, uint8_t* __restrict__ out, uint64_t abs0, uint64_t pos,
                                                 bool store, hbs_nal_entry* __restrict__ idx_out, u32x4& qk, uint32_t& jk, bool have_q)
{   /* qk, jk: the chunk's bytes and the first NAL start at or behind it, found by the counting call (the fetch then lies under the
     * look-back) and handed to the emitting one */
    const uint32_t c = ent & kTListChunk;
    const uint32_t p = 16u * c;
    const uint64_t x0 = t.tile_lo + p;
    const uint32_t nb = t.arena_len - x0 < 16u ? (uint32_t)(t.arena_len - x0) : 16u;
    if (!have_q) { qk = load_chunk_guarded(t.arena, x0, t.arena_len); jk = lower_bound_lds(l.starts, t.m, p); }
    const uint32_t j0 = jk;
    if (ent & kTListStart) {
        /* NALs begin here and chunk_flag() cleared the chunk: no 03 can go in (a start only resets the count), so what goes in
         * is the gaps, and the chunk's bytes go out in pieces between them */
        uint32_t ins = 0, j = j0;
        uint64_t lo = 0, hi = 0;
        uint8_t* dst = out + pos;
        uint32_t done = 0;                                           /* bytes of the chunk written so far */
        if (kEmit && store) { lo = ((uint64_t)qk.y << 32) | qk.x; hi = ((uint64_t)qk.w << 32) | qk.z; }
#pragma unroll 1
        while (j < t.m) {
            const uint32_t sj = l.starts[j];
            if (sj >= p + 16u) break;
            const uint32_t gap = l.gaps[j];
            if (kEmit) {
                const uint32_t i = sj - p;
                if (store) {
                    store_n(dst, lo, hi, i - done);
                    shift_bytes(lo, hi, i - done);
                    dst += i - done;
                    if (gap <= 8u) store_n(dst, gap ? 1ull << (8u * (gap - 1u)) : 0ull, 0ull, gap);
                    else { for (uint32_t g = 0; g + 1 < gap; ++g) dst[g] = 0; dst[gap - 1] = 1; }
                    dst += gap;
                }
                done = i;
                if (idx_out) {
                    const uint64_t k = t.k_lo + j;
                    const uint64_t at = abs0 + pos + i + ins;        /* where its gap begins = where the NAL in front ends */
                    if (k > 0) idx_out[k - 1].end = at;
                    idx_out[k].start = at + gap; idx_out[k].rbsp_off = t.a0 + x0 + i; idx_out[k].rbsp_len = l.lens[j]; idx_out[k].status = 0;
                }
            }
            ins += gap;
            ++j;
        }
        if (kEmit && store) store_n(dst, lo, hi, nb - done);
        return ins;
    }
    u32x4 q = qk;
    const uint32_t j1 = lower_bound_lds(l.starts, t.m, p + 16u);
    /* the count the chunk is entered with: zeros in front of it inside the NAL in progress */
    const uint64_t begin = j0 > 0 ? t.tile_lo + l.starts[j0 - 1] : t.prev_begin;
    uint32_t count = 0;
    if (x0 > begin) {
        const uint64_t d = x0 - begin;
        uint32_t xp = load_dword_guarded(t.arena, (int64_t)x0 - 4, t.arena_len);
        if (d < 4) xp |= 0xFFFFFFFFu >> (8u * (uint32_t)d);           /* bytes of the NAL in front: not zeros of this one */
        count = lead_count4(xp);
        if (count == kLeadUnknown) count = d == 4 ? 2u : lead_count_dev(t.arena, begin, x0);
    }
    if (j0 == j1) {                                                  /* no NAL begins here: a flagged chunk as in the other paths */
        const uint32_t mask = insert_mask16(q.x, q.y, q.z, q.w, nb, count);
        if (kEmit && store) {
            ExactChunk e;
            e.q = q; e.nb = nb; e.mask = mask;
            store_exact(out + pos, e);
        }
        return (uint32_t)__builtin_popcount(mask);
    }
    /* both (rare): byte by byte */
    uint32_t ins = 0, j = j0;
    uint8_t* dst = out + pos;
#pragma unroll 1
    for (uint32_t i = 0; i < 16u; ++i) {
        while (j < j1 && l.starts[j] == p + i) {                     /* NAL k_lo + j begins in front of byte i (empty NALs: several) */
            const uint32_t gap = l.gaps[j];
            if (kEmit) {
                const uint64_t k = t.k_lo + j;
                const uint64_t at = abs0 + pos + i + ins;
                if (store) {
                    for (uint32_t g = 0; g + 1 < gap; ++g) dst[i + ins + g] = 0;
                    if (gap) dst[i + ins + gap - 1] = 1;
                }
                if (idx_out) {
                    if (k > 0) idx_out[k - 1].end = at;
                    idx_out[k].start = at + gap; idx_out[k].rbsp_off = t.a0 + x0 + i; idx_out[k].rbsp_len = l.lens[j]; idx_out[k].status = 0;
                }
            }
            ins += gap;
            count = 0;
            ++j;
        }
        if (i < nb) {
            const uint32_t v = q.x & 0xFFu;
            if (count == 2u && v <= 3u) { if (kEmit && store) dst[i + ins] = 3; ++ins; count = 0; }
            if (kEmit && store) dst[i + ins] = (uint8_t)v;
            count = (v == 0u) ? count + 1u : 0u;
            q.x = (q.x >> 8) | (q.y << 24); q.y = (q.y >> 8) | (q.z << 24); q.z = (q.z >> 8) | (q.w << 24); q.w >>= 8;
        }
    }
    return ins;
}

#ifdef HBS_DZ_TIMING
/* dev aid (never in the shipped library): cycles (s_memtime, 100 MHz) wavefront 0 spends in each part of k3_dense_tile, summed over tiles */
__device__ unsigned long long g_dz_cycles[8];
extern "C" int hbs_debug_dz_cycles(unsigned long long* host_out, int reset)
{
    static const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int rc = (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dz_cycles), sizeof(z));
    if (reset) rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dz_cycles), z, sizeof(z));
    return rc;
}
#define DZ_T_DECL unsigned long long dz_prev = __builtin_amdgcn_s_memtime();
#define DZ_T_MARK(i) { __builtin_amdgcn_s_waitcnt(0); const unsigned long long dz_now = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) atomicAdd(&g_dz_cycles[i], dz_now - dz_prev); dz_prev = dz_now; }
#else
#define DZ_T_DECL
#define DZ_T_MARK(i)
#endif

/* ---- dense tiles (round 4) ------------------------------------------------------------------------------
 * A tile with more than kTDenseLimit elements -- a stretch of zeros in the arena: what cabac_zero_words or 00 00 03 padding
 * leave behind -- used to hand the WHOLE call to the kernel by NALs (a 16 GiB arena with 1 % of such bytes: 2.3 x the
 * uniform time).  Now such a tile is walked by rows, every chunk looked at, by all four wavefronts:
 *   - what a chunk does depends on the count it is entered with (h264_nal.c:110-116: 0, 1 or 2 zeros seen) only through its
 *     leading zeros; an all-zero chunk maps 0 -> 2, 1 -> 1, 2 -> 2 and takes 7 / 8 / 8 bytes in; any other chunk leaves a
 *     count that does not depend on the one it was entered with.  So the count a chunk is entered with comes from the
 *     nearest chunk in front that is not all zeros (a ballot and a shuffle per row), not from walking the run back;
 *   - a wavefront does not know the count ITS rows are entered with until the wavefronts in front are done, so the first half
 *     carries all three possibilities (they coincide behind the first chunk that is not all zeros) and lane 0 picks;
 *   - the count the TILE is entered with is found by wavefront 0 walking back a KiB at a time (parity of the run of zeros);
 *   - chunks in which a NAL begins (and the arena's partial last chunk) are walked byte by byte.
 * The tile then publishes its bytes like any other, resolves its look-back, and writes. */
/* (dz_map, dz_ins, dz_count_of, dz_lead_bits, dz_fast4, dz_one_start4: hbs_emit.h, shared with the CPU tests) */
__device__ __forceinline__ uint32_t top_zero_bytes(const u32x4& q) { return top_zero_bytes4(q.x, q.y, q.z, q.w); }
__device__ __forceinline__ uint32_t low_zero_bytes(const u32x4& q) { return low_zero_bytes4(q.x, q.y, q.z, q.w); }
__device__ __forceinline__ uint32_t byte_of(const u32x4& q, uint32_t i) { return byte_of4(q.x, q.y, q.z, q.w, i); }

/* the run of zeros in front of arena byte `pos`, inside the NAL that begins at `begin` (<= pos), followed back a KiB at a time by
 * the whole wavefront (lane l looks at the 16 bytes that end 16 l in front of the current end) -- for `max_kib` KiB at most:
 * open = the bound was reached with nothing but zeros seen (the run goes on) */
__device__ __forceinline__ uint64_t wave_lead_run(const uint8_t* __restrict__ arena, uint64_t begin, uint64_t pos, int lane, uint32_t max_kib, bool& open)
{
    struct __attribute__((packed, aligned(1))) U16 { u32x4 v; };
    uint64_t end = pos, run = 0;
    open = false;
    for (uint32_t it = 0;; ++it) {
        if (end <= begin) break;
        if (it >= max_kib) { open = true; break; }
        uint32_t tz;                                               /* zeros at the top of my window, counted down to `begin` at most */
        const int64_t lo = (int64_t)end - 16 * (int64_t)(lane + 1);
        if (lo >= (int64_t)begin) {
            tz = top_zero_bytes(reinterpret_cast<const U16*>(arena + lo)->v);
        } else {
            tz = 0;
            const uint64_t top = end - 16u * (uint64_t)lane;       /* one past my window; may already be at or below begin */
#pragma unroll 1
            for (uint64_t b = top; b > begin && b + 16u > top && arena[b - 1] == 0; --b) ++tz;
            if (top <= begin) tz = 0;
            /* a window cut by `begin` ends the run even when all of its bytes are zeros: mark it "not full" below */
        }
        const bool full = tz == 16u && lo >= (int64_t)begin;
        const uint64_t fm = __ballot(full);
        const int first = fm == ~0ull ? 64 : (int)__builtin_ctzll(~fm);
        if (first == 64) { run += 1024u; end -= 1024u; continue; }
        run += 16u * (uint64_t)first + (uint64_t)(uint32_t)__shfl((int)tz, first, 64);
        break;
    }
    return run;
}
/* ... and the count byte `pos` is entered with (the whole run) */
__device__ __forceinline__ uint32_t wave_lead_count(const uint8_t* __restrict__ arena, uint64_t begin, uint64_t pos, int lane)
{
    bool open;
    return dz_count_of(wave_lead_run(arena, begin, pos, lane, 0xFFFFFFFFu, open));
}

/* What a dense tile leaves in its look-back word's tag for the tiles behind it (round 4).  The count a dense tile is entered
 * with used to be found by walking the run of zeros in front of it back to its first byte: every tile of a long stretch of
 * zeros walked the whole stretch (a 16 GiB arena with 1 % of it in 640 KiB stretches of zeros: 35 ms against 6.1).  Now:
 *   tag 1 + e   the count behind the tile is e, whatever it was entered with (something in it is not a zero, or a NAL begins)
 *               -- written with the tile's first word, before anything is waited for;
 *   tag 4       nothing but zeros and no NAL begins: the count behind it is dz_map(the count in front) -- written at once
 *               too, and replaced by 1 + e with the status-1 word once the tile knows its own;
 *   tag 0       with a status: an ordinary tile (at most kTDenseLimit flagged chunks: its zeros end within ~16 KiB).
 * A dense tile first looks one KiB back; if that is all zeros it reads the words of the tiles in front: the nearest one that is
 * not tag 4 gives the count (directly, or by a walk from its end that is short by construction), and any tag-4 tiles between
 * map it once (dz_map is idempotent).  No tile waits for more than its predecessors' FIRST words. */
constexpr unsigned long long kDzTagPure = 4ull;
constexpr uint32_t kDzTableWords = 32;            /* (= emit_dz_table_words()) per tile: 4 wavefronts x (bytes for the three counts, count behind for the three), [24] = the call's number
                                                     when the tile was counted ahead (k3t_sample, below) */
__device__ __forceinline__ uint32_t dz_entry_count(const TileCtx& t, const unsigned long long* __restrict__ desc, uint64_t tile, int lane, uint32_t* err)
{
    if (tile == 0 || t.tile_lo <= t.prev_begin) return 0u;
    bool open;
    const uint64_t run1 = wave_lead_run(t.arena, t.prev_begin, t.tile_lo, lane, 1u, open);
    if (!open) return dz_count_of(run1);
    uint64_t win = 0;                                              /* tiles tile-1-win .. tile-64-win are looked at */
    uint32_t spins = 0;
    for (;;) {
        const uint64_t dist = win + (uint64_t)lane;                /* tile - 1 - dist */
        const bool valid = dist < tile;
        const unsigned long long v = valid ? ld_desc3(desc + (tile - 1 - dist)) : 0ull;
        const unsigned long long tag = v >> kDescTagShift;
        const bool pure = valid && tag == kDzTagPure;
        const uint64_t notpure = __ballot(!pure);
        if (notpure == 0ull) { win += 64; continue; }              /* 64 tiles of nothing but zeros: further back */
        const int f = (int)__builtin_ctzll(notpure);
        const unsigned long long vf = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), f) << 32) |
                                      (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, f);
        const uint64_t df = win + (uint64_t)f;
        uint32_t c;
        if (df >= tile) c = 0u;                                    /* (in front of the arena: cannot be, tile 0 holds a NAL start) */
        else if (vf == 0ull) {                                     /* that tile has not said anything yet */
            if (++spins > kEmitSpinLimit) { if (lane == 0) atomicMax(err, (uint32_t)(-HBS_E_TIMEOUT)); return 0u; }
            __builtin_amdgcn_s_sleep(1);
            continue;
        } else if ((vf >> kDescTagShift) != 0ull) c = (uint32_t)(vf >> kDescTagShift) - 1u;
        else c = wave_lead_count(t.arena, t.prev_begin, (tile - df) * (uint64_t)kTTileBytes, lane);   /* an ordinary tile: from its end */
        return df > 0 ? dz_map(c) : c;
    }
}

/* a chunk in which NALs begin (or the arena's partial last chunk): byte by byte, entered with `count`.  Returns the bytes that go
 * in (03s and gaps); count_out = the count behind it.  kEmit: writes the chunk at out + pos and the index entries. */
template <bool kEmit>
__device__ __forceinline__ uint32_t dz_start_chunk(const TileCtx& t, const LdsT& l, uint32_t c, u32x4 q, uint32_t nb, uint32_t count, uint32_t& count_out,
                                                   uint8_t* __restrict__ out, uint64_t abs0, uint64_t pos, bool store, hbs_nal_entry* __restrict__ idx_out)
{
    const uint32_t p = 16u * c;
    const uint64_t x0 = t.tile_lo + p;
    uint32_t j = lower_bound_lds(l.starts, t.m, p);
    uint32_t ins = 0;
    uint8_t* dst = out + pos;
#pragma unroll 1
    for (uint32_t i = 0; i <= 16u; ++i) {
        while (j < t.m && l.starts[j] == p + i && i < 16u) {       /* NAL k_lo + j begins in front of byte i (empty NALs: several) */
            const uint32_t gap = l.gaps[j];
            if (kEmit) {
                const uint64_t k = t.k_lo + j;
                const uint64_t at = abs0 + pos + i + ins;
                if (store) {
                    for (uint32_t g = 0; g + 1 < gap; ++g) dst[i + ins + g] = 0;
                    if (gap) dst[i + ins + gap - 1] = 1;
                }
                if (idx_out) {
                    if (k > 0) idx_out[k - 1].end = at;
                    idx_out[k].start = at + gap; idx_out[k].rbsp_off = t.a0 + x0 + i; idx_out[k].rbsp_len = l.lens[j]; idx_out[k].status = 0;
                }
            }
            ins += gap;
            count = 0;
            ++j;
        }
        if (i < nb) {
            const uint32_t v = q.x & 0xFFu;
            if (count == 2u && v <= 3u) { if (kEmit && store) dst[i + ins] = 3; ++ins; count = 0; }
            if (kEmit && store) dst[i + ins] = (uint8_t)v;
            count = (v == 0u) ? count + 1u : 0u;
            q.x = (q.x >> 8) | (q.y << 24); q.y = (q.y >> 8) | (q.z << 24); q.z = (q.z >> 8) | (q.w << 24); q.w >>= 8;
        }
    }
    count_out = count;
    return ins;
}

/* one row of a dense tile: what lane's chunk is and does.  q = its 16 bytes, xp = the dword in front of it */
struct DzChunk { uint32_t nb; bool active, start, reset; uint32_t ins[3], out; u32x4 q; uint32_t lz, mask0; bool flagged; };
__device__ __forceinline__ void dz_classify(DzChunk& d, const TileCtx& t, const LdsT& l, uint32_t c, bool has_start, uint32_t xp)
{
    const uint64_t x0 = t.tile_lo + 16ull * c;
    d.active = x0 < t.arena_len;
    d.nb = !d.active ? 0u : (t.arena_len - x0 < 16u ? (uint32_t)(t.arena_len - x0) : 16u);
    d.start = d.active && (has_start || d.nb < 16u);
    d.lz = 0; d.mask0 = 0; d.flagged = false;
    d.ins[0] = d.ins[1] = d.ins[2] = 0; d.out = 0;
    if (!d.active) { d.reset = false; return; }
    if (d.start) {
        d.reset = true;
        for (uint32_t h = 0; h < 3u; ++h) {
            uint32_t co;
            d.ins[h] = dz_start_chunk<false>(t, l, c, d.q, d.nb, h, co, nullptr, 0, 0, false, nullptr);
            d.out = co;                                            /* (the same for every h when a NAL begins in the chunk; the arena's last chunk has nothing behind it) */
        }
        return;
    }
    d.lz = low_zero_bytes(d.q);
    if (d.lz == 16u) { d.reset = false; d.ins[0] = 7u; d.ins[1] = 8u; d.ins[2] = 8u; return; }
    d.reset = true;
    d.out = dz_count_of(top_zero_bytes(d.q));
    d.flagged = chunk_flag(xp, d.q.x, d.q.y, d.q.z, d.q.w, 0xFFFFFFFFu);
    if (!d.flagged) return;                                        /* no two zeros next to each other in or just in front of it: nothing goes in */
    d.mask0 = insert_mask16(d.q.x, d.q.y, d.q.z, d.q.w, 16u, 0u);
    const uint32_t v = byte_of(d.q, d.lz);
    const uint32_t keep = d.mask0 & ~((2u << d.lz) - 1u);          /* behind the first byte that is not zero: the same whatever the count was */
#pragma unroll
    for (uint32_t h = 0; h < 3u; ++h) {
        const uint32_t lead = dz_lead_bits(h) & ((1u << d.lz) - 1u);
        const uint32_t first = (dz_count_of((uint64_t)h + d.lz) == 2u && v <= 3u) ? (1u << d.lz) : 0u;
        d.ins[h] = (uint32_t)__builtin_popcount(keep | lead | first);
    }
}
__device__ __forceinline__ uint32_t dz_mask_for(const DzChunk& d, uint32_t c_in)
{
    if (d.lz == 16u) return dz_lead_bits(c_in);
    if (!d.flagged) return 0u;
    const uint32_t v = byte_of(d.q, d.lz);
    const uint32_t keep = d.mask0 & ~((2u << d.lz) - 1u);
    const uint32_t lead = dz_lead_bits(c_in) & ((1u << d.lz) - 1u);
    const uint32_t first = (dz_count_of((uint64_t)c_in + d.lz) == 2u && v <= 3u) ? (1u << d.lz) : 0u;
    return keep | lead | first;
}
__device__ __forceinline__ uint32_t dz_pick(const uint32_t* a, uint32_t c) { return c == 0u ? a[0] : (c == 1u ? a[1] : a[2]); }

/* the count lane's chunk is entered with, given the row's (row_in), and the count behind the row */
__device__ __forceinline__ uint32_t dz_count_in(bool reset, uint32_t out, uint32_t row_in, int lane, uint32_t& row_out)
{
    const uint64_t rm = __ballot(reset);
    const uint64_t below = rm & ((1ull << lane) - 1ull);
    uint32_t c_in;
    const int src = below ? 63 - (int)__builtin_clzll(below) : 0;
    const uint32_t from = (uint32_t)__shfl((int)out, src, 64);
    if (below) c_in = (lane - src > 1) ? dz_map(from) : from;
    else c_in = lane == 0 ? row_in : dz_map(row_in);
    const uint32_t c_out = reset ? out : dz_map(c_in);
    row_out = (uint32_t)__shfl((int)c_out, 63, 64);
    return c_in;
}

/* ---- the first half again, without a branch in it (round 4) -----------------------------------------------------
 * Row by row with the rows' special cases as branches, the first half of a dense tile ran as ONE chain of dependent
 * instructions per wavefront -- 2.4 us a row, 117 us a tile by the shader clock, and every tile behind waits for the bytes it
 * adds up (deeper prefetch changed nothing: it is not the memory).  Here a group of eight rows is classified first, every
 * row by the same straight-line code (what each chunk adds for each of the three counts it may be entered with, the count
 * behind it, whether it is all zeros), so that the eight rows' instructions interleave; the combination in order behind
 * it is a ballot, a shuffle and a scan per row.  Chunks in which a NAL begins are redone by dz_start3 (not inlined: a few
 * rows of a tile).  Whole tiles only; the arena's last tile, cut by its end, takes the loop below as before. */
__device__ __forceinline__ DzFast dz_fast(const u32x4& q) { return dz_fast4(q.x, q.y, q.z, q.w); }
/* a chunk in which NALs begin, for the three counts: byte by byte (several NALs begin in it: NALs shorter than a chunk) */
__device__ __attribute__((noinline))
DzFast dz_start3_serial(const uint32_t* starts, const uint32_t* gaps, uint32_t m, uint32_t p, uint32_t j0, u32x4 q0)
{
    DzFast r;
    r.reset = 1u; r.out = 0u;
    uint32_t ins[3];
#pragma unroll 1
    for (uint32_t h = 0; h < 3u; ++h) {
        u32x4 q = q0;
        uint32_t j = j0, n_in = 0, count = h;
#pragma unroll 1
        for (uint32_t i = 0; i < 16u; ++i) {
            while (j < m && starts[j] == p + i) { n_in += gaps[j]; count = 0; ++j; }
            const uint32_t v = q.x & 0xFFu;
            if (count == 2u && v <= 3u) { ++n_in; count = 0; }
            count = (v == 0u) ? count + 1u : 0u;
            q.x = (q.x >> 8) | (q.y << 24); q.y = (q.y >> 8) | (q.z << 24); q.z = (q.z >> 8) | (q.w << 24); q.w >>= 8;
        }
        ins[h] = n_in;
        r.out = count;
    }
    r.i0 = ins[0]; r.i1 = ins[1]; r.i2 = ins[2];
    return r;
}
/* ... the usual case, one NAL begins in the chunk (dz_one_start4: two half chunks in closed form; byte by byte through LDS this
 * was 10 us a row, half of the first half) */
__device__ __forceinline__ DzFast dz_start3(const LdsT& l, uint32_t m, uint32_t c, const u32x4& q)
{
    const uint32_t p = 16u * c;
    const uint32_t j0 = lower_bound_lds(l.starts, m, p);
    const uint32_t s0 = l.starts[j0 < m ? j0 : 0u] - p, gap0 = l.gaps[j0 < m ? j0 : 0u];
    const bool several = j0 + 1u < m && l.starts[j0 + 1u] < p + 16u;
    if (several || j0 >= m || s0 >= 16u) return dz_start3_serial(l.starts, l.gaps, m, p, j0, q);
    return dz_one_start4(q.x, q.y, q.z, q.w, s0, gap0);
}
constexpr uint32_t kDzRowBuf = 2048;            /* bytes of a row with what goes into it that are put together in LDS (1024 + 512 of 03s + start codes) */
constexpr int kDzGroup = 8;
static_assert(kTRows % kDzGroup == 0, "whole groups of rows");

/* One dense tile, by the whole workgroup (every thread calls it); `l` still holds the tile's NAL starts, gaps and lengths and
 * the chunks they lie in.  Not inlined: its registers must not add to the 192 the rows of the ordinary path occupy. */
__device__ __attribute__((noinline))
void k3_dense_tile(LdsT& l, TileCtx t, uint64_t tile, bool last_tile, unsigned long long* __restrict__ desc,
                   uint8_t* __restrict__ out, uint64_t out_cap, hbs_nal_entry* __restrict__ idx_out, uint64_t n,
                   unsigned long long* __restrict__ total, uint32_t* __restrict__ err,
                   int ahead, uint32_t* __restrict__ dz_table, uint32_t call_no)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint64_t wseg = t.tile_lo + (uint64_t)(wv * (int)kTWaveBytes);
    const uint32_t chunk0 = (uint32_t)(64 * kTRows * wv);
    const uint64_t tile_bytes = t.arena_len - t.tile_lo < (uint64_t)kTTileBytes ? t.arena_len - t.tile_lo : (uint64_t)kTTileBytes;
    auto load_row = [&](int r) -> u32x4 {
        const uint64_t x0 = wseg + 1024ull * (uint32_t)r + 16ull * (uint32_t)lane;
        if (x0 + 16u <= t.arena_len) return *reinterpret_cast<const u32x4*>(t.arena + x0);
        if (x0 < t.arena_len) return load_chunk_guarded(t.arena, x0, t.arena_len);
        return u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    };
    const uint32_t seg_before = load_dword_guarded(t.arena, (int64_t)wseg - 4, t.arena_len);
    DZ_T_DECL

    /* ---- first half: bytes that go into my rows, for each count they may be entered with ---- */
    uint32_t tot[3] = {0u, 0u, 0u}, st[3] = {0u, 1u, 2u};
    uint32_t* const entry = dz_table ? dz_table + tile * (uint64_t)kDzTableWords : nullptr;
    /* counted ahead in this call (k3t_sample): the table has what the loops below would find */
    const bool counted = !ahead && entry && __hip_atomic_load(entry + 24, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == call_no;
    if (counted) {
#pragma unroll
        for (int h = 0; h < 3; ++h) { tot[h] = entry[6 * wv + h]; st[h] = entry[6 * wv + 3 + h]; }
    } else if (tile_bytes == (uint64_t)kTTileBytes) {
        /* a whole tile: groups of kDzGroup rows (see dz_fast above), the next group's loads in flight */
        u32x4 cur[kDzGroup], nxt[kDzGroup];
#pragma unroll
        for (int i = 0; i < kDzGroup; ++i) cur[i] = *reinterpret_cast<const u32x4*>(t.arena + wseg + 1024ull * (uint32_t)i + 16ull * (uint32_t)lane);
#pragma unroll 1
        for (int g = 0; g < kTRows / kDzGroup; ++g) {
            if (g + 1 < kTRows / kDzGroup) {
#pragma unroll
                for (int i = 0; i < kDzGroup; ++i)
                    nxt[i] = *reinterpret_cast<const u32x4*>(t.arena + wseg + 1024ull * (uint32_t)(kDzGroup * (g + 1) + i) + 16ull * (uint32_t)lane);
            }
            DzFast a[kDzGroup];
#pragma unroll
            for (int i = 0; i < kDzGroup; ++i) a[i] = dz_fast(cur[i]);
#pragma unroll
            for (int i = 0; i < kDzGroup; ++i) {
                const unsigned long long rb = l.rowbits[wv * kTRows + kDzGroup * g + i];
                if (rb != 0ull) {                                   /* NALs begin in this row: those chunks byte by byte */
                    if ((rb >> lane) & 1ull) a[i] = dz_start3(l, t.m, chunk0 + 64u * (uint32_t)(kDzGroup * g + i) + (uint32_t)lane, cur[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < kDzGroup; ++i) {
                const uint64_t rm = __ballot(a[i].reset != 0u);
                const int f = rm ? (int)__builtin_ctzll(rm) : 64;   /* the row's first chunk that resets the count (64: none) */
                const uint64_t below = rm & ((1ull << lane) - 1ull);
                const int src = below ? 63 - (int)__builtin_clzll(below) : 0;
                const uint32_t from = (uint32_t)__shfl((int)a[i].out, src, 64);
                const uint32_t c_any = (lane - src > 1) ? dz_map(from) : from;             /* right for the lanes behind f, whatever the row is entered with */
                const uint32_t c_out = a[i].reset ? a[i].out : dz_map(c_any);
                const uint32_t ro_any = (uint32_t)__builtin_amdgcn_readlane((int)c_out, 63);
                const uint32_t mine = c_any == 0u ? a[i].i0 : (c_any == 1u ? a[i].i1 : a[i].i2);
                const uint32_t common = wave_sum32(lane > f ? mine : 0u);
                const int fl = f < 64 ? f : 63;
                const uint32_t fi0 = (uint32_t)__builtin_amdgcn_readlane((int)a[i].i0, fl);
                const uint32_t fi1 = (uint32_t)__builtin_amdgcn_readlane((int)a[i].i1, fl);
                const uint32_t fi2 = (uint32_t)__builtin_amdgcn_readlane((int)a[i].i2, fl);
                const uint32_t nlead = (uint32_t)f;                 /* all-zero chunks in front of it */
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    const uint32_t c_f = nlead == 0u ? st[h] : dz_map(st[h]);
                    const uint32_t lead = nlead ? dz_ins(st[h]) + (nlead - 1u) * 8u : 0u;
                    tot[h] += common + lead + (f < 64 ? (c_f == 0u ? fi0 : c_f == 1u ? fi1 : fi2) : 0u);
                    st[h] = f < 64 ? ro_any : dz_map(st[h]);
                }
            }
#pragma unroll
            for (int i = 0; i < kDzGroup; ++i) cur[i] = nxt[i];
        }
    } else {
        /* two rows ahead: what the tiles behind this one wait for is this half */
        u32x4 qa = load_row(0), qb = load_row(1);
        uint32_t e_prev = seg_before;
#pragma unroll 1
        for (int r = 0; r < kTRows; ++r) {
            DzChunk d;
            d.q = qa;
            qa = qb;
            if (r + 2 < kTRows) qb = load_row(r + 2);
            const uint32_t xp = from_prev_lane(d.q.w, e_prev);
            e_prev = (uint32_t)__builtin_amdgcn_readlane((int)d.q.w, 63);
            const unsigned long long rb = l.rowbits[wv * kTRows + r];
            const uint64_t row_lo = wseg + 1024ull * (uint32_t)r;
            if (row_lo >= t.arena_len) break;                       /* behind the arena's end */
            const bool whole_row = row_lo + 1024u <= t.arena_len;
            /* a row of nothing but zeros (padding looks like that): closed form */
            if (whole_row && rb == 0ull && __ballot((d.q.x | d.q.y | d.q.z | d.q.w) != 0u) == 0ull) {
#pragma unroll
                for (int h = 0; h < 3; ++h) { tot[h] += dz_ins(st[h]) + 63u * 8u; st[h] = dz_map(st[h]); }
                continue;
            }
            dz_classify(d, t, l, chunk0 + 64u * (uint32_t)r + (uint32_t)lane, ((rb >> lane) & 1ull) != 0, xp);
            /* Behind the row's first chunk that is not all zeros the count a chunk is entered with does not depend on the count
             * the ROW is entered with: one pass for those, a closed form for the zeros in front, and the first such chunk by itself. */
            const uint64_t rm = __ballot(d.reset);
            const int f = rm ? (int)__builtin_ctzll(rm) : 64;       /* the first chunk that resets the count (64: none) */
            uint32_t ro_any;
            const uint32_t c_any = dz_count_in(d.reset, d.out, 0u, lane, ro_any);     /* right for lanes behind f, whatever the row's count */
            const uint32_t common = wave_sum32((d.active && lane > f) ? dz_pick(d.ins, c_any) : 0u);
            const uint32_t fi0 = f < 64 ? (uint32_t)__shfl((int)d.ins[0], f, 64) : 0u;
            const uint32_t fi1 = f < 64 ? (uint32_t)__shfl((int)d.ins[1], f, 64) : 0u;
            const uint32_t fi2 = f < 64 ? (uint32_t)__shfl((int)d.ins[2], f, 64) : 0u;
            const uint32_t nact = (uint32_t)__builtin_popcountll(__ballot(d.active));
            const uint32_t nlead = f < 64 ? (uint32_t)f : nact;     /* all-zero chunks in front of it (a cut row ends in a chunk that resets) */
#pragma unroll
            for (int h = 0; h < 3; ++h) {
                const uint32_t c_f = nlead == 0u ? st[h] : dz_map(st[h]);
                const uint32_t lead = nlead ? dz_ins(st[h]) + (nlead - 1u) * 8u : 0u;
                tot[h] += common + lead + (f < 64 ? (c_f == 0u ? fi0 : c_f == 1u ? fi1 : fi2) : 0u);
                st[h] = f < 64 ? ro_any : dz_map(st[h]);
            }
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int h = 0; h < 3; ++h) { l.dz_tot[wv][h] = tot[h]; l.dz_out[wv][h] = st[h]; }
    }
    DZ_T_MARK(0)
    if (ahead) {                                                   /* count ahead: the results to the table, nothing else */
        if (entry && lane == 0) {
#pragma unroll
            for (int h = 0; h < 3; ++h) { entry[6 * wv + h] = tot[h]; entry[6 * wv + 3 + h] = st[h]; }
        }
        __syncthreads();
        if (entry && tid == 0) { __threadfence(); __hip_atomic_store(entry + 24, call_no, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        return;
    }
    __syncthreads();
    DZ_T_MARK(1)
    if (wv == 0) {
        /* what the tile does for each count it may be entered with; the same count behind it for all three: the tiles behind need
         * not wait for this one's own (dz_entry_count) */
        uint32_t f_out[3];
#pragma unroll
        for (int h = 0; h < 3; ++h) {
            uint32_t c = (uint32_t)h;
            for (int w = 0; w < kTWaves; ++w) c = dz_pick(l.dz_out[w], c);
            f_out[h] = c;
        }
        const bool constant = f_out[0] == f_out[1] && f_out[1] == f_out[2];
        if (lane == 0) st_desc3(desc + tile, (constant ? 1ull + f_out[0] : kDzTagPure) << kDescTagShift);
        /* the count the tile is entered with: the zeros in front of it inside the NAL in progress */
        const uint32_t c0 = dz_entry_count(t, desc, tile, lane, err);
        DZ_T_MARK(2)
        unsigned long long tile_ins = 0;
        uint32_t c = c0;
        for (int w = 0; w < kTWaves; ++w) {
            if (lane == 0) { l.dz_in[w] = c; l.dz_base[w] = (uint32_t)tile_ins; }
            tile_ins += dz_pick(l.dz_tot[w], c);
            c = dz_pick(l.dz_out[w], c);
        }
        const unsigned long long tag = (1ull + c) << kDescTagShift;  /* c: the count behind the tile */
        if (lane == 0 && tile != 0) st_desc3(desc + tile, tag | (tile_ins << 2) | 1ull);
        __builtin_amdgcn_s_setprio(0);
        const unsigned long long bf = k3_look_back(desc, tile, lane, err);
        DZ_T_MARK(3)
        if (lane == 0) {
            st_desc3(desc + tile, tag | ((bf + tile_ins) << 2) | 2ull);
            const uint64_t end_pos = t.tile_lo + tile_bytes + bf + tile_ins;
            l.before = bf;
            l.ok = end_pos <= out_cap ? 1u : 0u;
            if (end_pos > out_cap) atomicMax(err, (uint32_t)(-HBS_E_CAPACITY));
            if (last_tile) {
                *total = end_pos;
                if (idx_out) idx_out[n - 1].end = end_pos;
            }
        }
    } else {
        __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();

    /* ---- second half: the bytes ---- */
    const bool can_store = l.ok != 0u;
    if (!can_store && !idx_out) return;
    uint8_t* const tout = out + t.tile_lo + l.before;
    const uint64_t abs0 = t.tile_lo + l.before;
    uint32_t base_ins = l.dz_base[wv], row_in = l.dz_in[wv];
    typedef __attribute__((address_space(3))) uint8_t lds_u8;
    uint8_t* const rowbuf = reinterpret_cast<uint8_t*>(&l.park[0][0][0]) + (size_t)(kDzRowBuf + 64u) * (uint32_t)wv;   /* (the parked rows' place: nothing is parked in a dense tile) */
    u32x4 qa = load_row(0), qb = load_row(1);
    uint32_t e_prev = seg_before;
#pragma unroll 1
    for (int r = 0; r < kTRows; ++r) {
        DzChunk d;
        d.q = qa;
        qa = qb;
        if (r + 2 < kTRows) qb = load_row(r + 2);
        const uint32_t xp = from_prev_lane(d.q.w, e_prev);
        e_prev = (uint32_t)__builtin_amdgcn_readlane((int)d.q.w, 63);
        const bool hs = ((l.rowbits[wv * kTRows + r] >> lane) & 1ull) != 0;
        const uint32_t c = chunk0 + 64u * (uint32_t)r + (uint32_t)lane;
        if (wseg + 1024ull * (uint32_t)r >= t.arena_len) break;
        dz_classify(d, t, l, c, hs, xp);
        uint32_t ro;
        const uint32_t c_in = dz_count_in(d.reset, d.out, row_in, lane, ro);
        row_in = ro;
        const uint32_t mine = d.active ? dz_pick(d.ins, c_in) : 0u;
        uint32_t row_tot;
        const uint32_t before_me = wave_excl_scan_u32(mine, lane, row_tot);
        const uint64_t pos = 16ull * c + base_ins + before_me;
        const uint32_t row_base = base_ins;
        base_ins += row_tot;
        /* a row into which nothing goes and in which no NAL begins: the chunks as they are */
        const bool special = __ballot(d.active && (d.start || mine != 0u)) != 0ull;
        if (!special) {
            if (d.active && can_store) arena_store16(tout + pos, d.q);
            continue;
        }
        /* Otherwise the row's bytes are put together in LDS -- every lane writes its chunk with its 03s (a NAL start: with its
         * start code) at its place in the row -- and go out as whole 16-byte pieces: written straight to the stream, a chunk with
         * 03s was up to 24 single-byte stores per lane (~300 us a tile of padding).  A row longer than the buffer (a start code of
         * hundreds of bytes) takes the old way. */
        const uint32_t row_out0 = 16u * (chunk0 + 64u * (uint32_t)r) + row_base;       /* where the row's first byte goes, from tout */
        const uint32_t row_len = wave_sum32(d.active ? d.nb + mine : 0u);
        if (row_len > kDzRowBuf || !can_store) {
            if (!d.active) continue;
            if (d.start) {
                uint32_t co;
                (void)dz_start_chunk<true>(t, l, c, d.q, d.nb, c_in, co, tout, abs0, pos, can_store, idx_out);
            } else if (can_store) {
                const uint32_t mask = dz_mask_for(d, c_in);
                if (mask == 0u) arena_store16(tout + pos, d.q);
                else (void)emit_chunk16(tout + pos, d.q.x, d.q.y, d.q.z, d.q.w, 16u, mask);
            }
            continue;
        }
        const uint32_t off = (uint32_t)(pos - (uint64_t)row_out0);
        if (d.active) {
            if (d.start) {
                uint32_t co;                                       /* (its index entries carry stream offsets: abs0 + pos, as before) */
                (void)dz_start_chunk<true>(t, l, c, d.q, d.nb, c_in, co, rowbuf - row_out0, abs0, pos, true, idx_out);
            } else {
                const uint32_t mask = dz_mask_for(d, c_in);
                lds_u8* const lb = (lds_u8*)(rowbuf) + off;
                uint32_t o = 0, w0 = d.q.x, w1 = d.q.y, w2 = d.q.z, w3 = d.q.w;
#pragma unroll
                for (uint32_t i = 0; i < 16u; ++i) {
                    if ((mask >> i) & 1u) lb[o++] = (uint8_t)3;
                    lb[o++] = (uint8_t)w0;
                    w0 = (w0 >> 8) | (w1 << 24); w1 = (w1 >> 8) | (w2 << 24); w2 = (w2 >> 8) | (w3 << 24); w3 >>= 8;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");          /* (the start chunk's bytes went through flat stores) */
        {
            const lds_u8* const lb = (const lds_u8*)(rowbuf);
            uint8_t* const dst = tout + row_out0;
#pragma unroll 1
            for (uint32_t j = (uint32_t)lane; 16u * j + 16u <= row_len; j += 64u) arena_store16(dst + 16u * j, *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(lb + 16u * j));
            const uint32_t t0 = row_len & ~15u;
            if ((uint32_t)lane < row_len - t0) dst[t0 + (uint32_t)lane] = lb[t0 + (uint32_t)lane];
        }
        __builtin_amdgcn_wave_barrier();
    }
    DZ_T_MARK(4)
#ifdef HBS_DZ_TIMING
    if (threadIdx.x == 0) atomicAdd(&g_dz_cycles[7], 1ull);
#endif
}

#ifndef HBS3T_COPY_DEPTH
#define HBS3T_COPY_DEPTH 3      /* stores of a wavefront in flight during k3_tiles' copy */
#endif
template <class F, int... Is>
__device__ __forceinline__ void t_rows_apply(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void t_for_n(F&& f) { t_rows_apply(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

/* ---- dense tiles counted ahead of the tile kernel (round 4) ---------------------------------------------------------------
 * A dense tile's first half -- the bytes it adds, for the three counts it may be entered with -- takes ~60 us, and in k3_tiles every
 * tile behind it waits for that sum: a 16 GiB arena with 1 % of it in stretches of padding ran 1.65 x the uniform time.  But the
 * sum does not depend on anything in front of the tile.  So: k3t_sample looks at 64 bytes in every 64 KiB of every tile, at 64 bytes
 * in every 16 KiB of the tiles that show something, and lists the tiles in which a sixth of those chunks end a pattern 00 00 {<= 3}; k3_tiles runs ONCE OVER THAT LIST in "count ahead" mode --
 * rows, flags, NAL starts as always, then k3_dense_tile's first half, whose per-wavefront results go to a table -- and then as
 * before, where a dense tile whose table entry carries this call's number skips its first half.  A tile the sample misses (a
 * stretch shorter than ~16 KiB) is counted in place as before; a listed tile that is not dense costs its rows once more. */
constexpr int kSampleMin = 8;       /* 8 of 48 sampled chunks (two of twelve sectors): a sixth of the tile; a zero-heavy arena (2-3 % of its chunks) shows 1-2 */
__global__ __launch_bounds__(256)
void k3t_sample(const uint8_t* __restrict__ rbsp, const hbs_nal_entry* __restrict__ idx, uint64_t n,
                uint32_t* __restrict__ cand_list, uint32_t* __restrict__ cand_count, uint64_t cand_cap, uint32_t* __restrict__ dz_table,
                const uint32_t* __restrict__ probe, const uint32_t* __restrict__ tflag, uint32_t call_no)
{
    if ((probe && emit_probe_dense_tiles(probe)) || !tile_path_on(tflag)) return;
    const uint64_t a0 = idx[0].rbsp_off;
    const uint64_t arena_len = idx[n - 1].rbsp_off + idx[n - 1].rbsp_len - a0;
    const uint64_t ntiles = arena_len / kTTileBytes + 1;
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    /* Two looks.  First 64 bytes in every 64 KiB (three sectors a tile, lanes 0-11): on coded video no chunk of them ends a pattern
     * 00 00 {<= 3} and the tile is done -- a quarter of the sectors of the full look, which took 50 us over a 16 GiB arena, all of
     * it DRAM row misses.  A tile with a hit gets the full look: 64 bytes in every 16 KiB (twelve sectors, 48 chunks); listed when
     * kSampleMin of them hit. */
    auto sample_at = [&](uint64_t tile, int sec, int sub) -> uint64_t {       /* sector `sec` of 12, chunk `sub` of 4 */
        return tile * (uint64_t)kTTileBytes + 16384ull * (uint64_t)sec + 1024ull * (uint64_t)((5 * sec + (int)tile) & 15) +
               64ull * (uint64_t)((7 * sec + (int)(tile >> 2)) & 15) + 16ull * (uint64_t)sub;
    };
    auto hit = [&](const u32x4& q) -> bool {                         /* the question the tile kernel's flag pass asks */
        return chunk_flag(0xFFFFFFFFu, q.x, q.y, q.z, q.w, 0xFFFFFFFFu) && chunk_pattern_any_dev(0xFFFFFFFFu, q.x, q.y, q.z, q.w, 0xFFFFFFFFu);
    };
    constexpr int kAtOnce = 4;                                      /* tiles a wavefront samples together: their loads in flight at the same time */
    const u32x4 none = u32x4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    for (uint64_t tile0 = wave * kAtOnce; tile0 < ntiles; tile0 += nwaves * kAtOnce) {
        u32x4 q[kAtOnce];
#pragma unroll
        for (int i = 0; i < kAtOnce; ++i) {
            const uint64_t tile = tile0 + (uint64_t)i;
            const uint64_t x = sample_at(tile, 4 * (lane >> 2) + 1, lane & 3);          /* sectors 1, 5, 9 */
            q[i] = none;
            if (lane < 12 && tile < ntiles && x + 16u <= arena_len) q[i] = *reinterpret_cast<const u32x4*>(rbsp + a0 + x);
        }
#pragma unroll
        for (int i = 0; i < kAtOnce; ++i) {
            const uint64_t tile = tile0 + (uint64_t)i;
            if (tile >= ntiles) break;
            if (lane == 0) { dz_table[tile * (uint64_t)kDzTableWords + 24u] = 0u; dz_table[tile * (uint64_t)kDzTableWords + 25u] = 0u; }   /* (the workspace is not cleared: no stale entry may carry this call's number -- a call replayed from a HIP graph has the number it was captured with) */
            if (__ballot(hit(q[i])) == 0ull) continue;
            const uint64_t x = sample_at(tile, lane >> 2, lane & 3);
            u32x4 qq = none;
            if (lane < 48 && x + 16u <= arena_len) qq = *reinterpret_cast<const u32x4*>(rbsp + a0 + x);
            const uint32_t hits = (uint32_t)__builtin_popcountll(__ballot(hit(qq)));
            if (hits >= (uint32_t)kSampleMin && lane == 0) {
                /* ... and the tile on either side, where the stretch begins and ends (round 5): a row of it makes that tile a dense
                 * one, the sample above sees a sixth of a tile at best, and the tiles behind waited for its count as before.  Word 25
                 * of a tile's entry says who listed it in this call (a stale word that happens to carry this call's number costs the
                 * listing, nothing else; listed twice it would be counted twice, with the same result) */
                for (int dt = -1; dt <= 1; ++dt) {
                    const uint64_t u = tile + (uint64_t)(int64_t)dt;
                    if ((dt < 0 && tile == 0) || u >= ntiles) continue;
                    if (atomicExch(&dz_table[u * (uint64_t)kDzTableWords + 25u], call_no) == call_no) continue;
                    const uint32_t slot = atomicAdd(cand_count, 1u);
                    if (slot < cand_cap) cand_list[slot] = (uint32_t)u;
                }
            }
        }
    }
}

template <int kAheadMode>       /* 1: the listed tiles only, up to their dense first half (see k3t_sample) -- an instance of its own, so that the main pass carries none of it */
__global__ __launch_bounds__(kTThreads, 2)
void k3_tiles(const uint8_t* __restrict__ rbsp, const hbs_nal_entry* __restrict__ idx, uint64_t n, int gap_mode,
              const unsigned long long* __restrict__ first_k, unsigned long long* __restrict__ desc, uint32_t* __restrict__ ticket,
              uint8_t* __restrict__ out, uint64_t out_cap, hbs_nal_entry* __restrict__ idx_out,
              unsigned long long* __restrict__ total, uint32_t* __restrict__ err,
              const uint32_t* __restrict__ probe, const uint32_t* __restrict__ tflag,
              const uint32_t* __restrict__ cand_list, const uint32_t* __restrict__ cand_count, uint64_t cand_cap, uint32_t* __restrict__ cand_ticket,
              uint32_t* __restrict__ dz_table, uint32_t call_no, int first_static)
{
    if ((probe && emit_probe_dense_tiles(probe)) || !tile_path_on(tflag)) return;
    constexpr int ahead = kAheadMode;
    __shared__ LdsT l;
    const int tid0 = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    TileCtx t;
    t.a0 = idx[0].rbsp_off;
    t.arena = rbsp + t.a0;
    t.arena_len = idx[n - 1].rbsp_off + idx[n - 1].rbsp_len - t.a0;
    const uint64_t ntiles = t.arena_len / kTTileBytes + 1;
    HBS3_T_DECL
#ifdef HBS_DZ_TIMING
    unsigned long long dz_tile_t0 = 0;
#endif
    uint64_t d_tile = 0;
    bool first_tile = first_static != 0;
    for (;;) {
    int pending = 0;                   /* 1: a dense tile -- walked below the tile loop, where no row is live (as in hbs_scan4.hip) */
    for (;;) {
        const int lane = launder_lane(tid0) & 63;
        const int tid = launder_lane(tid0);
        __syncthreads();                                           /* the previous tile is done with l */
        if (tid == 0) {
            if (ahead) {
                const uint32_t tk = atomicAdd(cand_ticket, 1u);
                const uint64_t have = *cand_count < cand_cap ? *cand_count : cand_cap;
                l.ticket = tk < have ? cand_list[tk] : 0xFFFFFFFFu;
            } else {
                /* every tile by ticket; first_static (the context owns the device): the first one is the workgroup's number (as hbs_scan4.hip) */
                l.ticket = first_tile ? blockIdx.x : (first_static ? gridDim.x : 0u) + atomicAdd(ticket, 1u);
            }
        }
        first_tile = false;
        if (tid < kTWaves * kTRows) l.rowbits[tid] = 0ull;
        __syncthreads();
        const uint64_t tile = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)l.ticket);
        if (tile >= ntiles) break;
        HBS3_T_MARK(0)
#ifdef HBS_DZ_TIMING
        dz_tile_t0 = __builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_s_setprio(3);
        t.tile_lo = tile * (uint64_t)kTTileBytes;
        const bool last_tile = tile == ntiles - 1;
        const uint64_t wseg = t.tile_lo + (uint64_t)(wv * (int)kTWaveBytes);

        /* ---- my rows, fetched INSIDE the flag pass, four at a time (as K12 does since round 3, hbs_scan4.hip): a group is
         * flagged as soon as its rows are there, with the next group's loads in flight, so a wavefront never has more than
         * eight row loads outstanding.  That hides the flag pass under the fetch and, more to the point, keeps the CU's memory
         * queue short, so that the look-back polls of the other workgroup on this CU do not wait behind 48 loads (or, in the
         * copy, 48 stores: throttled below).  The NAL starts of the tile are fetched on the way: the two first_k words in front
         * of the rows, the index entries behind the second group's loads -- in issue order each is back when it is needed. */
        u32x4 q[kTRows];
        uint32_t before;
        /* unpredicated (k3t_check: the arena holds at least one chunk and lies inside the caller's buffer): a chunk that would
         * reach past the arena's end reads the arena's last 16 bytes instead (its register is never used: the chunk is behind the
         * end, or it is the partial last chunk, an element that fetches its own bytes) */
        /* (the arena's last tile may hold fewer than 16 bytes -- none at all when the arena is a whole number of tiles: every load
         * of such a tile reads the arena's last chunk.  Until round 6 `room` wrapped there and the tile read up to 192 KiB BEHIND the
         * arena: unused bytes, but a memory fault where the arena ends with its allocation -- found by tests/tools/soak_emit_small.py
         * once it pinned this path, and then on the automatic path at exactly 1 024 tiles) */
        const bool short_tile = t.tile_lo + 16u > t.arena_len;
        const uint64_t room = short_tile ? 0ull : t.arena_len - 16u - t.tile_lo;
        const uint32_t lim = room < 0xFFFFFFF0ull ? (uint32_t)room : 0xFFFFFFF0u;
        const uint8_t* const tb = short_tile ? t.arena + (t.arena_len - 16u) : t.arena + t.tile_lo;
        auto load_group = [&](auto gc) {
            constexpr int g0 = 4 * decltype(gc)::value;
            t_for_n<4>([&](auto kc) {
                constexpr int r = g0 + decltype(kc)::value;
                const uint32_t rel = (uint32_t)(wv * (int)kTWaveBytes + 1024 * r) + 16u * (uint32_t)lane;
                q[r] = stream_load16(reinterpret_cast<const u32x4*>(tb + (rel < lim ? rel : lim)));
            });
            __builtin_amdgcn_sched_barrier(0);
        };
        t.k_lo = first_k[tile];
        const uint64_t k_hi = first_k[tile + 1];
        before = load_dword_guarded(t.arena, (int64_t)wseg - 4, t.arena_len);
        __builtin_amdgcn_sched_barrier(0);
        load_group(std::integral_constant<int, 0>{});
        static_assert(kTRows >= 8 && kTRows % 4 == 0, "groups of four rows, two groups in flight");
        load_group(std::integral_constant<int, 1>{});
        const uint32_t cut_chunk = (last_tile && (t.arena_len & 15ull) != 0) ? (uint32_t)((t.arena_len - t.tile_lo) >> 4) : 0xFFFFFFFFu;

        HBS3_T_MARK(1)
        /* ---- flags: chunks a 03 may have to go into (four rows per branch, as in K12) ----------------------------- */
        uint32_t fm_lo = 0, fm_hi = 0;
        uint64_t nk_prev = 0;                                      /* rbsp_off of the NAL in front of the tile */
        uint64_t nk_off[2] = {0, 0};                               /* this thread's two NALs of the tile (numbers tid and tid + 256): rbsp_off */
        uint64_t nk_start[2] = {0, 0}, nk_pend[2] = {0, 0};        /* ... start in the caller's stream and the end of the NAL in front (gap_of) */
        uint32_t nk_len[2] = {0, 0};
        static_assert(kTFastStarts == 2 * kTThreads && kTMaxStarts >= kTFastStarts, "two per thread inside the flag pass");
        t_for_n<kTRows / 4>([&](auto gc) {
            constexpr int g0 = 4 * decltype(gc)::value;
            uint64_t fmask[4];
            t_for_n<4>([&](auto kc) {
                constexpr int k = decltype(kc)::value, r = g0 + k;
                const uint32_t e_prev = (r == 0) ? before : (uint32_t)__builtin_amdgcn_readlane((int)q[r ? r - 1 : 0].w, 63);
                const uint32_t xp = from_prev_lane(q[r].w, e_prev);
                fmask[k] = __ballot(chunk_flag(xp, q[r].x, q[r].y, q[r].z, q[r].w, 0xFFFFFFFFu));
            });
            if ((fmask[0] | fmask[1] | fmask[2] | fmask[3]) != 0) {
                t_for_n<4>([&](auto kc) {
                    constexpr int k = decltype(kc)::value, r = g0 + k;
                    if (__builtin_popcountll(fmask[k]) > kExactFlagMin) {      /* many zero pairs in this KiB: which are followed by a byte <= 3? */
                        const uint32_t e_prev = (r == 0) ? before : (uint32_t)__builtin_amdgcn_readlane((int)q[r ? r - 1 : 0].w, 63);
                        const uint32_t xp = from_prev_lane(q[r].w, e_prev);
                        fmask[k] = __ballot(((fmask[k] >> lane) & 1ull) != 0 && chunk_pattern_any_dev(xp, q[r].x, q[r].y, q[r].z, q[r].w, 0xFFFFFFFFu));
                    }
                    if (fmask[k] != 0) { write_lane_c<r>(fm_lo, (uint32_t)fmask[k]); write_lane_c<r>(fm_hi, (uint32_t)(fmask[k] >> 32)); }
                });
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (g0 == 0) {
                /* the NALs that begin in the tile: their index entries are asked for now (first_k is back: it was issued in front
                 * of the rows this group waited for) and used behind the next group */
                t.m = (uint32_t)(k_hi - t.k_lo);
                /* without a branch (clamped entry numbers; gap_of() by hand): a load inside a branch makes the compiler wait for
                 * everything in flight where the paths meet */
                nk_prev = idx[t.k_lo > 0 ? t.k_lo - 1 : 0].rbsp_off;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint64_t want = t.k_lo + (uint32_t)tid + (uint64_t)(h * kTThreads);
                    const uint64_t kk = want < n ? want : n - 1;
                    nk_off[h] = idx[kk].rbsp_off;
                    nk_len[h] = idx[kk].rbsp_len;
                    nk_start[h] = idx[kk].start;
                    nk_pend[h] = idx[kk > 0 ? kk - 1 : 0].end;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (g0 / 4 + 2 < kTRows / 4) load_group(std::integral_constant<int, g0 / 4 + 2>{});
            if constexpr (g0 == 4 || kTRows == 4) {
                /* every loaded register stays live up to here: one that is dead earlier (the upper half of an offset of which
                 * only the lower is used) is handed out as a temporary at once, and writing it waits for the load */
                asm volatile("" :: "v"(nk_off[0]), "v"(nk_start[0]), "v"(nk_pend[0]), "v"(nk_len[0]), "v"(nk_off[1]), "v"(nk_start[1]), "v"(nk_pend[1]), "v"(nk_len[1]));
                t.prev_begin = t.k_lo > 0 ? nk_prev - t.a0 : 0ull;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t j = (uint32_t)tid + (uint32_t)(h * kTThreads);
                    if (j < t.m) {
                        const uint32_t rel = (uint32_t)(nk_off[h] - t.a0 - t.tile_lo);
                        l.starts[j] = rel;
                        const uint64_t k = t.k_lo + j;
                        l.gaps[j] = (uint32_t)(gap_mode == 1 ? synth_gap(k) : nk_start[h] - (k ? nk_pend[h] : 0ull));
                        l.lens[j] = nk_len[h];
                        atomicOr(&l.rowbits[rel >> 10], 1ull << ((rel >> 4) & 63u));
                    }
                }
                if (last_tile && tid == 0 && (t.arena_len & 15ull) != 0) {  /* the arena's last, partial chunk is written bytewise */
                    const uint32_t rel = (uint32_t)(t.arena_len - t.tile_lo);
                    atomicOr(&l.rowbits[rel >> 10], 1ull << ((rel >> 4) & 63u));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        if (t.m > kTFastStarts) {
            /* a tile of small NALs: the starts past the first 512, fetched here (the rows are all in by now) */
#pragma unroll 1
            for (uint32_t j = kTFastStarts + (uint32_t)tid; j < t.m; j += (uint32_t)kTThreads) {
                const uint64_t k = t.k_lo + j;
                const uint32_t rel = (uint32_t)(idx[k].rbsp_off - t.a0 - t.tile_lo);
                l.starts[j] = rel;
                l.gaps[j] = (uint32_t)(gap_mode == 1 ? synth_gap(k) : idx[k].start - (k ? idx[k - 1].end : 0ull));
                l.lens[j] = idx[k].rbsp_len;
                atomicOr(&l.rowbits[rel >> 10], 1ull << ((rel >> 4) & 63u));
            }
        }
        __syncthreads();                                           /* rowbits, starts, gaps are complete */
        uint32_t sm_lo = 0, sm_hi = 0;                             /* lane r: chunks of row r in which a NAL begins */
        if (lane < kTRows) {
            const unsigned long long sb = l.rowbits[wv * kTRows + lane];
            sm_lo = (uint32_t)sb; sm_hi = (uint32_t)(sb >> 32);
            unsigned long long f = (((unsigned long long)fm_hi << 32) | fm_lo) | sb;
            /* chunks behind the arena's end hold nothing */
            const uint64_t row_lo = wseg + 1024ull * (uint32_t)lane;
            if (row_lo > t.arena_len) f = 0;
            else if (row_lo + 1024u > t.arena_len) f &= (2ull << ((t.arena_len - row_lo) >> 4)) - 1ull;     /* up to the chunk that holds the end */
            /* what chunk_flag() said, apart: fz = flagged (a 03 may go in), the rest of f = only a NAL start */
            const unsigned long long fz = f & (((unsigned long long)fm_hi << 32) | fm_lo);
            sm_lo = (uint32_t)(f & ~fz) ; sm_hi = (uint32_t)((f & ~fz) >> 32);        /* start only: no 03 can go into these */
            fm_lo = (uint32_t)f; fm_hi = (uint32_t)(f >> 32);
            (void)sb;
        }
        const uint32_t cnt = (lane < kTRows) ? (uint32_t)__builtin_popcount(fm_lo) + (uint32_t)__builtin_popcount(fm_hi) : 0u;
        const uint32_t inc = wave_incl_scan32(cnt, lane);
        const uint32_t local_pre = inc - cnt;
        const uint64_t rowmask = __ballot(cnt != 0u);
        /* a KiB row with every chunk flagged: a run of zeros (or padding) at least that long.  Such a tile is walked by rows too,
         * however few its elements: as elements, each chunk of the run finds the count it is entered with by walking the run back */
        const bool full_row = __ballot(lane < kTRows && (fm_lo & fm_hi) == 0xFFFFFFFFu) != 0ull;
        if (lane == 63) l.wave_tot[wv] = inc | (full_row ? 0x80000000u : 0u);
        __syncthreads();
        const uint32_t ww0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)l.wave_tot[0]), ww1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)l.wave_tot[1]);
        const uint32_t ww2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)l.wave_tot[2]), ww3 = (uint32_t)__builtin_amdgcn_readfirstlane((int)l.wave_tot[3]);
        const uint32_t wt0 = ww0 & 0x7FFFFFFFu, wt1 = ww1 & 0x7FFFFFFFu, wt2 = ww2 & 0x7FFFFFFFu, wt3 = ww3 & 0x7FFFFFFFu;
        const uint32_t nflag = wt0 + wt1 + wt2 + wt3;
        const uint32_t wave_base = (wv == 0) ? 0u : (wv == 1) ? wt0 : (wv == 2) ? wt0 + wt1 : wt0 + wt1 + wt2;
        if (nflag > kTDenseLimit || ((ww0 | ww1 | ww2 | ww3) & 0x80000000u) != 0u) {   /* dense in elements: this tile is walked by rows (k3_dense_tile) */
            d_tile = tile;
            pending = 1;
            break;
        }
        if (ahead) continue;                                       /* listed, not dense: nothing to count ahead */
        /* the tile's list: lane r writes the elements of ITS row (it holds the row's flag word), one per step -- as many steps as
         * the fullest row has elements.  (Until round 4 a step per row with elements, all lanes on one row: with NALs of 1 KiB
         * that was 48 steps a wavefront, ~8 k cycles of every tile.) */
        if (lane < kTRows) {
            uint64_t f = ((uint64_t)fm_hi << 32) | fm_lo;
            const uint64_t so = ((uint64_t)sm_hi << 32) | sm_lo;
            uint32_t j = wave_base + local_pre;
            const uint32_t row0 = (uint32_t)(64 * (kTRows * wv + lane));
#pragma unroll 1
            while (f != 0ull) {
                const uint32_t b = (uint32_t)__builtin_ctzll(f);
                const uint32_t chunk = row0 + b;     /* (the arena's partial last chunk was not seen by chunk_flag(): the general way) */
                l.list[j++] = (uint16_t)(chunk | ((((so >> b) & 1ull) && chunk != cut_chunk) ? kTListStart : kTListFlag));
                f &= f - 1ull;
            }
        }
        __syncthreads();
        HBS3_T_MARK(2)

        /* ---- wavefront 0: the elements' inserted bytes -> the tile's size -> where it starts -------------------------- */
        const uint32_t npass = (nflag + (uint32_t)kTElemPass - 1u) / (uint32_t)kTElemPass;
        uint32_t e_first = 0;
        u32x4 q_first = u32x4{0u, 0u, 0u, 0u};
        uint32_t j_first = 0;
        const uint64_t tile_bytes = t.arena_len - t.tile_lo < (uint64_t)kTTileBytes ? t.arena_len - t.tile_lo : (uint64_t)kTTileBytes;
        const bool multi = npass > 1u;
        const uint32_t stride = multi ? (uint32_t)kTElemWaves : 1u;
        if (wv == 0 || (multi && wv < kTElemWaves)) {
#pragma unroll
            for (int i = 0; i < kTParkRows; ++i) l.park[wv][i][lane] = q[kTRows - kTParkRows + i];
            unsigned long long tile_ins = 0;
#pragma unroll 1
            for (uint32_t p = (uint32_t)wv; p < npass; p += stride) {
                const uint32_t i = p * (uint32_t)kTElemPass + (uint32_t)lane;
                uint32_t e = 0;
                u32x4 qtmp;
                uint32_t jtmp;
                const bool first = p == (uint32_t)wv;
                if (i < nflag) e = tile_element<false>(t, l, l.list[i], nullptr, 0, 0, false, nullptr, first ? q_first : qtmp, first ? j_first : jtmp, false);
                if (first) e_first = e;                          /* a wavefront's first batch (nearly always the only one) is not counted again */
                const uint32_t bs = wave_sum32(e);
                tile_ins += bs;
                if (multi && lane == 0) l.bsum[p] = bs;
            }
            HBS3_T_MARK(3)
            if (multi) __syncthreads();                          /* (the wavefronts without elements: below) */
            if (wv == 0) {
                if (multi) {
                    tile_ins = 0;
#pragma unroll 1
                    for (uint32_t p = 0; p < npass; ++p) tile_ins += l.bsum[p];
                }
                if (lane == 0 && tile != 0) st_desc3(desc + tile, (tile_ins << 2) | 1ull);
                __builtin_amdgcn_s_setprio(0);
                const unsigned long long bf = k3_look_back(desc, tile, lane, err);
                HBS3_T_MARK(4)
                if (lane == 0) {
                    st_desc3(desc + tile, ((bf + tile_ins) << 2) | 2ull);
                    const uint64_t end_pos = t.tile_lo + tile_bytes + bf + tile_ins;       /* output offset behind the tile */
                    l.before = bf;
                    l.ok = end_pos <= out_cap ? 1u : 0u;
                    if (end_pos > out_cap) atomicMax(err, (uint32_t)(-HBS_E_CAPACITY));
                    if (last_tile) {
                        *total = end_pos;
                        if (idx_out) idx_out[n - 1].end = end_pos;
                    }
                    l.seg[0] = 0;
                }
            } else {
                __builtin_amdgcn_s_setprio(0);
            }
            __syncthreads();
            HBS3_T_MARK(5)
            /* ---- the bytes: the elements, 64 at a time -- every batch before anybody copies (the rows parked once; until round 3
             * each batch was followed by a copy pass of its own over all 48 rows) -- then one copy of the chunks between them: a
             * chunk with k elements in front of it goes to its arena offset + seg[k] ------------------------------------- */
            const bool can_store_e = l.ok != 0u;
            uint8_t* const tout_e = out + t.tile_lo + l.before;
            uint32_t ins_run = 0;                                   /* bytes inserted by the batches in front */
            const uint32_t np = npass ? npass : 1u;
#pragma unroll 1
            for (uint32_t p = (uint32_t)wv; p < np; p += stride) {
                if (multi) {
                    ins_run = 0;
#pragma unroll 1
                    for (uint32_t qq = 0; qq < p; ++qq) ins_run += l.bsum[qq];
                }
                const uint32_t pbase = p * (uint32_t)kTElemPass;
                const uint32_t i = pbase + (uint32_t)lane;
                uint32_t e = 0, c = 0;
                u32x4 qe = q_first;
                uint32_t je = j_first;
                if (i < nflag) { c = l.list[i]; e = p == (uint32_t)wv ? e_first : tile_element<false>(t, l, c, nullptr, 0, 0, false, nullptr, qe, je, false); }
                const uint32_t inc_e = wave_incl_scan32(e, lane);
                const uint32_t mine_before = ins_run + inc_e - e;
                if (i < nflag && (can_store_e || idx_out))
                    (void)tile_element<true>(t, l, c, tout_e, t.tile_lo + l.before, 16ull * (c & kTListChunk) + mine_before, can_store_e, idx_out, qe, je, true);
                if (i < nflag) l.seg[i + 1] = ins_run + inc_e;
            }
#pragma unroll
            for (int i2 = 0; i2 < kTParkRows; ++i2) q[kTRows - kTParkRows + i2] = l.park[wv][i2][lane];
        } else {
            /* (a barrier counts wavefronts, wherever they are in the code: these meet the ones above) */
            __builtin_amdgcn_s_setprio(0);
            if (multi) __syncthreads();
            __syncthreads();
        }
        __syncthreads();
        const bool can_store = l.ok != 0u;
        uint8_t* const tout = out + t.tile_lo + l.before;            /* where byte 0 of the tile goes when nothing is inserted in it */
        const uint32_t whole = (uint32_t)(tile_bytes >> 4);           /* chunks of the tile that are complete */
        __syncthreads();
        if (can_store) {
            const int lane = launder_lane(tid0) & 63;          /* the store addresses are formed here, not in front of the element code */
            const uint32_t cc0 = (uint32_t)(64 * kTRows * wv + lane);
            const uint32_t segv = l.seg[lane];
            const uint32_t seg64 = (uint32_t)__builtin_amdgcn_readfirstlane((int)l.seg[kTElemPass]);
            /* (asking for a row's seg word one row ahead was tried in round 4: slower, 1.10 -> 1.14 ms on 2 GiB of 1 KiB NALs) */
            t_for_n<kTRows>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const uint32_t cc = cc0 + 64u * r;
                const uint32_t rowpre = wave_base + (uint32_t)__builtin_amdgcn_readlane((int)local_pre, r);
                if (!((rowmask >> r) & 1ull)) {             /* no element in this row: one word for all lanes */
                    const uint32_t k = rowpre;
                    const uint32_t w = (k < (uint32_t)kTElemPass) ? (uint32_t)__builtin_amdgcn_readlane((int)segv, (int)(k & 63u))
                                     : (k == (uint32_t)kTElemPass) ? seg64 : (uint32_t)__builtin_amdgcn_readfirstlane((int)l.seg[k]);
                    if (cc < whole) arena_store16(tout + w + 16u * cc, q[r]);
                } else {
                    const uint64_t f = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)fm_hi, r) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)fm_lo, r);
                    const uint32_t k = rowpre + lanes_below(f);
                    if (!((f >> lane) & 1ull) && cc < whole) arena_store16(tout + l.seg[k] + 16u * cc, q[r]);
                }
                /* at most HBS3T_COPY_DEPTH stores of a wavefront in flight: see the fetch */
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(HBS3T_COPY_DEPTH) : "memory");
            });
        }
        HBS3_T_MARK(6)
    }
    if (pending == 0) break;
#ifdef HBS_DZ_TIMING
    if (threadIdx.x == 0) atomicAdd(&g_dz_cycles[5], __builtin_amdgcn_s_memtime() - dz_tile_t0);
#endif
    k3_dense_tile(l, t, d_tile, d_tile == ntiles - 1, desc, out, out_cap, idx_out, n, total, err, ahead, dz_table, call_no);
    }
    HBS3_T_FLUSH
}

} // namespace hbs
#include "hbs_emit_groups.h"
namespace hbs {

int emit_tile_grid_blocks(int device)
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return -1;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k3_tiles<0>, kTThreads, 0) != hipSuccess || per_cu < 1) return -1;
    return prop.multiProcessorCount * per_cu;
}

/* the tile kernel with its dense tiles counted ahead: the sample, the listed tiles up to their first half, then all tiles */
static void launch_tiles(const EmitArgs& a, unsigned tb, const uint32_t* probe, const uint32_t* tflag, hipStream_t st)
{
    const bool ahead = a.dz_table && a.cand_list && a.cand_cap;
    if (ahead) {
        k3t_sample<<<1024, 256, 0, st>>>(a.rbsp, a.index_in, a.n, a.cand_list, a.cand_count, a.cand_cap, a.dz_table, probe, tflag, a.call_no);
        /* (a workgroup per 64 tiles of the arena, at least 64, at most all: the list is short or empty, and a launch of workgroups
         * with 77 KiB of LDS each is not free -- 10 us for 512 of them, under the profiler) */
        const uint64_t want = a.rbsp_bytes / kTTileBytes / 64u + 1u;
        const unsigned tb_ahead = (unsigned)(want < 64u ? (tb < 64u ? tb : 64u) : (want > tb ? tb : want));
        k3_tiles<1><<<dim3(tb_ahead), kTThreads, 0, st>>>(a.rbsp, a.index_in, a.n, a.gap_mode, a.first_k, a.desc, a.ticket, a.out, a.out_cap, a.index_out, a.total, a.err,
                                                 probe, tflag, a.cand_list, a.cand_count, a.cand_cap, a.cand_ticket, a.dz_table, a.call_no, 0);
    }
    k3_tiles<0><<<dim3(tb), kTThreads, 0, st>>>(a.rbsp, a.index_in, a.n, a.gap_mode, a.first_k, a.desc, a.ticket, a.out, a.out_cap, a.index_out, a.total, a.err,
                                             probe, tflag, a.cand_list, a.cand_count, a.cand_cap, a.cand_ticket, ahead ? a.dz_table : nullptr, a.call_no, a.first_static);
}

hipError_t launch_emit_annexb(const EmitArgs& a, hipStream_t st)
{
    if (emit_takes_small_path(a.n, a.rbsp_bytes, a.two_pass)) {
        k3_small<<<1, 256, 0, st>>>(a.rbsp, a.rbsp_bytes, a.index_in, (uint32_t)a.n, a.gap_mode, a.out, a.out_cap, a.index_out, a.err, a.summary);
        return hipGetLastError();
    }
    /* one clear: the look-back words and, behind them in the workspace, total / err / ticket / n_items / total_dense / probe */
    hipError_t e = hipMemsetAsync(a.desc, 0, a.clear_bytes, st);
    if (e != hipSuccess) return e;
    const unsigned grid = 256 * 16;
    if (emit_takes_tiny_path(a.n, a.rbsp_bytes, a.two_pass, a.tiles)) {
        /* the index checked against the arena (tflag[3]) and, for means the arena tiles can hold (up to 1024 NAL starts per 192 KiB),
         * against their conditions: the tile kernel when they hold, a lane per NAL otherwise -- sizes, their scan, the bytes */
        const bool try_tiles = a.rbsp_bytes >= kTMinArena && a.rbsp_bytes / a.n >= kTilesMinMeanBytes;
        /* (means the arena tiles cannot hold: the group kernel's sizes pass checks the index itself -- one pass over it less, 0.35 ms
         * of 2.9 over 2 GiB of 64-byte NALs) */
        if (try_tiles) {
            k3t_check<<<kCheckBlocks, 256, 0, st>>>(a.rbsp, a.rbsp_bytes, a.index_in, a.n, a.gap_mode, a.first_cap, emit_desc_words(a.items_cap),
                                                    1, 0, a.tflag, a.err, a.probe, a.first_k);
            uint64_t tb = (uint64_t)a.tile_blocks;
            const uint64_t max_tiles = a.rbsp_bytes / kTTileBytes + 2;
            if (tb > max_tiles) tb = max_tiles;
            launch_tiles(a, (unsigned)tb, nullptr, a.tflag, st);
        }
        /* an index that is one stretch of the arena: 64 consecutive NALs a wavefront, cooperatively (hbs_emit_groups.h: sizes,
         * the scan of the wavefronts' sums, the bytes); the lane per NAL behind it is what other indexes get */
        {
            uint32_t gcap = groups_region_cap(a.rbsp_bytes / a.n), npw = groups_nals_per_wave(a.rbsp_bytes / a.n);
            if (const char* e = getenv("HBS_K3G_NPW")) {                 /* tuning aid: NALs a wavefront takes (16, 32 or 64) */
                const int v = atoi(e);
                if (v == 16 || v == 32 || v == 64) { npw = (uint32_t)v; gcap = groups_region_cap_for(npw, a.rbsp_bytes / a.n); }
            }
            const uint64_t nw = (a.n + npw - 1u) / npw, wgs = (nw + kGWaves - 1) / kGWaves;
            static const int cus = [] { int d = 0, c = 256; if (hipGetDevice(&d) == hipSuccess) (void)hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, d); return c; }();
            const size_t per_cu_lds = (size_t)160 * 1024 / (groups_lds_bytes(gcap) + 64);
            const uint64_t gmax_emit = (uint64_t)cus * (uint64_t)(per_cu_lds > 16 ? 16 : per_cu_lds < 1 ? 1 : per_cu_lds);
            const uint64_t gmax_sizes = (uint64_t)cus * 16u;
            k3g_sizes<<<dim3((unsigned)(wgs < gmax_sizes ? wgs : gmax_sizes)), dim3(64 * kGWaves), 0, st>>>(
                a.rbsp, a.rbsp_bytes, a.index_in, a.n, a.gap_mode, a.nal_total, a.tflag, a.err, gcap, npw, try_tiles ? 0 : 1);
            launch_scan_u64(a.nal_total, a.out_off, nw, a.total_dense, a.scan_tmp, st, nullptr, kWhenGroups, a.tflag);
            k3g_emit<<<dim3((unsigned)(wgs < gmax_emit ? wgs : gmax_emit)), dim3(64 * kGWaves), groups_lds_bytes(gcap), st>>>(
                a.rbsp, a.rbsp_bytes, a.index_in, a.n, a.gap_mode, a.out_off, reinterpret_cast<const uint8_t*>(a.nal_total + nw), a.out, a.out_cap, a.index_out, a.err, a.tflag, gcap, npw);
        }
        const uint64_t want = (a.n + 255) / 256;
        const unsigned tgrid = (unsigned)(want < 8192 ? want : 8192);
        k3_count_tiny<<<tgrid, 256, 0, st>>>(a.rbsp, a.rbsp_bytes, a.index_in, a.n, a.gap_mode, a.nal_total, a.tflag);
        launch_scan_u64(a.nal_total, a.out_off, a.n, a.total_dense, a.scan_tmp, st, nullptr, kWhenNoTiles, a.tflag);
        k3_emit_tiny<<<tgrid, 256, 0, st>>>(a.rbsp, a.rbsp_bytes, a.index_in, a.n, a.gap_mode, a.nal_total, a.out_off, a.out, a.out_cap, a.index_out, a.err, a.tflag);
        k3_summary_small<<<1, 1, 0, st>>>(a.total, a.total_dense, a.n, a.rbsp_bytes, a.err, a.summary, a.tflag, a.verdict_out);
        return hipGetLastError();
    }
    /* forced one way (HBS_EMIT_TWO_PASS=1 / =0), or -- the default -- picked on the device from a density probe */
    const uint32_t* probe = a.two_pass < 0 ? a.probe : nullptr;
    const bool want_dense = a.two_pass != 0, want_sparse = a.two_pass <= 0;
    /* arena tiles when the index allows it (decided on the device), the item kernel otherwise; path 0 pins the item kernel */
    const uint32_t* tflag = (a.n && want_sparse && a.tiles != 0 && (a.tiles == 2 || a.rbsp_bytes >= kTMinArena)) ? a.tflag : nullptr;
    /* always: it is also what checks every entry of the index against rbsp_bytes (tflag[3]) before anything follows one into the
     * arena; its last kProbeBlocks workgroups are the density probe */
    if (a.n) k3t_check<<<kCheckBlocks + (probe ? kProbeBlocks : 0u), 256, 0, st>>>(a.rbsp, a.rbsp_bytes, a.index_in, a.n, a.gap_mode, a.first_cap,
                                             emit_desc_words(a.items_cap), tflag ? 1 : 0, a.tiles == 2 ? 1 : 0, a.tflag, a.err, a.probe, a.first_k);
    /* (round 4: the chains behind the tile kernel that rule themselves out on the device were tried on side streams, forked behind
     * k3t_check and joined in front of the summary; the event waits cost more than the empty launches -- 0.461 against 0.440 ms at
     * 1 GiB.  What helps is fewer launches: k3t_reset is gone with the give-up, and the two chains share k3_sizes and one scan.) */
    const bool both = a.n && want_sparse && want_dense;              /* the automatic mode: the chains share their first step and their scan */
    if (a.n && want_sparse) {
        if (tflag) {
            /* arena tiles first; the kernel by NALs behind them runs when they do not apply */
            uint64_t tb = (uint64_t)a.tile_blocks;
            const uint64_t max_tiles = a.rbsp_bytes / kTTileBytes + 2;
            if (tb > max_tiles) tb = max_tiles;
            launch_tiles(a, (unsigned)tb, probe, tflag, st);
        }
        /* items of the kernel by NALs: segments per NAL, their exclusive scan, the item list (skipped on the device when the tile
         * kernel does the call, or when the list is the identity) */
        if (both) {
            k3_sizes<<<2048, 256, 0, st>>>(a.rbsp, a.index_in, a.n, a.gap_mode, a.nal_total, probe, a.tflag);
            launch_scan_u64(a.nal_total, a.out_off, a.n, a.n_items, a.scan_tmp, st, probe, kWhenEither, a.tflag, a.total_dense);
        } else {
            k3_seg_count<<<1024, 256, 0, st>>>(a.index_in, a.n, a.nal_total, probe, a.tflag);
            launch_scan_u64(a.nal_total, a.out_off, a.n, a.n_items, a.scan_tmp, st, probe, kWhenSparse, a.tflag);
        }
        k3_expand<<<1024, 256, 0, st>>>(a.nal_total, a.out_off, a.n, a.n_items, a.items, a.items_cap, probe, a.tflag);
        const uint64_t ngroups = (a.items_cap + kEmitGroup - 1) / kEmitGroup;       /* upper bound */
        uint64_t blocks = (uint64_t)a.grid_blocks;
        if (blocks > ngroups) blocks = ngroups;
        k3_fused<<<dim3((unsigned)blocks), 256, 0, st>>>(a.rbsp, a.rbsp_bytes, a.index_in, a.n, a.gap_mode, a.items, a.n_items, a.items_cap,
                                                         a.desc, a.ticket, a.out, a.out_cap, a.index_out, a.total, a.err, probe, tflag, a.tflag);
    }
    if (a.n && want_dense) {
        if (!both) {
            k3_count<<<grid, 256, 0, st>>>(a.rbsp, a.index_in, a.n, a.gap_mode, a.nal_total, probe, a.tflag);
            launch_scan_u64(a.nal_total, a.out_off, a.n, a.total_dense, a.scan_tmp, st, probe, kWhenDense, a.tflag);
        }
        k3_emit<<<grid, 256, 0, st>>>(a.rbsp, a.index_in, a.n, a.gap_mode, a.nal_total, a.out_off, a.out, a.out_cap, a.index_out, a.err, probe, a.tflag);
    }
    if (a.n && a.two_pass > 0) k3_summary<<<1, 1, 0, st>>>(a.total_dense, a.n, a.rbsp_bytes, a.err, a.summary, nullptr, nullptr, nullptr, a.tflag, a.verdict_out);
    else k3_summary<<<1, 1, 0, st>>>(a.total, a.n, a.rbsp_bytes, a.err, a.summary, a.total_dense, probe, a.tflag, a.tflag, a.verdict_out);
    return hipGetLastError();
}

hipError_t launch_synth_rbsp(const SynthArgs& a, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(a.err, 0, sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    if (a.n) {
        k_synth_len<<<1024, 256, 0, st>>>(a.seed, a.n, a.lens);
        launch_scan_u64(a.lens, a.offs, a.n, a.total, a.scan_tmp, st);
        k_synth_fill<<<256 * 8, 256, 0, st>>>(a.seed, a.n, a.mode, a.lens, a.offs, a.rbsp, a.rbsp_cap, a.index, a.err);
    } else {
        e = hipMemsetAsync(a.total, 0, sizeof(unsigned long long), st);
        if (e != hipSuccess) return e;
    }
    k3_summary<<<1, 1, 0, st>>>(a.total, a.n, 0, a.err, a.summary);   /* stream_bytes <- total RBSP bytes here */
    return hipGetLastError();
}

} // namespace hbs
