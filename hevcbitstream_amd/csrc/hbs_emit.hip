/*
 * hbs_emit.hip -- K3: RBSP arena + NAL index -> Annex-B stream (emulation
 * prevention insertion and start codes), and the device generator of the
 * synthetic stream S(seed, n_nals, mode).
 *
 * Replaces rbsp_to_nal() per NAL (reference h264_nal.c:92-132, called from
 * write_hevc_nal_unit, hevc_stream.c:1324-1327) plus the start-code bytes the
 * reference's callers put in front of each NAL.
 *
 * One wavefront per NAL, a row of 1 KiB at a time (lane l = 16 bytes at 16 l,
 * the next row's load in flight).  A 03 is only ever inserted behind two zero
 * bytes, so chunk_flag() (hbs_sparse.h: no two adjacent zeros start in bytes
 * [-2, 16) of the chunk) clears nearly every chunk for a plain copy; flagged
 * chunks run the byte-exact rbsp_to_nal rules of hbs_emit.h.  Pass 1 counts the
 * inserted bytes per NAL, a scan turns NAL sizes into output offsets, pass 2
 * copies into byte-aligned 16-byte stores.  Traffic: RBSP read twice, stream
 * written once (3 B/B).
 */
#include <hip/hip_runtime.h>
#include "hbs_wave.h"
#include "hbs_sparse.h"
#include "hbs_emit.h"
#include "hbs_emit_launch.h"

namespace hbs {

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

/* One row of a NAL: which of its chunks may need a 03 (conservative), from the row's registers. */
struct RowFlags {
    bool mine;             /* my chunk */
    uint64_t mask;         /* the row's  */
};
__device__ __forceinline__ RowFlags row_flags(const u32x4& q, uint32_t e_prev, uint32_t e_next, uint32_t off, uint32_t len)
{
    const uint32_t xp = from_prev_lane(q.w, e_prev);
    const uint32_t xn = from_next_lane(q.x, e_next);
    RowFlags r;
    r.mine = off < len && chunk_flag(xp, q.x, q.y, q.z, q.w, xn);
    r.mask = __ballot(r.mine);
    return r;
}

__global__ __launch_bounds__(256)
void k3_count(const uint8_t* __restrict__ rbsp, const hbs_nal_entry* __restrict__ idx, uint64_t n, int gap_mode,
              unsigned long long* __restrict__ nal_total)
{
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t k = wave; k < n; k += nwaves) {
        const uint64_t begin = idx[k].rbsp_off;
        const uint32_t len = idx[k].rbsp_len;
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
                if (f.mine) c = count_segment(rbsp, begin, begin + off, begin + (off + 16u < len ? off + 16u : len));
                ins += wave_sum_u32(c);
            }
            e_prev = (uint32_t)__builtin_amdgcn_readlane((int)q.w, 63);
        }
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
void k_scan_reduce(const unsigned long long* __restrict__ v, uint64_t n, unsigned long long* __restrict__ part)
{
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
void k_scan_parts(unsigned long long* __restrict__ part, unsigned long long* __restrict__ total)
{
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
                  const unsigned long long* __restrict__ part)
{
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

static void launch_scan_u64(const unsigned long long* v, unsigned long long* out, uint64_t n, unsigned long long* total,
                            unsigned long long* part /* kScanBlocks entries */, hipStream_t st)
{
    k_scan_reduce<<<kScanBlocks, 256, 0, st>>>(v, n, part);
    k_scan_parts<<<1, kScanBlocks, 0, st>>>(part, total);
    k_scan_apply<<<kScanBlocks, 256, 0, st>>>(v, out, n, part);
}

__global__ __launch_bounds__(256)
void k3_emit(const uint8_t* __restrict__ rbsp, const hbs_nal_entry* __restrict__ idx, uint64_t n, int gap_mode,
             const unsigned long long* __restrict__ nal_total, const unsigned long long* __restrict__ out_off,
             uint8_t* __restrict__ out, uint64_t out_cap, hbs_nal_entry* __restrict__ idx_out, uint32_t* __restrict__ err)
{
    struct __attribute__((packed, aligned(1))) U16 { u32x4 v; };
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t k = wave; k < n; k += nwaves) {
        const uint64_t begin = idx[k].rbsp_off;
        const uint32_t len = idx[k].rbsp_len;
        const uint32_t nrows = (len + 1023u) / 1024u;
        const uint64_t gap = gap_of(idx, k, gap_mode);
        const uint64_t base = out_off[k];
        const uint64_t nal_start = base + gap;
        const uint64_t nal_end = base + nal_total[k];
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
                    const uint64_t se = begin + (off + 16u < len ? off + 16u : len);
                    if (f.mine) c = count_segment(rbsp, begin, begin + off, se);
                    uint32_t tot;
                    dst += wave_excl_scan_u32(c, lane, tot);
                    ins += tot;
                    if (f.mine) emit_segment(rbsp, begin, begin + off, se, dst);
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
}

__global__ void k3_summary(const unsigned long long* total, uint64_t n, uint64_t rbsp_bytes, const uint32_t* err, hbs_summary* sum)
{
    sum->nal_count = n; sum->nal_found = n; sum->rbsp_bytes = rbsp_bytes; sum->stream_bytes = *total;
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

hipError_t launch_emit_annexb(const EmitArgs& a, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(a.err, 0, sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    const unsigned grid = 256 * 16;
    if (a.n) {
        k3_count<<<grid, 256, 0, st>>>(a.rbsp, a.index_in, a.n, a.gap_mode, a.nal_total);
        launch_scan_u64(a.nal_total, a.out_off, a.n, a.total, a.scan_tmp, st);
        k3_emit<<<grid, 256, 0, st>>>(a.rbsp, a.index_in, a.n, a.gap_mode, a.nal_total, a.out_off, a.out, a.out_cap, a.index_out, a.err);
    } else {
        e = hipMemsetAsync(a.total, 0, sizeof(unsigned long long), st);
        if (e != hipSuccess) return e;
    }
    k3_summary<<<1, 1, 0, st>>>(a.total, a.n, a.rbsp_bytes, a.err, a.summary);
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
