/*
 * hbs_emit_groups.h -- K3 for arenas of tiny NALs, cooperatively (round 6): rbsp_to_nal (h264_nal.c:92-132) over NALs of a few
 * dozen to ~230 bytes.  Included by hbs_emit.hip (device code; uses its helpers).
 *
 * The arena tiles hold at most 1024 NAL starts per 192 KiB (NALs from ~224 bytes), and below that a LANE walked each NAL
 * (k3_count_tiny / k3_emit_tiny): every load and store of a wavefront touched 64 different lines for 16 bytes each, the arena and
 * the index were read twice (sizes, a scan, the bytes) -- 0.13 of the HBM peak at 64 to 192 bytes.  An LDS-staged form of that
 * walk (round 5) was slower still: the walk is a chain of dependent byte steps whatever it reads from.
 *
 * Here nobody walks a NAL unless it needs a 03.  A wavefront takes 64 CONSECUTIVE NALs -- in an index whose NALs lie back to
 * back in the arena (k3t_check: tflag[4]) they are one stretch of it --, on its own:
 *   1. lane k reads entry k of the index (whole 16-byte halves, lane-contiguous);
 *   2. the wavefront copies its stretch of the arena into LDS with whole 16-byte loads, lane-contiguous, and asks of every chunk
 *      on the way what the scan kernels ask (chunk_flag, then chunk_pattern_any_dev): can a 00 00 {<= 3} end here?  In coded
 *      payload the answer is no for the whole stretch 997 times in 1000, and then a NAL's output is its gap (zeros + 01) and
 *      its bytes: sizes without a walk;
 *   3. sizes -> a sum per wavefront -> the three steps' exclusive scan over those sums -> where every wavefront's NALs land (two
 *      passes: k3g_sizes, the scan, k3g_emit -- see the note in front of the kernels for the single pass that was measured first);
 *   4. the OUTPUT is produced by chunks of 16 bytes aligned in the output, a lane each: a binary search over the 64 landing
 *      offsets says which NAL a chunk begins in, its bytes are the arena's at a shifted (unaligned) LDS address, the gap bytes
 *      of the one or two NALs that begin inside it are patched in -- and the store is one aligned, lane-contiguous 16-byte
 *      store.  The few bytes in front of a wavefront's first and behind its last whole chunk go out byte by byte.
 * A stretch in which some chunk may need a 03, one that does not fit the LDS share (the share is sized by the call's mean NAL),
 * or an output that would pass out_cap takes the old way for those 64 NALs: a lane per NAL, byte by byte, exact.
 */
#ifndef HBS_EMIT_GROUPS_H
#define HBS_EMIT_GROUPS_H

namespace hbs {

constexpr int kGWaves = 2;                        /* wavefronts per workgroup (each on its own: the workgroup is only how they are launched) */
constexpr uint32_t kGMaxGap = 16;                 /* gaps (zeros + 01) of up to 15 bytes; k3t_check: tflag[4]   */
constexpr uint32_t kGHead = 640;                  /* per wavefront in LDS: S[65], PA[64] (and slack)            */
constexpr uint32_t kGGuard = 32;                  /* bytes in front of the stretch's copy (unaligned 16-byte reads reach back 15) */
constexpr uint32_t kGCapMin = 2048, kGCapMax = 15360;
constexpr uint32_t kGQueue = 136;                 /* chunks of a wavefront's output that are not one NAL's payload: each of the 64 NALs' gaps (up to 15 bytes) and
                                                     beginnings touches at most two, and the wavefront's first and last chunk                                    */

/* NALs a wavefront takes, and the LDS share of its stretch, for a call whose NALs average `mean` bytes */
inline uint32_t groups_nals_per_wave(uint64_t mean) { return mean <= 100u ? 64u : 32u; }     /* (2 GiB sweep, profiles/r06/emit_groups_npw.txt: 16 never pays) */
inline uint32_t groups_region_cap_for(uint32_t npw, uint64_t mean)
{
    uint64_t cap = ((uint64_t)npw * mean * 5u / 4u + 512u + 1023u) & ~1023ull;
    if (cap < kGCapMin) cap = kGCapMin;
    if (cap > kGCapMax) cap = kGCapMax;
    return (uint32_t)cap;
}
inline uint32_t groups_region_cap(uint64_t mean) { return groups_region_cap_for(groups_nals_per_wave(mean), mean); }
/* a wavefront's share: S, PA | guard | the stretch (+ 32 readable behind it) | per output chunk: T (the NAL it begins in, a byte),
 * Src (where its sixteen bytes lie in the stretch's copy when they are one NAL's payload, else 0xFFFF; two bytes), Rk (its place in
 * the queue otherwise; a byte) | Q (the queued chunks, two bytes each) | Qv (what they come to, sixteen bytes each) */
inline __host__ __device__ uint32_t groups_table_bytes(uint32_t cap) { return (cap / 16u + 96u + 15u) & ~15u; }     /* (64 gaps of up to 15 bytes on top of the stretch's chunks) */
inline __host__ __device__ uint32_t groups_share_bytes(uint32_t cap) { return kGHead + kGGuard + cap + 32u + 4u * groups_table_bytes(cap) + 2u * kGQueue + 16u * kGQueue; }
inline size_t groups_lds_bytes(uint32_t cap) { return (size_t)kGWaves * groups_share_bytes(cap); }

/* the index is one stretch of the arena with small gaps: the group kernel does the call unless the arena tiles did */
__device__ __forceinline__ bool group_path_on(const uint32_t* tflag)
{
    return tflag && !index_bad(tflag) && !tile_path_on(tflag) && tflag[4] == 0u;
}

/* bytes [lo, hi) of a dword (positions outside 0..3 clipped) as a mask */
__device__ __forceinline__ uint32_t byte_range_mask(int lo, int hi)
{
    const uint32_t below_hi = hi <= 0 ? 0u : hi >= 4 ? 0xFFFFFFFFu : (1u << (8 * hi)) - 1u;
    const uint32_t below_lo = lo <= 0 ? 0u : lo >= 4 ? 0xFFFFFFFFu : (1u << (8 * lo)) - 1u;
    return below_hi & ~below_lo;
}

/* 16 bytes from LDS at any byte address: the two aligned 16-byte pieces they lie in (lanes that read neighbouring chunks hit
 * every bank once; five dword reads at a lane stride of 16 bytes were 8-way bank conflicts each), a dword shifter of two
 * stages, four funnel shifts.  `lds` is 16-byte aligned. */
__device__ __forceinline__ u32x4 lds_read16_unaligned(const uint8_t* lds, uint32_t at)
{
    const u32x4* p = reinterpret_cast<const u32x4*>(lds + (at & ~15u));
    const u32x4 a = p[0], b = p[1];
    const bool s1 = (at & 4u) != 0u, s2 = (at & 8u) != 0u;
    /* t[i] = D[i + (s1 ? 1 : 0)], i = 0 .. 6, D = a.x a.y a.z a.w b.x b.y b.z b.w */
    const uint32_t t0 = s1 ? a.y : a.x, t1 = s1 ? a.z : a.y, t2 = s1 ? a.w : a.z, t3 = s1 ? b.x : a.w, t4 = s1 ? b.y : b.x, t5 = s1 ? b.z : b.y, t6 = s1 ? b.w : b.z;
    /* e[i] = t[i + (s2 ? 2 : 0)], i = 0 .. 4 */
    const uint32_t e0 = s2 ? t2 : t0, e1 = s2 ? t3 : t1, e2 = s2 ? t4 : t2, e3 = s2 ? t5 : t3, e4 = s2 ? t6 : t4;
    const uint32_t sh = at & 3u;
    u32x4 v;
    v.x = alignbyte(e1, e0, sh); v.y = alignbyte(e2, e1, sh); v.z = alignbyte(e3, e2, sh); v.w = alignbyte(e4, e3, sh);
    return v;
}

/* what a wavefront knows about its 64 NALs once their sizes are out */
struct GSizes {
    uint64_t off;               /* my NAL: arena offset, length, gap, bytes it takes in all, the same summed over the lanes below */
    uint32_t len, gap;
    uint64_t tot, excl, wtot;
    uint64_t a0al;              /* the stretch's first aligned chunk, its chunks */
    uint32_t nch, cnt;
    bool have, fits, dirty;
};

constexpr int kGBatch = 4;      /* chunk loads of a lane in flight (a load per step was a memory round trip per KiB of the stretch) */

/* entries k0 .. k0 + 63 of the index, the stretch's flags, the sizes; kStage: the stretch's chunks go to `region` (LDS) on the way */
template <bool kStage>
__device__ __forceinline__ void groups_sizes(GSizes& z, uint64_t k0, uint32_t npw, int lane, const uint8_t* __restrict__ rbsp, uint64_t rbsp_bytes,
                                             const hbs_nal_entry* __restrict__ idx, uint64_t n, int gap_mode, uint32_t cap, uint8_t* region,
                                             uint32_t* __restrict__ vflag = nullptr, uint32_t* __restrict__ err = nullptr, int known_dirty = -1)
{
    const uint64_t k = k0 + (uint64_t)lane;
    z.have = (uint32_t)lane < npw && k < n;
    z.cnt = k0 >= n ? 0u : (n - k0 < (uint64_t)npw ? (uint32_t)(n - k0) : npw);
    uint64_t e_start = 0, e_end = 0;
    z.off = 0; z.len = 0;
    /* the entry in front of the wavefront's first (lane 0: the gap's other end, and the contiguity check) is asked for together
     * with the wavefront's own: behind them it was one more round trip a step */
    u32x4 p0 = u32x4{0u, 0u, 0u, 0u}, p1 = u32x4{0u, 0u, 0u, 0u};
    const bool want_prev = lane == 0 && k != 0 && z.have && (gap_mode != 1 || vflag != nullptr);
    if (want_prev) {
        p1 = reinterpret_cast<const u32x4*>(idx + (k - 1))[1];
        if (gap_mode != 1) p0 = reinterpret_cast<const u32x4*>(idx + (k - 1))[0];
    }
    if (z.have) {
        const u32x4 h1 = reinterpret_cast<const u32x4*>(idx + k)[1];
        z.off = ((uint64_t)h1.y << 32) | h1.x; z.len = h1.z;
        if (gap_mode != 1) {
            const u32x4 h0 = reinterpret_cast<const u32x4*>(idx + k)[0];
            e_start = ((uint64_t)h0.y << 32) | h0.x; e_end = ((uint64_t)h0.w << 32) | h0.z;
        }
    }
    uint64_t gap64;
    if (gap_mode == 1) gap64 = synth_gap(k);
    else {
        uint64_t prev_end = (uint64_t)__shfl_up((unsigned long long)e_end, 1, 64);
        if (lane == 0) prev_end = want_prev ? (((uint64_t)p0.w << 32) | p0.z) : 0ull;
        gap64 = e_start - prev_end;
    }
    if (vflag) {
        /* k3t_check's part for this route (the sizes pass is then the call's first look at the index: nothing has followed it into
         * the arena yet, and every load below is guarded by rbsp_bytes): entries outside the caller's buffer (tflag[3], HBS_E_ARG),
         * NALs that do not lie back to back or gaps of 16 bytes and more (tflag[4]: the lane per NAL takes the call) */
        const uint64_t endk = z.off + z.len;
        uint64_t prev_stop = (uint64_t)__shfl_up((unsigned long long)endk, 1, 64);
        if (lane == 0 && want_prev) prev_stop = (((uint64_t)p1.y << 32) | p1.x) + p1.z;
        const bool outside = z.have && (z.off > rbsp_bytes || (uint64_t)z.len > rbsp_bytes - z.off);
        const bool apart = z.have && ((k != 0 && z.off != prev_stop) || gap64 >= (uint64_t)kGMaxGap);
        const bool wave_outside = __ballot(outside) != 0ull;
        if (wave_outside && lane == 0) { atomicOr(&vflag[3], 1u); atomicMax(err, (uint32_t)(-HBS_E_ARG)); }
        if (__ballot(apart) != 0ull && lane == 0) atomicOr(&vflag[4], 1u);
        /* the call ends with HBS_E_ARG: nothing of these entries is followed (no second way out of this function: with a `return` here
         * the compiler kept z in scratch memory, and the pass took 0.87 ms instead of 0.54 over 2 GiB of 64-byte NALs) */
        if (wave_outside) { z.cnt = 0; z.have = false; z.len = 0; gap64 = 0; }
    }
    z.gap = z.have ? (uint32_t)gap64 : 0u;
    z.fits = false; z.dirty = false; z.a0al = 0; z.nch = 0;
    if (z.cnt) {
        const uint64_t a0 = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)z.off) | ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(z.off >> 32)) << 32);
        const uint64_t endl = z.off + z.len;
        const uint64_t a1 = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)endl, (int)z.cnt - 1) |
                            ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(endl >> 32), (int)z.cnt - 1) << 32);
        z.a0al = a0 & ~15ull;
        const uint64_t span = a1 - z.a0al;
        z.fits = span + 16u <= (uint64_t)cap;
        if (z.fits) {
            z.nch = (uint32_t)((span + 15u) >> 4);
            uint32_t e_prev = 0xFFFFFFFFu;                                 /* (bytes in front of the first chunk belong to other NALs) */
            uint64_t any = 0;
            const bool whole = z.a0al + 16ull * z.nch <= rbsp_bytes;       /* every chunk inside the arena: plain loads */
            if (kStage && lane == 0) { *reinterpret_cast<u32x4*>(region - 16) = u32x4{~0u, ~0u, ~0u, ~0u}; *reinterpret_cast<u32x4*>(region - 32) = u32x4{~0u, ~0u, ~0u, ~0u}; }
#pragma unroll 1
            for (uint32_t c0 = 0; c0 < z.nch; c0 += 64u * kGBatch) {
                u32x4 q[kGBatch];
#pragma unroll
                for (int i = 0; i < kGBatch; ++i) {
                    const uint32_t c = c0 + 64u * (uint32_t)i + (uint32_t)lane;
                    q[i] = u32x4{~0u, ~0u, ~0u, ~0u};
                    if (c < z.nch) q[i] = whole ? k3_load16(rbsp + z.a0al + 16ull * c) : load_chunk_guarded(rbsp, z.a0al + 16ull * c, rbsp_bytes);
                }
#pragma unroll
                for (int i = 0; i < kGBatch; ++i) {
                    const uint32_t c = c0 + 64u * (uint32_t)i + (uint32_t)lane;
                    if (kStage && c < z.nch) *reinterpret_cast<u32x4*>(region + 16u * c) = q[i];
                    if (!kStage) {                                         /* (the second pass stages only: the first one left its verdict) */
                        const uint32_t xp = from_prev_lane(q[i].w, e_prev);
                        const bool f = c < z.nch && chunk_flag(xp, q[i].x, q[i].y, q[i].z, q[i].w, 0xFFFFFFFFu) &&
                                       chunk_pattern_any_dev(xp, q[i].x, q[i].y, q[i].z, q[i].w, 0xFFFFFFFFu);
                        any |= __ballot(f);
                        e_prev = (uint32_t)__builtin_amdgcn_readlane((int)q[i].w, 63);
                    }
                }
            }
            if (kStage && lane == 0) *reinterpret_cast<u32x4*>(region + 16u * z.nch) = u32x4{~0u, ~0u, ~0u, ~0u};     /* (reads reach 15 bytes past the last chunk) */
            z.dirty = kStage ? known_dirty != 0 : any != 0ull;
        }
    }
    /* bytes that go in: none unless some chunk of the stretch may take one (or the stretch is too long for its LDS share) */
    uint32_t ins = 0;
    if (z.have && (!z.fits || z.dirty)) ins = tiny_count(rbsp, rbsp_bytes, z.off, z.len);
    z.tot = z.have ? (uint64_t)z.gap + z.len + ins : 0ull;
    const unsigned long long inc = wave_incl_scan_u64(z.tot, lane);
    z.wtot = __shfl(inc, 63, 64);
    z.excl = inc - z.tot;
}

/* Two passes, every wavefront on its own (no ticket, no barrier, no look-back):
 *   k3g_sizes   the bytes each wavefront's 64 NALs take (entries of the index, the stretch's flags)  -> gsum[]
 *   (scan)      the three steps' own exclusive scan over the n / 64 sums                             -> gbase[], the total
 *   k3g_emit    the entries and the stretch again, staged in LDS this time, and the bytes out.
 * The single-pass form -- a workgroup of four wavefronts per ticket, sizes published into a look-back word, k3_look_back -- was
 * built first and measured (profiles/r06, scripts/experiments/README.md): 0.19 of peak at 64-byte NALs, 0.17 with the sizes of
 * the next ticket published a step ahead.  A unit of 256 NALs is 34 KB of traffic behind three dependent round trips, a ticket
 * and three barriers: 12 us a step even with the look-back and the output left out, and a CU holds three such workgroups.
 * Reading the arena and the index twice (+ 60 % traffic) buys units that do not wait for each other. */
/* npw: NALs a wavefront takes (64 or 32: the call's mean NAL decides, so that a stretch is 4-8 KiB whatever the NALs' size --
 * the LDS share is what bounds the wavefronts a CU holds in the second pass); validate: this pass is the call's index check too */
__global__ __launch_bounds__(64 * kGWaves)
void k3g_sizes(const uint8_t* __restrict__ rbsp, uint64_t rbsp_bytes, const hbs_nal_entry* __restrict__ idx, uint64_t n, int gap_mode,
               unsigned long long* __restrict__ gsum, uint32_t* __restrict__ tflag, uint32_t* __restrict__ err, uint32_t cap, uint32_t npw, int validate)
{
    if (!validate && !group_path_on(tflag)) return;
    const int lane = threadIdx.x & 63;
    const uint64_t nw = (n + (uint64_t)npw - 1) / (uint64_t)npw;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
#pragma unroll 1
    for (uint64_t w = wave; w < nw; w += nwaves) {
        GSizes z;
        groups_sizes<false>(z, w * (uint64_t)npw, npw, lane, rbsp, rbsp_bytes, idx, n, gap_mode, cap, nullptr, validate ? tflag : nullptr, err);
        if (lane == 0) { gsum[w] = z.wtot; reinterpret_cast<uint8_t*>(gsum + nw)[w] = z.dirty ? 1u : 0u; }    /* (the flags live behind the sums) */
    }
}

__global__ __launch_bounds__(64 * kGWaves)
void k3g_emit(const uint8_t* __restrict__ rbsp, uint64_t rbsp_bytes, const hbs_nal_entry* __restrict__ idx, uint64_t n, int gap_mode,
              const unsigned long long* __restrict__ gbase, const uint8_t* __restrict__ gdirty, uint8_t* __restrict__ out, uint64_t out_cap, hbs_nal_entry* __restrict__ idx_out,
              uint32_t* __restrict__ err, const uint32_t* __restrict__ tflag, uint32_t cap, uint32_t npw)
{
    if (!group_path_on(tflag)) return;
    extern __shared__ __attribute__((aligned(16))) uint8_t g_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint8_t* const mine = g_lds + (size_t)wv * groups_share_bytes(cap);
    uint32_t* const S = reinterpret_cast<uint32_t*>(mine);                 /* [65]: where NAL k of the wavefront lands, from the wavefront's first output byte */
    uint32_t* const PA = S + 66;                                           /* [64]: its first byte in the stretch's copy | gap << 20                          */
    uint8_t* const region = mine + kGHead + kGGuard;                       /* the stretch from its first aligned chunk on; kGGuard bytes in front readable      */
    uint8_t* const T = region + cap + 32u;                                 /* [output chunks of the wavefront]: the NAL the chunk begins in                     */
    uint16_t* const Src = reinterpret_cast<uint16_t*>(T + groups_table_bytes(cap));
    uint8_t* const Rk = T + 3u * groups_table_bytes(cap);
    uint16_t* const Q = reinterpret_cast<uint16_t*>(T + 4u * groups_table_bytes(cap));   /* the chunks that are not one NAL's payload, in the order met       */
    u32x4* const Qv = reinterpret_cast<u32x4*>(T + 4u * groups_table_bytes(cap) + 2u * kGQueue);
    const uint64_t nw = (n + (uint64_t)npw - 1) / (uint64_t)npw;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
#pragma unroll 1
    for (uint64_t w = wave; w < nw; w += nwaves) {
        const unsigned long long wbase = gbase[w];                         /* (asked for first: the entries and the stretch do not wait for it) */
        const int was_dirty = (int)gdirty[w];                              /* the sizes pass's verdict on this stretch: nothing is flagged twice */
        GSizes z;
        groups_sizes<true>(z, w * (uint64_t)npw, npw, lane, rbsp, rbsp_bytes, idx, n, gap_mode, cap, region, nullptr, nullptr, was_dirty);
        const uint64_t k = w * (uint64_t)npw + (uint64_t)lane;
        const bool clean = z.fits && !z.dirty;
        const uint64_t my_base = wbase + z.excl;                           /* my NAL's gap begins here */
        if (idx_out && z.have) {
            u32x4 h0, h1;
            const uint64_t ns = my_base + z.gap, ne = my_base + z.tot;
            h0.x = (uint32_t)ns; h0.y = (uint32_t)(ns >> 32); h0.z = (uint32_t)ne; h0.w = (uint32_t)(ne >> 32);
            h1.x = (uint32_t)z.off; h1.y = (uint32_t)(z.off >> 32); h1.z = z.len; h1.w = 0u;
            reinterpret_cast<u32x4*>(idx_out + k)[0] = h0;
            reinterpret_cast<u32x4*>(idx_out + k)[1] = h1;
        }
        const bool over = wbase + z.wtot > out_cap;                        /* wave-uniform */
        if (!clean || over) {
            /* the old way for these 64 NALs: a lane per NAL, byte by byte (k3_emit_tiny) */
            if (z.have) {
                if (my_base + z.tot > out_cap) { atomicMax(err, (uint32_t)(-HBS_E_CAPACITY)); }
                else {
                    TinyOut o;
                    o.dst = out + my_base; o.lo = o.hi = 0ull; o.have = 0u;
                    for (uint32_t zz = 0; zz + 1 < z.gap; ++zz) o.put(0u);
                    if (z.gap) o.put(1u);
                    uint32_t count = 0;
                    tiny_bytes(rbsp, rbsp_bytes, z.off, z.len, [&](uint32_t v) {
                        if (count == 2u && v <= 3u) { o.put(3u); count = 0u; }
                        o.put(v);
                        count = v == 0u ? count + 1u : 0u;
                    });
                    o.flush();
                }
            }
            continue;
        }
        S[lane] = (uint32_t)z.excl;
        if (lane == 63) { S[64] = (uint32_t)z.wtot; S[65] = (uint32_t)z.wtot; }
        PA[lane] = (uint32_t)(z.off - z.a0al) | (z.gap << 20);

        /* the output by aligned chunks */
        const uint32_t cnt = z.cnt;
        const uint32_t w32 = (uint32_t)z.wtot;
        const uint64_t o0 = wbase & ~15ull;                                /* the aligned chunk my first byte lies in */
        const int32_t lead = (int32_t)(wbase - o0);                        /* bytes of it that are not mine */
        const uint32_t nout = (uint32_t)((lead + w32 + 15u) >> 4);
        /* T[c] = the NAL in which chunk c begins (chunk 0: in which the wavefront's output begins): every lane enters its NAL for
         * the chunks that begin inside it -- four or five at 64 bytes -- where a binary search over S per chunk was six dependent
         * LDS reads and half the pass's instructions */
        if (z.tot != 0ull) {
            const uint32_t s0 = (uint32_t)z.excl, s1 = (uint32_t)(z.excl + z.tot);
            const uint32_t c_lo = s0 == 0u ? 0u : (s0 + (uint32_t)lead + 15u) >> 4;
            const uint32_t c_hi = (s1 + (uint32_t)lead + 15u) >> 4;
            for (uint32_t c = c_lo; c < c_hi; ++c) T[c] = (uint8_t)lane;
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        /* Three passes over the chunks.  (a) What each is: the sixteen bytes of one NAL's payload (three chunks in four at 64
         * bytes) -- then Src[c] says where they lie in the stretch's copy -- or a chunk in which a NAL begins (gap bytes, two
         * payloads): queued.  (b) The queue, every lane on such a chunk, results into Qv.  (c) Every chunk out, lane-contiguous:
         * whole lines.  With both kinds in one loop every step of the wavefront paid for both (1 400 instructions a stretch: the
         * pass was bound by them); with the queued chunks stored by a pass of their own every line was written twice, in halves. */
        uint32_t qn = 0;
#pragma unroll 1
        for (uint32_t c0 = 0; c0 < nout; c0 += 64u) {
            const uint32_t c = c0 + (uint32_t)lane;
            const int32_t p0 = (int32_t)(16u * c) - lead;                  /* position of the chunk's first byte in the wavefront's output */
            bool later = false;
            if (c < nout) {
                const uint32_t j = T[c];
                const uint32_t sj = S[j], sn = S[j + 1], pa = PA[j];
                const int32_t pay = (int32_t)(sj + (pa >> 20)) - p0;
                later = !(pay <= 0 && (int32_t)sn >= p0 + 16);
                Src[c] = later ? (uint16_t)0xFFFFu : (uint16_t)((int32_t)(pa & 0xFFFFFu) - pay + (int32_t)kGGuard);
            }
            const uint64_t lm = __ballot(later);
            if (later) { const uint32_t r = qn + lanes_below(lm); Q[r] = (uint16_t)c; Rk[c] = (uint8_t)r; }
            qn += (uint32_t)__builtin_popcountll(lm);
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
        for (uint32_t i = (uint32_t)lane; i < qn; i += 64u) {
            const uint32_t c = Q[i];
            const int32_t p0 = (int32_t)(16u * c) - lead;
            uint32_t j = T[c];
            uint32_t sj = S[j], sn = S[j + 1], pa = PA[j];
            u32x4 v = u32x4{0u, 0u, 0u, 0u};
#pragma unroll 1
            for (;;) {
                const uint32_t gj = pa >> 20, aj = pa & 0xFFFFFu;
                const int32_t pay = (int32_t)(sj + gj) - p0;               /* chunk-relative position of NAL j's first byte (its 01 at pay - 1) */
                const int32_t end = (int32_t)sn - p0;                      /* ... of the byte behind its last                                    */
                if (end > 0 && pay < 16 && end > pay) {
                    const u32x4 src = lds_read16_unaligned(region - kGGuard, (uint32_t)((int32_t)aj - pay + (int32_t)kGGuard));
                    v.x |= src.x & byte_range_mask(pay, end);
                    v.y |= src.y & byte_range_mask(pay - 4, end - 4);
                    v.z |= src.z & byte_range_mask(pay - 8, end - 8);
                    v.w |= src.w & byte_range_mask(pay - 12, end - 12);
                }
                if (gj != 0u && pay >= 1 && pay <= 16) {                   /* the 01 that closes the gap (the zeros in front of it are v's zeros) */
                    const uint32_t b = (uint32_t)(pay - 1);
                    const uint32_t one = 1u << (8u * (b & 3u));
                    if ((b >> 2) == 0u) v.x |= one; else if ((b >> 2) == 1u) v.y |= one; else if ((b >> 2) == 2u) v.z |= one; else v.w |= one;
                }
                if (end >= 16 || j + 1 >= cnt) break;
                ++j; sj = sn; sn = S[j + 1]; pa = PA[j];
            }
            Qv[i] = v;
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
        for (uint32_t c = (uint32_t)lane; c < nout; c += 64u) {
            const int32_t p0 = (int32_t)(16u * c) - lead;
            const uint32_t sx = Src[c];
            const u32x4 v = sx != 0xFFFFu ? lds_read16_unaligned(region - kGGuard, sx) : Qv[Rk[c]];
            uint8_t* const dst = out + o0 + 16ull * c;
            if (p0 >= 0 && (uint32_t)p0 + 16u <= w32) {
                k3_store16(dst, v);
            } else {
                /* my bytes of a chunk shared with the wavefront in front or behind */
                const int32_t lo = p0 < 0 ? -p0 : 0, hi = (int32_t)w32 - p0 < 16 ? (int32_t)w32 - p0 : 16;
                uint32_t w4[4] = {v.x, v.y, v.z, v.w};
                for (int32_t b = lo; b < hi; ++b) dst[b] = (uint8_t)(w4[b >> 2] >> (8 * (b & 3)));
            }
        }
        __builtin_amdgcn_wave_barrier();                                   /* (the next stretch overwrites the LDS share) */
    }
}

} // namespace hbs
#endif
