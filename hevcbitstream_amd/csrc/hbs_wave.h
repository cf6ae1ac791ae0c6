/*
 * hbs_wave.h -- wavefront-level helpers shared by the event-sparse kernels
 * (hbs_scan4.hip, hbs_scan5.hip): guarded loads at the stream edges, streaming
 * load / store forms, DPP neighbour access, wave scans, byte-aligned stores, and
 * the look-back descriptor accessors.  Device only.
 */
#ifndef HBS_WAVE_H
#define HBS_WAVE_H

#include <hip/hip_runtime.h>
#include "hbs_chunk.h"

namespace hbs {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_u1 __attribute__((aligned(1)));        /* a 16-byte access at any byte address */

/* Streaming accesses.  The stream is read once and the arena written once, so neither should displace
 * anything in the caches: `nt` loads read at 6.9 TB/s where default-policy loads of the same pattern read
 * at 6.2 (scripts/ubench/ceiling2.hip, profiles/r02/ceiling2.txt). */
#ifndef HBS_NT_LOAD
#define HBS_NT_LOAD 1
#endif
#ifndef HBS_NT_STORE
#define HBS_NT_STORE 1
#endif
/* The pointer is cast to the global address space: where hipcc cannot prove that (a pointer picked between two
 * buffers, as the last tile's padded copy is) it emits flat_load, which counts on lgkmcnt too and so is waited
 * for by every `s_waitcnt lgkmcnt(0)` in front of a barrier. */
typedef const __attribute__((address_space(1))) u32x4* global_u32x4_ptr;
typedef const __attribute__((address_space(1))) uint32_t* global_u32_ptr;
__device__ __forceinline__ u32x4 stream_load16(const u32x4* p)
{
    const global_u32x4_ptr g = (global_u32x4_ptr)(uintptr_t)p;
#if HBS_NT_LOAD
    return __builtin_nontemporal_load(g);
#else
    return *g;
#endif
}
__device__ __forceinline__ uint32_t stream_load4(const uint8_t* p)
{
    return *(global_u32_ptr)(uintptr_t)p;
}
__device__ __forceinline__ void arena_store16(uint8_t* p, u32x4 v)
{
#if HBS_NT_STORE
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4_u1*>(p));
#else
    *reinterpret_cast<u32x4_u1*>(p) = v;
#endif
}

/* Loads that may straddle either end of the stream (bytes outside read as 0xFF).  Rolled byte
 * loops on purpose: these run for the one tile that holds the stream end and for elements next
 * to it, and unrolled they would be most of the kernel's code. */
__device__ __forceinline__ uint32_t load_dword_guarded(const uint8_t* __restrict__ s, int64_t g, uint64_t n)
{
    if (g >= 0 && (uint64_t)g + 4 <= n) return *reinterpret_cast<const uint32_t*>(s + g);
    uint32_t v = 0xFFFFFFFFu;
#pragma unroll 1
    for (int b = 0; b < 4; ++b) {
        const int64_t q = g + b;
        if (q >= 0 && (uint64_t)q < n) v = (v & ~(0xFFu << (8 * b))) | ((uint32_t)s[q] << (8 * b));
    }
    return v;
}

__device__ __forceinline__ u32x4 load_chunk_guarded(const uint8_t* __restrict__ s, uint64_t g, uint64_t n)
{
    if (g + 16 <= n) return *reinterpret_cast<const u32x4*>(s + g);
    uint32_t w0 = 0xFFFFFFFFu, w1 = 0xFFFFFFFFu, w2 = 0xFFFFFFFFu, w3 = 0xFFFFFFFFu;
#pragma unroll 1
    for (uint32_t b = 0; b < 16; ++b) {
        if (g + b < n) {
            const uint32_t m = ~(0xFFu << (8u * (b & 3u)));
            const uint32_t x = (uint32_t)s[g + b] << (8u * (b & 3u));
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

/* Opaque copy of a lane-constant value: hipcc otherwise hoists every address and predicate that
 * only depends on the lane out of the tile loop and spills the lot around it. */
__device__ __forceinline__ int launder_lane(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}

/* lane l <- value of lane l-1 (lane 0 <- edge) / lane l+1 (lane 63 <- edge) */
__device__ __forceinline__ uint32_t from_prev_lane(uint32_t v, uint32_t edge)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t from_next_lane(uint32_t v, uint32_t edge)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
}

/* DPP moves for scans over the 64 lanes (gfx9: rows of 16 lanes; row_shr within a row, row_bcast15 / row_bcast31 across rows):
 * a lane without a source, or in a row outside kRowMask, reads 0 -- the identity of a sum, and of the tile algebra (kKindNone = 0).
 * One VALU instruction each, where __shfl_up is address arithmetic plus an LDS permute. */
constexpr int kDppRowShr1 = 0x111, kDppRowShr2 = 0x112, kDppRowShr4 = 0x114, kDppRowShr8 = 0x118;
constexpr int kDppBcast15 = 0x142, kDppBcast31 = 0x143, kDppWaveShr1 = 0x138;
template <int kCtrl, int kRowMask>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, kCtrl, kRowMask, 0xF, true);
}

/* chunk_pattern_any() (hbs_chunk.h) in half the instructions: does a pattern 00 00 {<=3} end in bytes [0, 18) of the chunk?
 * For each of the 18 places p in [-2, 16) one v_perm_b32 forms b[p] << 16 | b[p+1] << 8 | b[p+2] from the two dwords the three
 * bytes lie in; a pattern is a value of at most 3, so the minimum over the places (v_min3_u32, two places an instruction)
 * is the whole test. */
__device__ __forceinline__ bool chunk_pattern_any_dev(uint32_t xp, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t xn)
{
#define HBS_P(hi, lo, p) __builtin_amdgcn_perm((hi), (lo), 0x0C000102u + 0x00010101u * (p))   /* bytes p, p+1, p+2 of lo:hi, first byte on top */
#define HBS_MIN3(a, b, c) __builtin_elementwise_min(__builtin_elementwise_min((a), (b)), (c))
    uint32_t m = HBS_MIN3(HBS_P(x0, xp, 2u), HBS_P(x0, xp, 3u), HBS_P(x1, x0, 0u));
    m = HBS_MIN3(m, HBS_P(x1, x0, 1u), HBS_P(x1, x0, 2u));
    m = HBS_MIN3(m, HBS_P(x1, x0, 3u), HBS_P(x2, x1, 0u));
    m = HBS_MIN3(m, HBS_P(x2, x1, 1u), HBS_P(x2, x1, 2u));
    m = HBS_MIN3(m, HBS_P(x2, x1, 3u), HBS_P(x3, x2, 0u));
    m = HBS_MIN3(m, HBS_P(x3, x2, 1u), HBS_P(x3, x2, 2u));
    m = HBS_MIN3(m, HBS_P(x3, x2, 3u), HBS_P(xn, x3, 0u));
    m = HBS_MIN3(m, HBS_P(xn, x3, 1u), HBS_P(xn, x3, 2u));
    m = __builtin_elementwise_min(m, HBS_P(xn, x3, 3u));
#undef HBS_MIN3
#undef HBS_P
    return m <= 3u;
}

__device__ __forceinline__ uint32_t wave_incl_scan32(uint32_t v, int /*lane*/)
{
    v += dpp_or_zero<kDppRowShr1, 0xF>(v);
    v += dpp_or_zero<kDppRowShr2, 0xF>(v);
    v += dpp_or_zero<kDppRowShr4, 0xF>(v);
    v += dpp_or_zero<kDppRowShr8, 0xF>(v);
    v += dpp_or_zero<kDppBcast15, 0xA>(v);       /* rows 1 and 3 <- the total of the row in front */
    v += dpp_or_zero<kDppBcast31, 0xC>(v);       /* rows 2 and 3 <- the total of the first half   */
    return v;
}
__device__ __forceinline__ uint32_t wave_sum32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan32(v, 0), 63);
}

struct __attribute__((packed, aligned(1))) Unaligned16_3 { u32x4 v; };
struct __attribute__((packed, aligned(1))) U8_3 { uint64_t v; };
struct __attribute__((packed, aligned(1))) U4_3 { uint32_t v; };
struct __attribute__((packed, aligned(1))) U2_3 { uint16_t v; };

__device__ __forceinline__ void store_pieces(uint8_t* p, uint64_t lo, uint64_t hi, uint32_t cnt)
{
    if (cnt & 8u) { reinterpret_cast<U8_3*>(p)->v = lo; p += 8; lo = hi; }
    if (cnt & 4u) { reinterpret_cast<U4_3*>(p)->v = (uint32_t)lo; p += 4; lo >>= 32; }
    if (cnt & 2u) { reinterpret_cast<U2_3*>(p)->v = (uint16_t)lo; p += 2; lo >>= 16; }
    if (cnt & 1u) { *p = (uint8_t)lo; }
}

/* look-back descriptor accessors (agent scope: the other party is usually on another XCD) */
__device__ __forceinline__ uint64_t ld_desc3(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_desc3(unsigned long long* p, uint64_t v) { __hip_atomic_store(p, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ TileAgg window_fold3(const TileAgg& a, int lstar, int lane)
{
    const uint64_t need = (lstar >= 64) ? ~0ull : ((1ull << lstar) - 1ull);
    const bool mine = lane < lstar;
    const uint64_t m_ev = __ballot(mine && a.last != kKindNone) & need;
    const uint64_t m_st = __ballot(mine && a.last == kKindStart) & need;
    const uint64_t above = (lane >= 63) ? 0ull : (m_ev & ~((2ull << lane) - 1ull));
    uint32_t st = 2u;
    if (above != 0) st = (uint32_t)((m_st >> __builtin_ctzll(above)) & 1ull);
    uint32_t k = 0, g = 0, c = 0;
    if (mine) {
        k = a.known + (st == 1u ? a.sig : 0u);
        g = (st == 2u) ? a.sig : 0u;
        c = a.cnt;
    }
    TileAgg w;
    w.known = wave_sum32(k);
    w.sig = wave_sum32(g);
    w.cnt = wave_sum32(c);
    w.last = kKindNone;
    if (m_ev != 0) w.last = ((m_st >> __builtin_ctzll(m_ev)) & 1ull) ? kKindStart : kKindStop;
    return w;
}

} // namespace hbs
#endif
