/*
 * hbs_emit.h -- per-segment logic of K3 (RBSP -> Annex-B: emulation-prevention
 * insertion + start codes) and of the synthetic-stream generator.
 *
 * Restates rbsp_to_nal (reference h264_nal.c:92-132): walking the RBSP bytes
 * with `count` = zero bytes seen since the last non-zero byte or inserted 03;
 * before a byte <= 3 that arrives with count == 2 a 03 is emitted and count
 * restarts (:110-116); no 03 is appended after a trailing 00 00 (:130).
 * Parallel form: a NAL's RBSP is cut into 256-byte segments; the state a
 * segment starts in only depends on the run of zeros in front of it
 * (z zeros -> count 0, 1 (z odd) or 2 (z even, >= 2)), so segments are
 * independent once their insert counts are prefix-summed.
 * Compiles for gfx950 and, under tests/sim, for the host.
 */
#ifndef HBS_EMIT_H
#define HBS_EMIT_H

#include "hbs_common.h"

namespace hbs {

constexpr uint32_t kSegBytes = 256;       /* only the generator's segmenting of tests/sim still uses this */

/* value of the reference's `count` on entering byte p of a NAL that begins at
 * nal_begin (arena offsets) */
HBS_HD uint32_t lead_count(const uint8_t* rbsp, uint64_t nal_begin, uint64_t p)
{
    uint64_t z = 0;
    while (p - z > nal_begin && rbsp[p - 1 - z] == 0) ++z;
    return z == 0 ? 0u : ((z & 1ull) ? 1u : 2u);
}

/* emulation-prevention bytes rbsp_to_nal inserts while copying [seg_begin, seg_end) */
HBS_HD uint32_t count_segment(const uint8_t* rbsp, uint64_t nal_begin, uint64_t seg_begin, uint64_t seg_end)
{
    uint32_t count = lead_count(rbsp, nal_begin, seg_begin), ins = 0;
    for (uint64_t p = seg_begin; p < seg_end; ++p) {
        const uint32_t v = rbsp[p];
        if (count == 2 && v <= 3) { ++ins; count = 0; }
        count = (v == 0) ? count + 1 : 0;
    }
    return ins;
}

/* ---- the same walk over a 16-byte chunk held in registers (what the kernels run: one 16-byte load per flagged chunk
 * instead of a chain of byte loads) -------------------------------------------------------------------------- */

/* `count` on entering a chunk, from the four bytes in front of it (little-endian dword, nearest byte on top):
 * 0, 1, 2, or kLeadUnknown when all four are zero and the run has to be followed further back (lead_count()) */
constexpr uint32_t kLeadUnknown = 3;
HBS_HD uint32_t lead_count4(uint32_t xp)
{
    if ((xp >> 24) != 0u) return 0u;
    if (((xp >> 16) & 0xFFu) != 0u) return 1u;
    if (((xp >> 8) & 0xFFu) != 0u) return 2u;
    if ((xp & 0xFFu) != 0u) return 1u;
    return kLeadUnknown;
}

/* bytes 0..nb-1 of the chunk w0..w3 (little-endian words) entered with `count`: bit i of the result = a 03 goes in
 * front of byte i (reference h264_nal.c:110-116) */
HBS_HD uint32_t insert_mask16(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t nb, uint32_t count)
{
    uint32_t m = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (uint32_t i = 0; i < 16; ++i) {
        const uint32_t w = i < 4 ? w0 : i < 8 ? w1 : i < 12 ? w2 : w3;
        const uint32_t v = (w >> (8u * (i & 3u))) & 0xFFu;
        const bool ins = count == 2u && v <= 3u && i < nb;
        m |= (ins ? 1u : 0u) << i;
        if (ins) count = 0;
        count = (v == 0u) ? count + 1u : 0u;
    }
    return m;
}

/* writes the chunk's nb bytes with the 03s of `mask`; returns the bytes written */
HBS_HD uint32_t emit_chunk16(uint8_t* dst, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t nb, uint32_t mask)
{
    uint32_t j = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (uint32_t i = 0; i < nb; ++i) {
        if ((mask >> i) & 1u) dst[j++] = 3u;
        dst[j++] = (uint8_t)w0;
        w0 = (w0 >> 8) | (w1 << 24); w1 = (w1 >> 8) | (w2 << 24); w2 = (w2 >> 8) | (w3 << 24); w3 >>= 8;
    }
    return j;
}

/* 16-byte staging register for byte-aligned output */
struct OutBuf {
    uint64_t lo, hi;
    uint32_t n;
    uint8_t* dst;
};

HBS_D void out_store16(uint8_t* dst, uint64_t lo, uint64_t hi)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef uint32_t v4 __attribute__((ext_vector_type(4)));
    struct __attribute__((packed, aligned(1))) U16 { v4 v; };
    v4 v;
    v.x = (uint32_t)lo; v.y = (uint32_t)(lo >> 32); v.z = (uint32_t)hi; v.w = (uint32_t)(hi >> 32);
    reinterpret_cast<U16*>(dst)->v = v;
#else
    for (int i = 0; i < 8; ++i) { dst[i] = (uint8_t)(lo >> (8 * i)); dst[8 + i] = (uint8_t)(hi >> (8 * i)); }
#endif
}

HBS_D void out_push(OutBuf& b, uint32_t v)
{
    const uint64_t x = (uint64_t)v << (8u * (b.n & 7u));
    if (b.n < 8) b.lo |= x; else b.hi |= x;
    if (++b.n == 16) {
        out_store16(b.dst, b.lo, b.hi);
        b.dst += 16; b.lo = 0; b.hi = 0; b.n = 0;
    }
}

HBS_D void out_flush(OutBuf& b)
{
    uint64_t lo = b.lo;
    uint8_t* p = b.dst;
    if (b.n & 8u) { for (int i = 0; i < 8; ++i) p[i] = (uint8_t)(lo >> (8 * i)); p += 8; lo = b.hi; }
    for (uint32_t i = 0; i < (b.n & 7u); ++i) p[i] = (uint8_t)(lo >> (8 * i));
    b.n = 0;
}

/* copy [seg_begin, seg_end) of the arena to dst with emulation prevention */
HBS_D void emit_segment(const uint8_t* rbsp, uint64_t nal_begin, uint64_t seg_begin, uint64_t seg_end, uint8_t* dst)
{
    uint32_t count = lead_count(rbsp, nal_begin, seg_begin);
    OutBuf b;
    b.lo = 0; b.hi = 0; b.n = 0; b.dst = dst;
    for (uint64_t p = seg_begin; p < seg_end; ++p) {
        const uint32_t v = rbsp[p];
        if (count == 2 && v <= 3) { out_push(b, 3u); count = 0; }
        out_push(b, v);
        count = (v == 0) ? count + 1 : 0;
    }
    out_flush(b);
}

/* ---- synthetic stream S(seed, n_nals, mode): SURVEY.md 8(d) ---------------- */

/* ---- chunk algebra of k3_tiles' dense tiles (hbs_emit.hip: k3_dense_tile), host + device so that the CPU tests walk it too ----
 * What a 16-byte chunk does to rbsp_to_nal's state (h264_nal.c:110-116) depends on the count it is entered with -- 0, 1 or 2
 * zeros seen -- only through its leading zeros: an all-zero chunk maps the count 0 -> 2, 1 -> 1, 2 -> 2 and takes 7 / 8 / 8 bytes
 * in; any other chunk leaves a count of its own. */
HBS_HD uint32_t dz_map(uint32_t c) { return c == 1u ? 1u : 2u; }          /* an all-zero chunk: the count behind it */
HBS_HD uint32_t dz_ins(uint32_t c) { return c == 0u ? 7u : 8u; }          /* ... and the 03s that go into it */
HBS_HD uint32_t dz_count_of(uint64_t run) { return run == 0 ? 0u : ((run & 1ull) ? 1u : 2u); }
HBS_HD uint32_t dz_lead_bits(uint32_t c) { return c == 0u ? 0x5554u : (c == 1u ? 0xAAAAu : 0x5555u); }   /* 03s in leading zeros */
/* zero bytes at the top (highest address) end of the chunk w0..w3 (little-endian words), 0..16 */
HBS_HD uint32_t top_zero_bytes4(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3)
{
    if (w3) return (uint32_t)__builtin_clz(w3) >> 3;
    if (w2) return 4u + ((uint32_t)__builtin_clz(w2) >> 3);
    if (w1) return 8u + ((uint32_t)__builtin_clz(w1) >> 3);
    if (w0) return 12u + ((uint32_t)__builtin_clz(w0) >> 3);
    return 16u;
}
/* ... at the low end */
HBS_HD uint32_t low_zero_bytes4(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3)
{
    if (w0) return (uint32_t)__builtin_ctz(w0) >> 3;
    if (w1) return 4u + ((uint32_t)__builtin_ctz(w1) >> 3);
    if (w2) return 8u + ((uint32_t)__builtin_ctz(w2) >> 3);
    if (w3) return 12u + ((uint32_t)__builtin_ctz(w3) >> 3);
    return 16u;
}
HBS_HD uint32_t byte_of4(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t i)
{
    const uint32_t w = i < 4u ? w0 : i < 8u ? w1 : i < 12u ? w2 : w3;
    return (w >> (8u * (i & 3u))) & 0xFFu;
}
/* what a chunk does, for the three counts: the bytes that go in (i0, i1, i2), the count behind it (out; when reset) and whether
 * it sets the count at all (reset = 0: all zeros) */
struct DzFast { uint32_t i0, i1, i2, out, reset; };
HBS_HD DzFast dz_fast4(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3)
{
    DzFast r;
    const uint32_t lz = low_zero_bytes4(w0, w1, w2, w3), tz = top_zero_bytes4(w0, w1, w2, w3);
    const bool allz = lz >= 16u;
    const uint32_t mask0 = insert_mask16(w0, w1, w2, w3, 16u, 0u);
    const uint32_t v = byte_of4(w0, w1, w2, w3, lz & 15u);
    const uint32_t keep = allz ? 0u : (mask0 & ~((2u << lz) - 1u)); /* behind the first byte that is not zero: the same whatever the count was */
    const uint32_t below = allz ? 0xFFFFu : ((1u << lz) - 1u);
    const bool small = !allz && v <= 3u;
    const uint32_t at = allz ? 0u : (1u << lz);
    /* entered with h: the leading zeros take 03s by dz_lead_bits(h); the first other byte takes one when it is <= 3 and the
     * zeros in front of it (h + lz) are two, four, ... */
    const uint32_t f0 = (small && lz != 0u && (lz & 1u) == 0u) ? at : 0u;
    const uint32_t f1 = (small && (lz & 1u) != 0u) ? at : 0u;
    const uint32_t f2 = (small && (lz & 1u) == 0u) ? at : 0u;
    r.i0 = (uint32_t)__builtin_popcount(keep | (dz_lead_bits(0u) & below) | f0);
    r.i1 = (uint32_t)__builtin_popcount(keep | (dz_lead_bits(1u) & below) | f1);
    r.i2 = (uint32_t)__builtin_popcount(keep | (dz_lead_bits(2u) & below) | f2);
    r.out = dz_count_of(tz);
    r.reset = allz ? 0u : 1u;
    return r;
}
/* word `base / 4` of the chunk with its bytes outside [from, to) made 0xFF (from, to in 0..16) */
HBS_HD uint32_t dz_keep_word(uint32_t w, uint32_t base, uint32_t from, uint32_t to)
{
    uint32_t ff = 0u;
    for (uint32_t b = 0; b < 4u; ++b) ff |= ((base + b < from || base + b >= to) ? 0xFFu : 0u) << (8u * b);
    return w | ff;
}
/* a chunk in which ONE NAL begins, at byte s (its start code of `gap` bytes in front): the bytes in front of s belong to the NAL
 * in progress and are entered with the count in question, the bytes from s on to the new NAL, entered with 0 */
HBS_HD DzFast dz_one_start4(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t s, uint32_t gap)
{
    DzFast r = dz_fast4(dz_keep_word(w0, 0u, 0u, s), dz_keep_word(w1, 4u, 0u, s), dz_keep_word(w2, 8u, 0u, s), dz_keep_word(w3, 12u, 0u, s));
    const uint32_t h0 = dz_keep_word(w0, 0u, s, 16u), h1 = dz_keep_word(w1, 4u, s, 16u), h2 = dz_keep_word(w2, 8u, s, 16u), h3 = dz_keep_word(w3, 12u, s, 16u);
    const uint32_t more = gap + (uint32_t)__builtin_popcount(insert_mask16(h0, h1, h2, h3, 16u, 0u));
    r.i0 += more; r.i1 += more; r.i2 += more;
    const uint32_t tz = top_zero_bytes4(h0, h1, h2, h3);
    r.out = dz_count_of(tz < 16u - s ? tz : 16u - s);
    r.reset = 1u;
    return r;
}

HBS_HD uint32_t synth_rbsp_len(uint64_t seed, uint64_t k)
{
    return 8192u + (uint32_t)(mix64(seed ^ ((k + 1) * kGolden)) % 4097u);
}

/* 8 RBSP bytes [8w, 8w+8) of NAL k, little-endian in the result; len = the NAL's RBSP length */
HBS_HD uint64_t synth_rbsp_word(uint64_t seed, uint64_t k, uint32_t w, uint32_t len, int mode)
{
    const uint64_t key = seed ^ ((k + 1) * kGolden);
    uint64_t x = mix64((key ^ kSalt) + (uint64_t)(w + 1) * kGolden);
    if (mode == 1) {
        uint64_t y = 0;
        for (int i = 0; i < 8; ++i) {
            uint32_t b = (uint32_t)(x >> (8 * i)) & 0xFFu;
            if (b < 26) b = 0;
            else if (b < 39) b = 1 + (b - 26) % 3;
            y |= (uint64_t)b << (8 * i);
        }
        x = y;
    }
    if (w == 0) x = (x & ~0xFFFFull) | 0x0102ull;                       /* bytes 0,1 = 02 01 */
    const uint32_t last = len - 1;
    if ((last >> 3) == w) x = (x & ~(0xFFull << (8 * (last & 7)))) | (0x80ull << (8 * (last & 7)));
    return x;
}

HBS_HD uint32_t synth_gap(uint64_t k) { return (k % 4 == 0) ? 4u : 3u; }

} // namespace hbs
#endif
