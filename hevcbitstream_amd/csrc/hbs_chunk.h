/*
 * hbs_chunk.h -- 16-byte-chunk form of the tile logic, the base of the event-sparse
 * kernels (hbs_sparse.h builds on it): a lane holds one 16-byte chunk of the
 * stream in four VGPRs plus the dword in front (xp) and behind (xn), and
 * everything hbs_tile.h does per 64-byte block of an LDS image happens here
 * per chunk straight from registers: same window rules, same walk_block /
 * emit_block code (instantiated with B = 16 over a RegView).
 * Compiles for gfx950 and, under tests/sim, for the host.
 */
#ifndef HBS_CHUNK_H
#define HBS_CHUNK_H

#include "hbs_tile.h"

namespace hbs {

constexpr int kChunk = 16;

/* bytes [-4, 20) around a chunk come from registers, anything else from the stream */
struct RegView {
    uint32_t xp, x0, x1, x2, x3, xn;
    const uint8_t* stream;
    uint64_t g0;          /* stream offset of the chunk's first byte */
    uint64_t n;

    HBS_M uint32_t byte(int32_t o) const
    {
        if (o >= -4 && o < 20) {
            const uint32_t k = (uint32_t)(o + 4) >> 2;
            uint32_t d = (k == 0) ? xp : (k == 1) ? x0 : (k == 2) ? x1 : (k == 3) ? x2 : (k == 4) ? x3 : xn;
            return (d >> (8u * ((uint32_t)(o + 4) & 3u))) & 0xFFu;
        }
        const int64_t q = (int64_t)g0 + o;
        return (q >= 0 && (uint64_t)q < n) ? stream[q] : 0xFFu;
    }
};

/* conservative filter: may any pattern 00 00 {<=3} end at bytes [0, 18) of this chunk? */
HBS_HD bool chunk_maybe_pattern(uint32_t xp, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t xn)
{
    const uint32_t ap = zero_bytes_approx(xp), a0 = zero_bytes_approx(x0), a1 = zero_bytes_approx(x1);
    const uint32_t a2 = zero_bytes_approx(x2), a3 = zero_bytes_approx(x3), an = zero_bytes_approx(xn);
    /* adjacent zero pairs ending at bytes -1 .. 16 (a pattern ending at j needs the pair ending at j-1) */
    const uint32_t lead = ap & (ap << 8) & 0x80000000u;                       /* pair (-2,-1) */
    const uint32_t mid = (a0 & alignbyte(a0, ap, 3)) | (a1 & alignbyte(a1, a0, 3)) |
                         (a2 & alignbyte(a2, a1, 3)) | (a3 & alignbyte(a3, a2, 3));   /* pairs ending at 0..15 */
    const uint32_t tail = an & alignbyte(an, a3, 3) & 0x80u;                   /* pair (15,16) */
    return (lead | mid | tail) != 0;
}

/* exact pattern-end masks: bits 0..15 = patterns ending inside the chunk, bits 16,17 = in the
 * first two bytes behind it */
HBS_HD uint32_t chunk_patterns(uint32_t xp, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t xn)
{
    uint32_t z = zero_bytes(xp);
    const uint32_t p0 = pattern_marks(x0, z), p1 = pattern_marks(x1, z), p2 = pattern_marks(x2, z), p3 = pattern_marks(x3, z);
    const uint32_t pn = pattern_marks(xn, z);
    return movemask4(p0) | (movemask4(p1) << 4) | (movemask4(p2) << 8) | (movemask4(p3) << 12) | ((movemask4(pn) & 3u) << 16);
}

/* chunk_patterns() != 0 without forming the mask: does a pattern 00 00 {<=3} end in bytes [0, 18) of the chunk? */
HBS_HD bool chunk_pattern_any(uint32_t xp, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t xn)
{
    uint32_t z = zero_bytes(xp);
    const uint32_t p0 = pattern_marks(x0, z), p1 = pattern_marks(x1, z), p2 = pattern_marks(x2, z), p3 = pattern_marks(x3, z);
    const uint32_t pn = pattern_marks(xn, z);
    return (p0 | p1 | p2 | p3 | (pn & 0x00008080u)) != 0u;
}

/* Kept bytes of a chunk with holes, packed low-to-high into lo/hi, straight from registers (bytes past the count: undefined).
 * By RUNS of dropped bytes, from the top: a run [a, t] goes by moving everything above it down over it.  A chunk with one
 * emulation prevention byte, or the end of a NAL, is one run -- ~25 instructions where a loop over the kept bytes (round 2) took
 * 15 turns of a dozen, and a batch of elements takes as long as its slowest lane. */
HBS_HD uint32_t compact_chunk_regs(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t keep16, uint64_t& lo, uint64_t& hi)
{
    uint64_t l = ((uint64_t)x1 << 32) | x0, h = ((uint64_t)x3 << 32) | x2;
    keep16 &= 0xFFFFu;
    uint32_t drop = ~keep16 & 0xFFFFu;
    while (drop != 0) {
        const uint32_t t = 31u - (uint32_t)__builtin_clz(drop);                 /* the highest dropped byte ...              */
        const uint32_t below = ~drop & ((1u << t) - 1u);
        const uint32_t a = below ? 32u - (uint32_t)__builtin_clz(below) : 0u;   /* ... and where its run begins              */
        const uint32_t s = 8u * (t - a + 1u);                                    /* bits to move down: 8 .. 128               */
        uint64_t sl, sh;
        if (s >= 64u) { sl = (s >= 128u) ? 0ull : (h >> (s - 64u)); sh = 0ull; }
        else { sl = (l >> s) | (h << (64u - s)); sh = h >> s; }
        uint64_t ml, mh;                                                         /* bytes below a stay                        */
        if (a >= 8u) { ml = ~0ull; mh = (1ull << (8u * (a - 8u))) - 1ull; }
        else { ml = (1ull << (8u * a)) - 1ull; mh = 0ull; }
        l = (l & ml) | (sl & ~ml);
        h = (h & mh) | (sh & ~mh);
        drop &= (1u << a) - 1u;
    }
    lo = l; hi = h;
    return (uint32_t)__builtin_popcount(keep16);
}

} // namespace hbs
#endif
