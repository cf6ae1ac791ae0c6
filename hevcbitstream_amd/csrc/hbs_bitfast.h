/*
 * hbs_bitfast.h -- several bits per step for the bit reader of hbs_parse.h.
 *
 * bs.h (and BitIOT's reference-shaped paths) read one bit at a time; a header
 * parse is a chain of ~20 dependent instructions per bit.  When the bits asked
 * for lie wholly inside the RBSP, none of bs.h's end-of-buffer rules can fire
 * and the same value is a shift and a mask over at most five bytes.  These
 * helpers take that case and say so; otherwise they leave the reader untouched
 * and the caller runs the bit-by-bit path.  They live in their own file so that
 * hbs_parse.h keeps its line numbers (HBS_SITE keys the trace names on them).
 *
 * B: a reader with pos (bits consumed), size (bytes) and byte_at(i).
 */
#ifndef HBS_BITFAST_H
#define HBS_BITFAST_H

#include "hbs_common.h"

namespace hbs {

/* The trace sink of hbs_parse.h's BitIOT::log: record n of a NAL's trace (three words: site, cursor, value), when there is room.
 * NOT inlined on the device: the trace variants of the parse kernels carry a read site every other line, and with the stores
 * and their 64-bit address arithmetic inlined at each of them the register allocator gave up (k4_parse<trace>: 3 153 spilled
 * registers, 2.3 KB of scratch per lane; the plain variants of the same walk: none).  A call costs a few dozen cycles per
 * syntax element of a reader that exists to print every one of them. */
#if defined(__HIPCC__)
static __device__ __host__ __attribute__((noinline))
#else
static inline
#endif
void trace_put(uint32_t* records, uint32_t cap, uint32_t n, uint32_t site, uint32_t at, uint32_t value)
{
    if (n < cap) { records[3u * n] = site; records[3u * n + 1u] = at; records[3u * n + 2u] = value; }
}

/* the 40 bits starting at byte i0, big-endian, in the low bits of the result; needs i0 + nbytes <= size */
template <class B>
HBS_HD uint64_t fast_window(const B& b, uint32_t i0, uint32_t nbytes)
{
    uint64_t acc = 0;
    for (uint32_t j = 0; j < 5; ++j) acc = (acc << 8) | (j < nbytes ? (uint64_t)b.byte_at(i0 + j) : 0ull);
    return acc;
}

/* bits(n), bs.h:160-169, for 0 <= n <= 32 with every bit inside the RBSP */
template <class B>
HBS_HD bool fast_bits(B& b, int n, uint32_t& r)
{
    if (n <= 0) { r = 0; return true; }
    if (n > 32 || ((b.pos + (uint32_t)n - 1u) >> 3) >= b.size) return false;
    const uint32_t i0 = b.pos >> 3, sh = b.pos & 7u;
    const uint32_t nbytes = (sh + (uint32_t)n + 7u) >> 3;
    const uint64_t acc = fast_window(b, i0, nbytes);                    /* nbytes real bytes, then 5 - nbytes zero bytes */
    const uint64_t v = acc >> (40u - sh - (uint32_t)n);
    r = (uint32_t)(n == 32 ? v : (v & ((1ull << n) - 1ull)));
    b.pos += (uint32_t)n;
    return true;
}

/* the zero-counting loop of bs_read_ue (bs.h:198-203): `while (bit() == 0 && i < 32 && !eof()) ++i`.
 * With the next 33 bits inside the RBSP and a byte to spare eof() cannot turn true inside it: the
 * loop consumes min(z, 32) + 1 bits and leaves i = min(z, 32), z = zeros in front of the first 1. */
template <class B>
HBS_HD bool fast_zeros(B& b, int& i)
{
    if (((b.pos + 33u) >> 3) >= b.size) return false;
    const uint32_t i0 = b.pos >> 3, sh = b.pos & 7u;
    const uint64_t acc = fast_window(b, i0, 5u);
    const uint32_t next32 = (uint32_t)(acc >> (8u - sh));
    const int z = next32 ? (int)__builtin_clz(next32) : 32;
    i = z;
    b.pos += (uint32_t)z + 1u;
    return true;
}

/* put_bits(n, v), bs.h:240-247, for 0 <= n <= 32 with every bit inside the buffer: up to five
 * read-modify-writes of whole bytes instead of one per bit */
template <class B>
HBS_HD bool fast_put_bits(B& b, int n, uint32_t v)
{
    if (n <= 0) return true;
    if (n > 32 || ((b.pos + (uint32_t)n - 1u) >> 3) >= b.size) return false;
    const uint32_t i0 = b.pos >> 3, sh = b.pos & 7u;
    const uint32_t nbytes = (sh + (uint32_t)n + 7u) >> 3;
    const uint64_t ones = n == 32 ? 0xFFFFFFFFull : ((1ull << n) - 1ull);
    const uint64_t val = ((uint64_t)v & ones) << (40u - sh - (uint32_t)n);
    const uint64_t msk = ones << (40u - sh - (uint32_t)n);
    for (uint32_t j = 0; j < nbytes; ++j) {
        const uint32_t bm = (uint32_t)(msk >> (8u * (4u - j))) & 0xFFu, bv = (uint32_t)(val >> (8u * (4u - j))) & 0xFFu;
        b.wbuf[i0 + j] = (uint8_t)((b.wbuf[i0 + j] & ~bm) | bv);
    }
    b.pos += (uint32_t)n;
    return true;
}

} // namespace hbs
#endif
