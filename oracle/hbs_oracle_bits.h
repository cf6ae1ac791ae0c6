/*
 * hbs_oracle_bits.h -- TEST INFRASTRUCTURE ONLY (see hbs_oracle.h).
 * Bit-serial restatement of the read half of the reference's bs.h.
 */
#ifndef HBS_ORACLE_BITS_H
#define HBS_ORACLE_BITS_H

#include <stdint.h>

typedef struct {
    const uint8_t* start;
    const uint8_t* p;
    const uint8_t* end;
    int bits_left;
} obs_t;

/* bs.h:82-89 */
static inline void obs_init(obs_t* b, const uint8_t* buf, long size)
{
    b->start = buf; b->p = buf; b->end = buf + size; b->bits_left = 8;
}
/* bs.h:112-122 */
static inline int obs_aligned(const obs_t* b) { return b->bits_left == 8; }
static inline int obs_eof(const obs_t* b) { return b->p >= b->end; }
static inline int obs_overrun(const obs_t* b) { return b->p > b->end; }
static inline long obs_pos(const obs_t* b) { return (b->p > b->end) ? (long)(b->end - b->start) : (long)(b->p - b->start); }

/* bs.h:126-140: past the end the cursor keeps moving and bits read as 0 */
static inline uint32_t obs_u1(obs_t* b)
{
    uint32_t r = 0;
    b->bits_left--;
    if (!obs_eof(b)) r = ((uint32_t)(*b->p) >> b->bits_left) & 1u;
    if (b->bits_left == 0) { b->p++; b->bits_left = 8; }
    return r;
}
/* bs.h:142-146 */
static inline void obs_skip1(obs_t* b)
{
    b->bits_left--;
    if (b->bits_left == 0) { b->p++; b->bits_left = 8; }
}
/* bs.h:160-169: n may be <= 0 (nothing read) */
static inline uint32_t obs_u(obs_t* b, int n)
{
    uint32_t r = 0;
    int i;
    for (i = 0; i < n; i++) r |= obs_u1(b) << (n - i - 1);
    return r;
}
/* bs.h:171-178 */
static inline void obs_skip(obs_t* b, int n)
{
    int i;
    for (i = 0; i < n; i++) obs_skip1(b);
}
/* bs.h:182-193 (FAST_U8 is on: bs.h:42-48) */
static inline uint32_t obs_u8(obs_t* b)
{
    if (b->bits_left == 8 && !obs_eof(b)) { uint32_t r = b->p[0]; b->p++; return r; }
    return obs_u(b, 8);
}
/* bs.h:195-207: the terminating bit is consumed before the i<32 / eof tests.
 * 1<<32 is undefined in the reference; x86 masks the count, so do we. */
static inline uint32_t obs_ue(obs_t* b)
{
    int i = 0;
    uint32_t r;
    while ((obs_u1(b) == 0) && (i < 32) && (!obs_eof(b))) i++;
    r = obs_u(b, i);
    r += (1u << (i & 31)) - 1u;
    return r;
}
/* bs.h:209-221 */
static inline int32_t obs_se(obs_t* b)
{
    int32_t r = (int32_t)obs_ue(b);
    if (r & 1) r = (r + 1) / 2; else r = -(r / 2);
    return r;
}

#endif
