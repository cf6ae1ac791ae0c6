/*
 * hbs_sparse.h -- event-sparse form of the tile logic (kernel hbs_scan4.hip).
 *
 * In coded video the patterns 00 00 {00..03} that find_nal_unit / nal_to_rbsp
 * (h264_nal.c:38-76, :147-200) react to are rare: a start code per slice and a
 * handful of emulation-prevention bytes per 64 KiB.  Every other byte is kept
 * or dropped purely by the state carried up to it.  So the tile is cut into
 * 16-byte chunks and only the chunks that can be touched by a pattern --
 * "elements" -- get the exact window logic of hbs_tile.h / hbs_chunk.h; the
 * chunks between two elements form a gap whose contribution is just its length.
 *
 *   chunk_flag      9 packed-min ops decide that no two adjacent zero bytes
 *                   start in bytes [-2, 16) of a chunk, i.e. that no pattern ends
 *                   in bytes [0, 18): such a chunk is pure payload;
 *   elem_agg        an element = (gap in front, exact summary of the chunk) as
 *                   one value of the tile algebra (hbs_tile.h combine());
 *   seg_*           what an unflagged chunk needs to place itself: the arena
 *                   bias and the state left by the nearest element in front.
 *
 * Compiles for gfx950 and, under tests/sim, for the host.
 */
#ifndef HBS_SPARSE_H
#define HBS_SPARSE_H

#include "hbs_chunk.h"

namespace hbs {

/* geometry of the event-sparse kernel: a workgroup of k4Waves wavefronts, each holding k4Rows
 * rows of 1 KiB in registers.  Few wavefronts with many registers each: the ~100 registers the
 * control code needs are paid per wavefront, so two fat wavefronts per SIMD keep twice the
 * bytes in flight of four lean ones. */
#ifndef HBS4_ROWS
#define HBS4_ROWS 48              /* the main geometry; hbs_scan4_r24.hip compiles the kernel's source again with 24 (96 KiB tiles) */
#endif
constexpr int k4Waves         = 4;
constexpr int k4Rows          = HBS4_ROWS;
constexpr int k4Threads       = 64 * k4Waves;
constexpr int k4RowBytes      = 1024;
constexpr int k4WaveBytes     = k4Rows * k4RowBytes;         /* 48 KiB  (24 rows: 24 KiB) */
constexpr int k4TileBytes     = k4Waves * k4WaveBytes;       /* 192 KiB (96 KiB)          */
constexpr int k4TileRows      = k4Waves * k4Rows;            /* 192     (96)              */
constexpr int k4ChunksPerTile = k4TileBytes / kChunk;        /* 12288   (6144)            */
constexpr int k4ElemPass      = 64;                          /* elements handled per pass: wavefront 0, one per lane */
constexpr int k4TailLead      = 16;                          /* bytes of the padded last-tile copy in front of the tile */
constexpr int k4TailBytes     = k4TailLead + k4TileBytes + 64;
static_assert(k4TileBytes >= kTileBytes, "the descriptor workspace is sized for kTileBytes tiles; larger tiles need less");
static_assert(k4ChunksPerTile <= 65536, "chunk numbers are kept in 16 bits");

/* per-halfword minimum of two dwords */
HBS_HD uint32_t pk_min_u16(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    const u16x2 r = __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b));
    return __builtin_bit_cast(uint32_t, r);
#else
    const uint32_t lo = ((a & 0xFFFFu) < (b & 0xFFFFu)) ? (a & 0xFFFFu) : (b & 0xFFFFu);
    const uint32_t hi = ((a >> 16) < (b >> 16)) ? (a >> 16) : (b >> 16);
    return lo | (hi << 16);
#endif
}

/*
 * May a pattern 00 00 v end in bytes [0, 18) of the chunk x0..x3 (xp = dword in
 * front, xn = dword behind)?  True iff two adjacent zero bytes start at some
 * p in [-2, 16) -- or in the harmless extra position 17.  Every byte pair is a
 * halfword of the data itself (even p) or of the data shifted by one byte
 * (odd p), so a running packed minimum that reaches 0 is the whole test.
 */
HBS_HD bool chunk_flag(uint32_t xp, uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t xn)
{
    uint32_t m = pk_min_u16(xp | 0x0000FFFFu, x0);                /* p = -2, 0, 2 */
    m = pk_min_u16(m, x1);
    m = pk_min_u16(m, x2);
    m = pk_min_u16(m, x3);                                         /* ... 12, 14   */
    m = pk_min_u16(m, alignbyte(x0, xp, 3));                       /* p = -1, 1    */
    m = pk_min_u16(m, alignbyte(x1, x0, 3));                       /* 3, 5         */
    m = pk_min_u16(m, alignbyte(x2, x1, 3));                       /* 7, 9         */
    m = pk_min_u16(m, alignbyte(x3, x2, 3));                       /* 11, 13       */
    m = pk_min_u16(m, alignbyte(xn, x3, 3));                       /* 15, (17)     */
    return (m & 0xFFFFu) == 0u || m < 0x10000u;
}

/* stream bytes in [from, to) that exist */
HBS_HD uint32_t span_bytes(uint64_t from, uint64_t to, uint64_t n)
{
    const uint64_t e = to < n ? to : n;
    return e > from ? (uint32_t)(e - from) : 0u;
}

HBS_HD TileAgg agg_identity() { TileAgg a; a.cnt = 0; a.known = 0; a.sig = 0; a.last = kKindNone; return a; }

/* a run of chunks without any pattern: `bytes` kept iff the state carried into it is "inside" */
HBS_HD TileAgg gap_agg(uint32_t bytes) { TileAgg a; a.cnt = 0; a.known = 0; a.sig = bytes; a.last = kKindNone; return a; }

/* element = gap of gap_bytes in front of a chunk with summary s */
HBS_HD TileAgg elem_agg(uint32_t gap_bytes, const BlockSum& s)
{
    TileAgg a;
    a.cnt = s.cnt; a.known = s.known; a.sig = gap_bytes + s.carry; a.last = s.last;
    return a;
}

/* An element's bytes [-8, 20) around its chunk, in registers: everything the window rules and
 * the index emission ever look at (pattern_kind reads back to o-5, emit_block_t to e-5), so an
 * element costs one round of loads and no dependent ones. */
struct ElemView {
    uint32_t xpp, xp, x0, x1, x2, x3, xn;
    const uint8_t* stream;
    uint64_t g0;          /* stream offset of the chunk's first byte */
    uint64_t n;

    HBS_M uint32_t byte(int32_t o) const
    {
        if (o >= -8 && o < 20) {
            const uint32_t k = (uint32_t)(o + 8) >> 2;
            const uint32_t d = (k == 0) ? xpp : (k == 1) ? xp : (k == 2) ? x0 : (k == 3) ? x1 : (k == 4) ? x2 : (k == 5) ? x3 : xn;
            return (d >> (8u * ((uint32_t)(o + 8) & 3u))) & 0xFFu;
        }
        const int64_t q = (int64_t)g0 + o;
        return (q >= 0 && (uint64_t)q < n) ? stream[q] : 0xFFu;
    }
};

typedef BlockMarksT<uint32_t> ChunkMarks;      /* 16 byte positions (+2 behind) fit 32 bits */

/* exact classification of one chunk: the window rules of hbs_tile.h, one pattern at a time */
HBS_HD void elem_walk_generic(const ElemView& v, ChunkMarks& m, BlockSum& s)
{
    const uint32_t pats = chunk_patterns(v.xp, v.x0, v.x1, v.x2, v.x3, v.xn);
    walk_block_t<kChunk, ElemView, uint32_t>(v, 0, v.g0, v.n, pats & 0xFFFFu, pats >> 16, m, s);
}

/*
 * The same rules with every position of the element at once.  An element's 28 bytes [-8, 20) are
 * classified into three bit masks (bit i = position i - 8): byte <= 3, bit 0 of the byte, bit 1 of
 * the byte -- from which "is 00 / 01 / 02 / 03" are three ANDs -- and the rules of pattern_kind() /
 * walk_block_t() become shifts and ANDs of those masks: no loop over patterns, no byte fetches, no
 * divergence between the lanes of the wavefront that walks a tile's elements (the walk is serial work
 * that a whole workgroup waits for).
 */
struct ElemClasses {
    uint32_t z, e1, e3;       /* positions holding 00 / 01 / 03: what the emit half looks at */
};
/* four flags at bits 0, 8, 16, 24 -> bits 0..3 */
HBS_HD uint32_t gather4(uint32_t t) { return (t | (t >> 7) | (t >> 14) | (t >> 21)) & 0xFu; }

HBS_HD void elem_class_masks(const ElemView& v, uint32_t& le3, uint32_t& b0, uint32_t& b1)
{
    const uint32_t d[7] = {v.xpp, v.xp, v.x0, v.x1, v.x2, v.x3, v.xn};
    le3 = 0; b0 = 0; b1 = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 7; ++k) {
        const uint32_t x = d[k];
        le3 |= gather4(zero_bytes(x & 0xFCFCFCFCu) >> 7) << (4 * k);
        b0 |= gather4(x & 0x01010101u) << (4 * k);
        b1 |= gather4((x >> 1) & 0x01010101u) << (4 * k);
    }
}

/* bits [0, k) of a position mask; k may be negative or beyond the word */
HBS_HD uint32_t below_pos(int64_t k) { return k <= 0 ? 0u : (k >= 32 ? 0xFFFFFFFFu : ((1u << (uint32_t)k) - 1u)); }

/* Exact classification of one chunk, every position at once.  The end-of-stream clauses of pattern_kind()
 * (h264_nal.c:52, :71 and the unchecked first end candidate :65-66) are position limits: with L = n - g0 bytes
 * of stream from the chunk's first byte on, a terminator ending at j counts if j + 2 <= L or if it sits right
 * behind a start code; a 00 00 01 ending at j starts a NAL if j + 3 <= L; the byte behind an emulation prevention
 * byte at j exists if j + 1 < L.  Bytes past the end of the stream read as FF and belong to no class. */
HBS_HD void elem_walk(const ElemView& v, ChunkMarks& m, BlockSum& s, ElemClasses& cls)
{
    uint32_t le3, b0, b1;
    elem_class_masks(v, le3, b0, b1);
    const uint32_t z = le3 & ~b0 & ~b1, e1 = le3 & b0 & ~b1, e2 = le3 & ~b0 & b1, e3 = le3 & b0 & b1;
    cls.z = z; cls.e1 = e1; cls.e3 = e3;
    const int64_t L = v.g0 < v.n ? (v.n - v.g0 > 64 ? 64 : (int64_t)(v.n - v.g0)) : 0;
    const uint32_t pat = ((z << 2) & (z << 1) & le3) >> 8;            /* 00 00 {00..03} ends at position 0..19      */
    const uint32_t zp = z >> 8, e1p = e1 >> 8, e2p = e2 >> 8, e3p = e3 >> 8;
    const uint32_t termc = pat & ((zp & ~(z >> 5)) | e1p);            /* 00 00 01, or the first 00 00 00 of a run    */
    const uint32_t tok = below_pos(L - 1) | ((z >> 3) & (z >> 4) & (e1 >> 5));   /* complete, or right behind a start code */
    const uint32_t term = termc & tok & 0x3FFFFu;                     /* positions 0..17                             */
    const uint32_t start = term & e1p & below_pos(L - 2);
    const uint32_t epb = pat & e3p & 0xFFFFu;
    m.ev = term & 0xFFFFu;
    m.ev_start = start & 0xFFFFu;
    m.err = ((pat & e2p) | (termc & ~tok) | (pat & e3p & (~le3 >> 9) & below_pos(L - 1))) & 0xFFFFu;
    m.cand = below_pos(L < 16 ? L : 16) & ~epb & ~(term | (term >> 1) | (term >> 2));   /* a terminator's bytes are outside every NAL */
    /* summary: bytes before the first event follow the carried state; bytes behind a start code are inside */
    const uint32_t x = ~m.ev & 0xFFFFu;
    const uint32_t run = (((m.ev_start << 1) + x) ^ x) & x;          /* positions from behind each start code up to the next event */
    s.cnt = (uint32_t)__builtin_popcount(m.ev_start);
    s.known = (uint32_t)__builtin_popcount(m.cand & run);
    if (m.ev != 0u) {
        s.carry = (uint32_t)__builtin_popcount(m.cand & ((1u << __builtin_ctz(m.ev)) - 1u));
        s.last = ((m.ev_start >> (31 - __builtin_clz(m.ev))) & 1u) ? kKindStart : kKindStop;
    } else {
        s.carry = (uint32_t)__builtin_popcount(m.cand);
        s.last = kKindNone;
    }
}

/*
 * Second half for one element (emit_block_t of hbs_tile.h with the byte fetches replaced by tests of the
 * class masks): index entries of the events, status flags, the final keep mask.  Valid for every element:
 * the bytes it asks about, [-5, 13) of the chunk, are among the 28 the masks describe.
 */
HBS_D uint32_t emit_chunk_fast(const ElemClasses& cls, uint64_t g0, const ChunkMarks& m, bool inside,
                               uint64_t nal_ord, uint64_t rbsp_pos, const EmitTarget& tgt)
{
    uint32_t inside_mask = 0, cur = 0;
    uint64_t ord = nal_ord;
    for (uint32_t r = m.ev; r != 0; r &= r - 1) {
        const uint32_t e = (uint32_t)__builtin_ctz(r);
        if (inside) {
            inside_mask |= ((1u << e) - 1u) & ~((1u << cur) - 1u);
            const uint64_t k = ord - 1;
            /* the terminator begins at e - 2: bytes e-5, e-4, e-3 in front of it <-> mask bits e+3, e+4, e+5 */
            const bool zz = ((cls.z >> (e + 3)) & (cls.z >> (e + 4)) & 1u) != 0u;
            if (k < tgt.index_cap) {
                tgt.index[k].end = g0 + e - 2;                        /* h264_nal.c:74 */
                if (zz && ((cls.e3 >> (e + 5)) & 1u)) atomic_or_status(&tgt.index[k], HBS_ST_TRAILING03);   /* h264_nal.c:170-173 */
            }
            if (zz && ((cls.e1 >> (e + 5)) & 1u)) atomic_min_u64(&tgt.hdr->first_empty, k);   /* empty NAL: the loop of hevc_analyze.c:135 stops */
        }
        if ((m.ev_start >> e) & 1u) {
            const uint64_t k = ord++;
            if (k < tgt.index_cap) {
                tgt.index[k].start = g0 + e + 1;                      /* h264_nal.c:61-62 */
                tgt.index[k].rbsp_off = rbsp_pos + (uint32_t)__builtin_popcount(m.cand & inside_mask);
            } else {
                flag_error(tgt.hdr, (uint32_t)(-HBS_E_CAPACITY));
            }
            inside = true;
        } else {
            inside = false;
        }
        cur = e + 1;
    }
    if (inside) inside_mask |= 0xFFFFu & ~((1u << cur) - 1u);
    for (uint32_t r = m.err & inside_mask; r != 0; r &= r - 1) {
        const uint32_t pos = (uint32_t)__builtin_ctz(r);
        const uint64_t k = nal_ord + (uint32_t)__builtin_popcount(m.ev_start & ((1u << pos) - 1u)) - 1;
        if (k < tgt.index_cap) atomic_or_status(&tgt.index[k], HBS_ST_ERROR);
    }
    return m.cand & inside_mask;
}

/* an element's marks and summary in three dwords (chunk-sized masks are 16 bits each) */
struct ElemPacked { uint32_t a, b, c; };
HBS_HD ElemPacked elem_pack(const ChunkMarks& m, const BlockSum& s)
{
    ElemPacked p;
    p.a = (m.cand & 0xFFFFu) | (m.ev << 16);
    p.b = (m.ev_start & 0xFFFFu) | (m.err << 16);
    p.c = s.cnt | (s.known << 8) | (s.carry << 16) | (s.last << 24);
    return p;
}
HBS_HD void elem_unpack(const ElemPacked& p, ChunkMarks& m, BlockSum& s)
{
    m.cand = p.a & 0xFFFFu; m.ev = p.a >> 16; m.ev_start = p.b & 0xFFFFu; m.err = p.b >> 16;
    s.cnt = p.c & 0xFFu; s.known = (p.c >> 8) & 0xFFu; s.carry = (p.c >> 16) & 0xFFu; s.last = p.c >> 24;
}

/* Where an element stands once the tile's carried state is known: e = aggregate of
 * everything in the tile in front of the element's gap. */
struct ElemStart {
    bool inside;           /* state at the element's chunk (a gap never changes it) */
    uint32_t kept;         /* kept bytes of the tile in front of the chunk          */
};
HBS_HD ElemStart elem_start(const TileAgg& e, uint32_t gap_bytes, uint32_t tile_inside)
{
    ElemStart r;
    r.inside = (e.last != kKindNone) ? (e.last == kKindStart) : (tile_inside != 0u);
    r.kept = e.known + (tile_inside ? e.sig : 0u) + (r.inside ? gap_bytes : 0u);
    return r;
}

/*
 * Segment word left by an element for the unflagged chunks behind it: chunk c of
 * the tile (not an element) is kept iff seg_inside(), and then its 16 bytes go
 * to tile rank seg_bias() + 16 c.  chunk = the element's chunk number, kept_after
 * = kept bytes of the tile up to and including the element's chunk.
 */
HBS_HD uint32_t seg_pack(int32_t chunk, uint32_t kept_after, bool inside_after)
{
    const int32_t bias = (int32_t)kept_after - 16 * (chunk + 1);
    return ((uint32_t)bias << 1) | (inside_after ? 1u : 0u);
}
HBS_HD bool seg_inside(uint32_t w) { return (w & 1u) != 0u; }
HBS_HD int32_t seg_bias(uint32_t w) { return (int32_t)w >> 1; }

} // namespace hbs
#endif
