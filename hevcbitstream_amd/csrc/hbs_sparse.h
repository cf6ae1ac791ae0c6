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
constexpr int k4Waves         = 4;
constexpr int k4Rows          = 48;
constexpr int k4Threads       = 64 * k4Waves;
constexpr int k4RowBytes      = 1024;
constexpr int k4WaveBytes     = k4Rows * k4RowBytes;         /* 48 KiB  */
constexpr int k4TileBytes     = k4Waves * k4WaveBytes;       /* 192 KiB */
constexpr int k4TileRows      = k4Waves * k4Rows;            /* 192     */
constexpr int k4ChunksPerTile = k4TileBytes / kChunk;        /* 12288   */
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

/* exact classification of one chunk */
HBS_HD void elem_walk(const ElemView& v, ChunkMarks& m, BlockSum& s)
{
    const uint32_t pats = chunk_patterns(v.xp, v.x0, v.x1, v.x2, v.x3, v.xn);
    walk_block_t<kChunk, ElemView, uint32_t>(v, 0, v.g0, v.n, pats & 0xFFFFu, pats >> 16, m, s);
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
