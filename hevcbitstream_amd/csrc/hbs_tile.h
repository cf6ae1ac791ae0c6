/*
 * hbs_tile.h -- per-tile logic of the fused scan / index / RBSP-extract kernel.
 *
 * A tile is 64 KiB of the stream staged in LDS (swizzled image, TileView);
 * each of the 512 threads owns 128 contiguous bytes = two 64-byte "blocks",
 * the unit of classification (one 64-bit mask per property).  Everything here
 * is per-thread code over the LDS image plus a few words of carried state, so
 * it compiles for gfx950 and, under tests/sim, for the host (hbs_common.h).
 *
 * What is restated, and from where (reference = leslie-wang/hevcbitstream):
 *   - start/end of a NAL: find_nal_unit, h264_nal.c:46-53 (start search for
 *     00 00 01 / 00 00 00 01), :64-72 (end search for 00 00 00 / 00 00 01) and
 *     their end-of-buffer rules `i+4 >= size` (:52), `i+3 >= size` (:71);
 *   - which bytes survive: nal_to_rbsp, h264_nal.c:156-177 (00 00 03 dropped,
 *     00 00 {00,01,02} and 00 00 03 {>03} rejected, trailing 03 dropped).
 * The sequential counters of the reference become local 3-byte window rules
 * (SURVEY.md App. A/B); the only state that crosses a block boundary is
 * "inside a NAL payload or not", carried by the scan.
 *
 * Event positions are the LAST byte j of a pattern 00 00 v (v <= 3):
 *   v == 3            emulation-prevention byte (dropped when inside a NAL)
 *   v == 2            error when inside a NAL
 *   v == 0 / v == 1   terminator: the NAL being read (if any) ends at j-2;
 *                     v == 1 also starts a new NAL at j+1.
 */
#ifndef HBS_TILE_H
#define HBS_TILE_H

#include "hbs_common.h"

namespace hbs {

struct TileGeom {
    uint64_t tile_base;   /* stream offset of the tile's first byte */
    uint64_t n;           /* stream length in bytes                 */
};

/* per-thread registers that live across the look-back wait; Mask = one bit per byte of the
 * block (uint64_t for 64-byte blocks, uint32_t is enough for 16-byte chunks) */
template <class Mask>
struct BlockMarksT {
    Mask cand;            /* bytes that are RBSP if their position is inside a NAL */
    Mask ev;              /* terminator events (position of the pattern's last byte) */
    Mask ev_start;        /* subset of ev that also starts a NAL                    */
    Mask err;             /* positions that make nal_to_rbsp fail if inside a NAL   */
};
typedef BlockMarksT<uint64_t> BlockMarks;

/* bit helpers for either mask width */
HBS_HD uint32_t mask_popc(uint64_t v) { return (uint32_t)__builtin_popcountll(v); }
HBS_HD uint32_t mask_popc(uint32_t v) { return (uint32_t)__builtin_popcount(v); }
HBS_HD uint32_t mask_ctz(uint64_t v)  { return (uint32_t)__builtin_ctzll(v); }
HBS_HD uint32_t mask_ctz(uint32_t v)  { return (uint32_t)__builtin_ctz(v); }
template <class T> struct type_id { typedef T type; };       /* keeps an argument out of template deduction */
/* bits [0, n) set; n in [0, bits of Mask] */
template <class Mask> HBS_HD Mask mask_below(uint32_t n)
{
    return n >= 8u * (uint32_t)sizeof(Mask) ? (Mask)~(Mask)0 : (Mask)(((Mask)1 << n) - (Mask)1);
}

/* what one block contributes to the scan */
struct BlockSum {
    uint32_t cnt;         /* NAL starts                                             */
    uint32_t known;       /* kept bytes whose inside/outside state is decided here  */
    uint32_t carry;       /* kept bytes before the first event (state = carry-in)   */
    uint32_t last;        /* kKindNone / kKindStart / kKindStop: state after block  */
};

/* aggregate of a whole tile and the running prefix it is folded into */
struct TileAgg { uint32_t cnt, known, sig, last; };
struct Prefix  { uint64_t kept, nals; uint32_t inside; };

HBS_HD Prefix fold(Prefix p, TileAgg a)
{
    p.kept += a.known + (p.inside ? a.sig : 0u);
    p.nals += a.cnt;
    if (a.last != kKindNone) p.inside = (a.last == kKindStart) ? 1u : 0u;
    return p;
}

/* look-back descriptor words (two self-validating 8-byte granules per tile) */
enum : uint64_t { kDescEmpty = 0, kDescAgg = 1, kDescPrefix = 2 };
HBS_HD uint64_t pack_agg0(TileAgg a)   { return kDescAgg | ((uint64_t)a.last << 2) | ((uint64_t)a.known << 4) | ((uint64_t)a.sig << 24); }
HBS_HD uint64_t pack_agg1(TileAgg a)   { return kDescAgg | ((uint64_t)a.cnt << 2); }
HBS_HD uint64_t pack_pre0(Prefix p)    { return kDescPrefix | ((uint64_t)p.inside << 2) | (p.kept << 3); }
HBS_HD uint64_t pack_pre1(Prefix p)    { return kDescPrefix | (p.nals << 2); }
/* aggregate of [earlier tiles a] followed by [later tiles b] (associative) */
HBS_HD TileAgg combine(TileAgg a, TileAgg b)
{
    TileAgg c;
    c.cnt = a.cnt + b.cnt;
    c.known = a.known + b.known + (a.last == kKindStart ? b.sig : 0u);
    c.sig = a.sig + (a.last == kKindNone ? b.sig : 0u);
    c.last = (b.last != kKindNone) ? b.last : a.last;
    return c;
}
HBS_HD TileAgg unpack_agg(uint64_t w0, uint64_t w1)
{
    TileAgg a;
    a.last = (uint32_t)(w0 >> 2) & 3u;
    a.known = (uint32_t)(w0 >> 4) & 0xFFFFFu;
    a.sig = (uint32_t)(w0 >> 24) & 0xFFFFFu;
    a.cnt = (uint32_t)(w1 >> 2);
    return a;
}
HBS_HD Prefix unpack_pre(uint64_t w0, uint64_t w1)
{
    Prefix p;
    p.inside = (uint32_t)(w0 >> 2) & 1u;
    p.kept = w0 >> 3;
    p.nals = w1 >> 2;
    return p;
}

/* pattern kinds */
enum : int { kPatEpb = 0, kPatErr = 1, kPatStart = 2, kPatStop = 3, kPatSkip = 4 };

/*
 * Kind of the pattern 00 00 v whose last byte is at LDS address `at` / stream
 * offset gj.  End-of-stream rules of find_nal_unit:
 *   a terminator that begins at p = gj-2 counts only if p <= n-4 (h264_nal.c:71:
 *   `i+3 >= size` ends the search), or if it sits right behind a start code (the
 *   first end candidate is never bounds-checked, :65-66) -- the empty-NAL case;
 *   a start code found while advancing needs i+4 < size (:52): gj <= n-3 always
 *   satisfies it; the two later positions are settled by tail_fixup().
 */
template <class View>
HBS_HD int pattern_kind(const View& v, int32_t o, uint64_t gj, uint64_t n)
{
    const uint32_t b = v.byte(o);
    if (b == 3) return kPatEpb;
    if (b == 2) return kPatErr;
    if (b == 0 && v.byte(o - 3) == 0) return kPatSkip;   /* not the first 00 00 00 of a zero run */
    bool term_ok = gj + 2 <= n;
    if (!term_ok) term_ok = (v.byte(o - 5) == 0 && v.byte(o - 4) == 0 && v.byte(o - 3) == 1);
    if (!term_ok) return kPatErr;
    return (b == 1 && gj + 3 <= n) ? kPatStart : kPatStop;
}

/* 0x80 marks of the bytes of x that end a pattern 00 00 {00..03}; zprev = zero
 * marks of the previous dword (in/out) */
HBS_HD uint32_t pattern_marks(uint32_t x, uint32_t& zprev)
{
    const uint32_t z = zero_bytes(x);
    const uint32_t lo = zero_bytes(x & 0xFCFCFCFCu);          /* bytes <= 3        */
    const uint32_t z1 = alignbyte(z, zprev, 3);               /* byte j-1 is zero  */
    const uint32_t z2 = alignbyte(z, zprev, 2);               /* byte j-2 is zero  */
    zprev = z;
    return z1 & z2 & lo;
}

/*
 * Cheap, conservative zero-byte marks: 0x80 in the lowest zero byte of x for
 * sure, possibly also in 0x01 bytes above a zero byte (borrow).  Good enough
 * to decide that a block holds no two adjacent zero bytes at all.
 */
HBS_HD uint32_t zero_bytes_approx(uint32_t x)
{
    return (x - 0x01010101u) & ~x & 0x80808080u;
}

/* exact pattern-end mask of the 64-byte block at logical offset o; zprev = exact
 * zero marks of the dword in front of it */
HBS_HD uint64_t block_patterns_exact(const TileView& v, int32_t o, uint32_t zprev)
{
    uint64_t pat = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int q = 0; q < 4; ++q) {
        const Quad d = v.quad(o + 16 * q);
        const uint32_t p0 = pattern_marks(d.x, zprev);
        const uint32_t p1 = pattern_marks(d.y, zprev);
        const uint32_t p2 = pattern_marks(d.z, zprev);
        const uint32_t p3 = pattern_marks(d.w, zprev);
        const uint32_t m16 = movemask4(p0) | (movemask4(p1) << 4) | (movemask4(p2) << 8) | (movemask4(p3) << 12);
        pat |= (uint64_t)m16 << (16 * q);
    }
    return pat;
}

/*
 * Pattern-end mask of the 64-byte block at logical offset o.  First a cheap
 * pass looks for any pair of adjacent zero bytes that could sit in front of a
 * pattern byte of this block (approximate marks, 5 ops per dword); only blocks
 * that have one -- about 1 in 1000 for entropy-coded payload -- run the exact
 * SWAR pass.  aprev = approximate marks of the dword in front of the block
 * (in/out: on return, of the block's last dword).
 */
HBS_HD uint64_t block_patterns(const TileView& v, int32_t o, uint32_t& aprev)
{
    uint32_t any = 0;
    uint32_t prev = aprev;
    /* a pair that ENDS in the byte before the block (bytes -2,-1) also matters */
    const uint32_t lead = prev & (prev << 8) & 0x80000000u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int q = 0; q < 4; ++q) {
        const Quad d = v.quad(o + 16 * q);
        const uint32_t a0 = zero_bytes_approx(d.x), a1 = zero_bytes_approx(d.y);
        const uint32_t a2 = zero_bytes_approx(d.z), a3 = zero_bytes_approx(d.w);
        any |= (a0 & alignbyte(a0, prev, 3)) | (a1 & alignbyte(a1, a0, 3)) | (a2 & alignbyte(a2, a1, 3)) | (a3 & alignbyte(a3, a2, 3));
        prev = a3;
    }
    aprev = prev;
    /* pairs ending at byte 63 only feed the next block, which sees them as `lead` */
    if ((any | lead) == 0) return 0;
    return block_patterns_exact(v, o, zero_bytes(v.dword(o - 4)));
}

/*
 * Turn the pattern mask of one block into its marks and scan summary.
 * o = logical offset of the block in the tile image, g0 = its stream offset,
 * pat_next = patterns ending in the first two bytes of the next block.
 */
template <int B, class View, class Mask = uint64_t>
HBS_HD void walk_block_t(const View& v, int32_t o, uint64_t g0, uint64_t n,
                         typename type_id<Mask>::type pat, uint32_t pat_next, BlockMarksT<Mask>& m, BlockSum& s)
{
    static_assert(B <= (int)(8 * sizeof(Mask)), "mask too narrow for the block");
    const Mask one = 1;
    const uint32_t nvalid = (g0 >= n) ? 0u : (n - g0 >= (uint64_t)B ? (uint32_t)B : (uint32_t)(n - g0));
    m.cand = mask_below<Mask>(nvalid);
    m.ev = m.ev_start = m.err = 0;

    for (Mask r = pat; r != 0; r &= r - 1) {
        const uint32_t j = mask_ctz(r);
        const Mask bit = one << j;
        const int kind = pattern_kind(v, o + (int32_t)j, g0 + j, n);
        if (kind == kPatEpb) {
            m.cand &= ~bit;
            /* 00 00 03 followed by a byte > 3 that is still part of the stream
             * (h264_nal.c:164: i < nal_size-1 && next > 3) */
            if (v.byte(o + (int32_t)j + 1) > 3 && g0 + j + 1 < n) m.err |= bit;
        } else if (kind == kPatErr) {
            m.err |= bit;
        } else if (kind == kPatStart || kind == kPatStop) {
            m.ev |= bit;
            if (kind == kPatStart) m.ev_start |= bit;
            /* the three pattern bytes are outside every NAL */
            m.cand &= ~(bit | (bit >> 1) | (bit >> 2));
        }
    }
    /* terminators that end in the next block exclude our last bytes */
    for (uint32_t b = 0; b < 2; ++b) {
        if (pat_next & (1u << b)) {
            const uint32_t j = (uint32_t)B + b;
            const int kind = pattern_kind(v, o + (int32_t)j, g0 + j, n);
            if (kind == kPatStart || kind == kPatStop)
                m.cand &= ~((b == 0) ? ((Mask)3 << (B - 2)) : (one << (B - 1)));
        }
    }

    /* summary: bytes before the first event follow the carried state */
    uint32_t carry = mask_popc(m.cand), known = 0, last = kKindNone;
    if (m.ev != 0) {
        uint32_t cur = mask_ctz(m.ev);
        carry = mask_popc((Mask)(m.cand & mask_below<Mask>(cur)));
        bool inside = false;
        for (Mask r = m.ev; r != 0; r &= r - 1) {
            const uint32_t e = mask_ctz(r);
            if (inside) known += mask_popc((Mask)(m.cand & mask_below<Mask>(e) & ~mask_below<Mask>(cur)));
            inside = (m.ev_start >> e) & one;
            cur = e + 1;
        }
        if (inside) known += mask_popc((Mask)(m.cand & ~mask_below<Mask>(cur)));
        last = inside ? kKindStart : kKindStop;
    }
    s.cnt = mask_popc(m.ev_start);
    s.known = known;
    s.carry = carry;
    s.last = last;
}

HBS_HD void walk_block(const TileView& v, int32_t o, uint64_t g0, uint64_t n,
                       uint64_t pat, uint32_t pat_next, BlockMarks& m, BlockSum& s)
{
    walk_block_t<kBlockBytes, TileView>(v, o, g0, n, pat, pat_next, m, s);
}

/* summary of a block as an element of the tile algebra */
HBS_HD TileAgg as_agg(const BlockSum& s) { TileAgg a; a.cnt = s.cnt; a.known = s.known; a.sig = s.carry; a.last = s.last; return a; }

/*
 * Pass 1 for one thread: pattern masks of its four consecutive blocks go to
 * pats[4*tid .. 4*tid+3] (LDS), the return value is the thread's summary.
 * o0 = logical offset of the thread's first byte (multiple of 256), g0 = its
 * stream offset.  The last thread of the tile also records the patterns that
 * end in the first two bytes after the tile (pats[kBlocks]).
 */
HBS_D TileAgg classify_thread(const TileView& v, int32_t o0, uint64_t g0, uint64_t n, uint64_t* pats, int tid)
{
    uint32_t aprev = zero_bytes_approx(v.dword(o0 - 4));
    uint64_t pat = block_patterns(v, o0, aprev);
    TileAgg acc = {0u, 0u, 0u, kKindNone};
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int b = 0; b < kBlocksPerThread; ++b) {
        const int32_t o = o0 + kBlockBytes * b;
        uint64_t nxt;
        if (b + 1 < kBlocksPerThread) {
            nxt = block_patterns(v, o + kBlockBytes, aprev);
        } else {
            /* patterns ending in the first two bytes behind this thread's bytes */
            uint32_t z = zero_bytes(v.dword(o + kBlockBytes - 4));
            nxt = movemask4(pattern_marks(v.dword(o + kBlockBytes), z)) & 3u;
        }
        pats[kBlocksPerThread * tid + b] = pat;
        if (b + 1 == kBlocksPerThread && tid == kThreads - 1) pats[kBlocks] = nxt;
        BlockMarks m;
        BlockSum s;
        walk_block(v, o, g0 + (uint64_t)(kBlockBytes * b), n, pat, (uint32_t)(nxt & 3u), m, s);
        acc = combine(acc, as_agg(s));
        pat = nxt;
    }
    return acc;
}

/* where the emit pass writes to */
struct EmitTarget {
    hbs_nal_entry* index;
    uint64_t index_cap;
    RunHeader* hdr;
};

/* workgroup barrier on the device; the CPU single-stepper runs the two halves of
 * emit_thread as separate loops instead (see tests/sim/hbs_sim.cpp) */
HBS_D void tile_barrier()
{
#if defined(__HIP_DEVICE_COMPILE__)
    __syncthreads();
#endif
}

HBS_D void atomic_or_status(hbs_nal_entry* e, int32_t bits)
{
#if defined(__HIP_DEVICE_COMPILE__)
    atomicOr(reinterpret_cast<int*>(&e->status), bits);
#else
    e->status |= bits;
#endif
}
HBS_D void atomic_min_u64(unsigned long long* p, unsigned long long v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMin(p, v);
#else
    if (v < *p) *p = v;
#endif
}
HBS_D void flag_error(RunHeader* hdr, uint32_t code)
{
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMax(&hdr->error, code);
#else
    if (code > hdr->error) hdr->error = code;
#endif
}

/*
 * Second half for one block, once its incoming state is known: write the index
 * entries its events open/close, raise status flags, and return the final keep
 * mask.  nal_ord = number of NAL starts before this block (global ordinal of
 * the next NAL to open); rbsp_pos = arena offset of the block's first kept byte.
 */
template <int B, class View, class Mask = uint64_t>
HBS_D Mask emit_block_t(const View& v, int32_t o, uint64_t g0, const BlockMarksT<Mask>& m, bool inside,
                        uint64_t nal_ord, uint64_t rbsp_pos, const EmitTarget& tgt)
{
    const Mask one = 1;
    Mask inside_mask = 0;
    uint32_t cur = 0;
    uint64_t ord = nal_ord;

    for (Mask r = m.ev; r != 0; r &= r - 1) {
        const uint32_t e = mask_ctz(r);
        if (inside) {
            inside_mask |= mask_below<Mask>(e) & ~mask_below<Mask>(cur);
            /* NAL ord-1 ends where the terminator begins (h264_nal.c:74) */
            const uint64_t k = ord - 1;
            const int32_t t = o + (int32_t)e - 2;             /* first byte of the terminator */
            const uint32_t b3 = v.byte(t - 3), b2 = v.byte(t - 2), b1 = v.byte(t - 1);
            if (k < tgt.index_cap) {
                tgt.index[k].end = g0 + e - 2;
                if (b3 == 0 && b2 == 0 && b1 == 3)            /* h264_nal.c:170-173 */
                    atomic_or_status(&tgt.index[k], HBS_ST_TRAILING03);
            }
            if (b3 == 0 && b2 == 0 && b1 == 1)                /* empty NAL: loop of hevc_analyze.c:135 stops */
                atomic_min_u64(&tgt.hdr->first_empty, k);
        }
        if ((m.ev_start >> e) & one) {
            const uint64_t k = ord++;
            if (k < tgt.index_cap) {
                tgt.index[k].start = g0 + e + 1;              /* h264_nal.c:61-62 */
                tgt.index[k].rbsp_off = rbsp_pos + mask_popc((Mask)(m.cand & inside_mask));
            } else {
                flag_error(tgt.hdr, (uint32_t)(-HBS_E_CAPACITY));
            }
            inside = true;
        } else {
            inside = false;
        }
        cur = e + 1;
    }
    if (inside) inside_mask |= mask_below<Mask>((uint32_t)B) & ~mask_below<Mask>(cur);

    for (Mask r = m.err & inside_mask; r != 0; r &= r - 1) {
        const uint32_t pos = mask_ctz(r);
        const uint64_t k = nal_ord + mask_popc((Mask)(m.ev_start & mask_below<Mask>(pos))) - 1;
        if (k < tgt.index_cap) atomic_or_status(&tgt.index[k], HBS_ST_ERROR);
    }
    return m.cand & inside_mask;
}

HBS_D uint64_t emit_block(const TileView& v, int32_t o, uint64_t g0, const BlockMarks& m, bool inside,
                          uint64_t nal_ord, uint64_t rbsp_pos, const EmitTarget& tgt)
{
    return emit_block_t<kBlockBytes, TileView>(v, o, g0, m, inside, nal_ord, rbsp_pos, tgt);
}

/*
 * Pass 2 for one thread, once the tile's exclusive prefix is known: re-derive
 * each block's marks from its stored pattern mask, emit its events, and leave
 * the final keep masks and ranks in LDS for the gather.  keep[] holds the
 * pattern masks on entry (pass 1) and the keep masks on exit; a thread needs
 * its right neighbour's first mask, so all of those are read ahead of a
 * barrier before anyone overwrites an entry.
 */
struct ThreadStart {
    uint32_t in_state;   /* 0 outside, 1 inside, 2 = the tile's carried state */
    uint32_t cnt, known, sig;   /* exclusive prefix at the thread's first byte */
};

HBS_D void emit_thread(const TileView& v, int32_t o0, uint64_t g0, uint64_t n, int tid,
                       const ThreadStart& ts, const Prefix& excl, uint64_t* keep, uint32_t* rank,
                       const EmitTarget& tgt)
{
    const uint64_t right = keep[kBlocksPerThread * tid + kBlocksPerThread];   /* neighbour's first mask */
    tile_barrier();          /* every neighbour mask is read before any keep mask is written */

    uint32_t st = ts.in_state, pk = ts.known, ps = ts.sig, pc = ts.cnt;
    uint64_t pat = keep[kBlocksPerThread * tid];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int b = 0; b < kBlocksPerThread; ++b) {
        const int32_t o = o0 + kBlockBytes * b;
        const uint64_t gb = g0 + (uint64_t)(kBlockBytes * b);
        const uint64_t nxt = (b + 1 < kBlocksPerThread) ? keep[kBlocksPerThread * tid + b + 1] : right;
        BlockMarks m;
        BlockSum s;
        walk_block(v, o, gb, n, pat, (uint32_t)(nxt & 3u), m, s);
        const bool inside = (st == 1u) || (st == 2u && excl.inside);
        const uint32_t rank0 = pk + (excl.inside ? ps : 0u);
        const uint64_t km = emit_block(v, o, gb, m, inside, excl.nals + pc, excl.kept + rank0, tgt);
        keep[kBlocksPerThread * tid + b] = km;
        rank[kBlocksPerThread * tid + b] = rank0;
        pk += s.known + (st == 1u ? s.carry : 0u);
        ps += (st == 2u) ? s.carry : 0u;
        pc += s.cnt;
        if (s.last != kKindNone) st = (s.last == kKindStart) ? 1u : 0u;
        pat = nxt;
    }
}

/*
 * Compaction, input driven.  The tile is cut into 16-byte chunks; chunk c
 * (bytes [16c, 16c+16) of the tile, quarter c&3 of block c>>2) knows where its
 * kept bytes go: tile rank = rank[block] + kept bytes of the block in front of
 * the quarter.  A chunk that is kept whole -- nearly all of them -- is one
 * 16-byte load from the image and one 16-byte store to arena + rank; the
 * destination is byte-aligned only, which gfx950 global stores handle at ~92 %
 * of the aligned rate (scripts/ubench/unaligned_store.hip).  Chunks with holes
 * are compacted byte by byte.
 */
struct ChunkDest {
    uint32_t sub;        /* keep bits of the chunk's 16 bytes            */
    uint32_t rank;       /* tile rank of its first kept byte             */
};

HBS_D ChunkDest chunk_dest(const uint32_t* rank, const uint64_t* keep, uint32_t c)
{
    ChunkDest d;
    const uint32_t b = c >> 2, q = c & 3u;
    const uint64_t km = keep[b];
    d.sub = (uint32_t)(km >> (16u * q)) & 0xFFFFu;
    d.rank = rank[b] + popc64(km & below(16u * q));
    return d;
}

/* kept bytes of a chunk with holes, packed low-to-high into lo/hi; returns how many */
HBS_D uint32_t compact_chunk(const TileView& v, uint32_t c, uint32_t sub, uint64_t& lo, uint64_t& hi)
{
    lo = 0; hi = 0;
    uint32_t o = 0;
    for (uint32_t r = sub; r != 0; r &= r - 1, ++o) {
        const uint32_t pos = (uint32_t)__builtin_ctz(r);
        const uint64_t val = (uint64_t)v.byte((int32_t)(16u * c + pos)) << (8u * (o & 7u));
        lo |= (o < 8) ? val : 0ull;
        hi |= (o < 8) ? 0ull : val;
    }
    return o;
}

/*
 * End-of-stream rules that need the finished index: executed once, after every
 * tile has been emitted (single thread).  Handles
 *   - a start code in the last 4 bytes (h264_nal.c:52 bound and its unchecked
 *     first candidate :46-48), which can append one final 0- or 1-byte NAL;
 *   - the final NAL left open: nal_end = size, return -1 (h264_nal.c:71);
 *   - truncation at the first empty NAL (hevc_analyze.c:135: loop ends on 0);
 *   - the summary block.
 * tail[] holds the last 8 stream bytes: tail[i] = S[n-8+i], 0xFF before the
 * stream start.  Returns through `sum`; may bump hdr->final_* for an appended NAL.
 */
struct TailOut { uint64_t found0, kept0, found, kept; };      /* NALs / kept bytes as the tiles left them, and with an appended last NAL */

/* commit = false: hdr->final_nals / final_kept stay as the tiles left them (k_scan_finish: other threads of the same launch
 * still read them); the caller takes the final values from the return value */
HBS_D TailOut tail_fixup(RunHeader* hdr, hbs_nal_entry* index, uint64_t index_cap,
                         uint8_t* rbsp, uint64_t rbsp_cap,
                         const uint8_t* tail, uint64_t n, hbs_summary* sum, bool commit = true)
{
    uint64_t found = hdr->final_nals;
    uint64_t kept = hdr->final_kept;
    TailOut ret;
    ret.found0 = found; ret.kept0 = kept;
    bool inside = hdr->final_inside != 0;
    const uint8_t* e = tail + 8;                 /* e[-1] is the last stream byte */

    bool appended = false;
    if (!inside) {
        const bool have_prev = found > 0 && found - 1 < index_cap;
        const uint64_t prev_end = have_prev ? index[found - 1].end : 0;
        if (n >= 4 && e[-4] == 0 && e[-3] == 0 && e[-2] == 1) {
            /* 00 00 01 x at the very end: accepted when reached through the
             * 4-byte form one position earlier (i = n-5 passes :52), or when it
             * is where the search starts (previous NAL ended exactly here, or
             * stream start) */
            const bool ok = (n >= 5 && e[-5] == 0) || (found > 0 && prev_end == n - 4) || n == 4;
            if (ok) {
                if (found < index_cap) {
                    index[found].start = n - 1;
                    index[found].end = n;
                    index[found].rbsp_off = kept;
                }
                if (rbsp != nullptr && kept < rbsp_cap) rbsp[kept] = e[-1];
                kept += 1;
                found += 1;
                appended = true;
            }
        } else if (n >= 3 && e[-3] == 0 && e[-2] == 0 && e[-1] == 1) {
            /* 00 00 01 as the last three bytes: only the unchecked first
             * candidate of a search can accept it */
            bool ok;
            if (n >= 4 && e[-4] == 0) ok = (found > 0) ? (prev_end == n - 4) : (n == 4);
            else ok = (found == 0 && n == 3);
            if (ok) {
                if (found < index_cap) {
                    index[found].start = n;
                    index[found].end = n;
                    index[found].rbsp_off = kept;
                }
                found += 1;
                appended = true;
            }
        }
    }

    int32_t stop = 0;
    if ((inside || appended) && found > 0) {
        stop = -1;
        if (found - 1 < index_cap) {
            hbs_nal_entry* last = &index[found - 1];
            last->end = n;
            last->status |= HBS_ST_UNTERMINATED;
            if (n - last->start >= 3 && e[-3] == 0 && e[-2] == 0 && e[-1] == 3)
                last->status |= HBS_ST_TRAILING03;
        }
    }
    uint64_t count = found;
    if (hdr->first_empty < count) { count = hdr->first_empty; stop = 1; }
    if (count > index_cap) { count = index_cap; flag_error(hdr, (uint32_t)(-HBS_E_CAPACITY)); }
    if (rbsp != nullptr && kept > rbsp_cap) flag_error(hdr, (uint32_t)(-HBS_E_CAPACITY));

    if (commit) {
        hdr->final_nals = found;
        hdr->final_kept = kept;
    }
    ret.found = found; ret.kept = kept;
    sum->nal_count = count;
    sum->nal_found = found;
    sum->rbsp_bytes = kept;
    sum->stream_bytes = n;
    sum->stop_reason = stop;
    sum->error = -(int32_t)hdr->error;
    sum->reserved[0] = sum->reserved[1] = sum->reserved[2] = 0;
    return ret;
}

/* rbsp_len of NAL k from the packed arena offsets (grid-stride over NALs); found / final_kept: the finished run's */
HBS_D void fill_rbsp_len_v(hbs_nal_entry* index, uint64_t index_cap, uint64_t k, uint64_t found, uint64_t final_kept)
{
    if (k >= found || k >= index_cap) return;
    const uint64_t next = (k + 1 < found && k + 1 < index_cap) ? index[k + 1].rbsp_off : final_kept;
    index[k].rbsp_len = (uint32_t)(next - index[k].rbsp_off);
    /* a rejected NAL reports no consumed size (h264_nal.c:158,166 return before :197) */
    if (index[k].status & HBS_ST_ERROR) index[k].status &= ~HBS_ST_TRAILING03;
}
HBS_D void fill_rbsp_len(const RunHeader* hdr, hbs_nal_entry* index, uint64_t index_cap, uint64_t k)
{
    fill_rbsp_len_v(index, index_cap, k, hdr->final_nals, hdr->final_kept);
}

} // namespace hbs
#endif
