/*
 * hbs_tile.h -- per-tile logic of the fused scan / index / RBSP-extract kernel.
 *
 * A tile is 16 KiB of the stream staged in LDS; each of the 256 threads owns
 * 64 contiguous bytes ("block").  Everything here is per-thread code over the
 * LDS image plus a few words of carried state, so it compiles for gfx950 and,
 * under tests/sim, for the host (see hbs_common.h).
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

/* per-thread registers that live across the look-back wait */
struct BlockMarks {
    uint64_t cand;        /* bytes that are RBSP if their position is inside a NAL */
    uint64_t ev;          /* terminator events (position of the pattern's last byte) */
    uint64_t ev_start;    /* subset of ev that also starts a NAL                    */
    uint64_t err;         /* positions that make nal_to_rbsp fail if inside a NAL   */
};

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
HBS_HD uint64_t pack_agg0(TileAgg a)   { return kDescAgg | ((uint64_t)a.last << 2) | ((uint64_t)a.known << 4) | ((uint64_t)a.sig << 20); }
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
    a.known = (uint32_t)(w0 >> 4) & 0xFFFFu;
    a.sig = (uint32_t)(w0 >> 20) & 0xFFFFu;
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
HBS_HD int pattern_kind(const uint8_t* at, uint64_t gj, uint64_t n)
{
    const uint32_t v = at[0];
    if (v == 3) return kPatEpb;
    if (v == 2) return kPatErr;
    if (v == 0 && at[-3] == 0) return kPatSkip;          /* not the first 00 00 00 of a zero run */
    bool term_ok = gj + 2 <= n;
    if (!term_ok) term_ok = (at[-5] == 0 && at[-4] == 0 && at[-3] == 1);
    if (!term_ok) return kPatErr;
    return (v == 1 && gj + 3 <= n) ? kPatStart : kPatStop;
}

/*
 * Classify one 64-byte block.  `blk` points at the block's first byte inside
 * the LDS image (4-byte aligned; at least 8 valid bytes before and after).
 * g0 = stream offset of the block.
 */
HBS_HD void classify_block(const uint8_t* blk, uint64_t g0, uint64_t n, BlockMarks& m, BlockSum& s)
{
    const uint32_t* w = reinterpret_cast<const uint32_t*>(blk);
    uint32_t p[17];
    uint32_t any = 0;
    uint32_t zprev = zero_bytes(w[-1]);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 17; ++k) {
        const uint32_t x = w[k];
        const uint32_t z = zero_bytes(x);
        const uint32_t lo = zero_bytes(x & 0xFCFCFCFCu);          /* bytes <= 3              */
        const uint32_t z1 = alignbyte(z, zprev, 3);               /* byte j-1 is zero        */
        const uint32_t z2 = alignbyte(z, zprev, 2);               /* byte j-2 is zero        */
        p[k] = z1 & z2 & lo;                                      /* 00 00 {00..03} ends at j */
        any |= p[k];
        zprev = z;
    }

    const uint32_t nvalid = (g0 >= n) ? 0u : (n - g0 >= 64 ? 64u : (uint32_t)(n - g0));
    m.cand = below(nvalid);
    m.ev = m.ev_start = m.err = 0;

    if (any != 0) {
        uint64_t pat = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int k = 0; k < 16; ++k) pat |= (uint64_t)movemask4(p[k]) << (4 * k);
        const uint32_t pat_next = movemask4(p[16]) & 3u;          /* patterns ending at 64, 65 */

        for (uint64_t r = pat; r != 0; r &= r - 1) {
            const uint32_t j = ctz64(r);
            const uint64_t bit = 1ull << j;
            const int kind = pattern_kind(blk + j, g0 + j, n);
            if (kind == kPatEpb) {
                m.cand &= ~bit;
                /* 00 00 03 followed by a byte > 3 that is still part of the stream
                 * (h264_nal.c:164: i < nal_size-1 && next > 3) */
                if (blk[j + 1] > 3 && g0 + j + 1 < n) m.err |= bit;
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
                const uint32_t j = 64 + b;
                const int kind = pattern_kind(blk + j, g0 + j, n);
                if (kind == kPatStart || kind == kPatStop)
                    m.cand &= ~((b == 0) ? (3ull << 62) : (1ull << 63));
            }
        }
    }

    /* summary: bytes before the first event follow the carried state */
    s.cnt = popc64(m.ev_start);
    if (m.ev == 0) {
        s.known = 0;
        s.carry = popc64(m.cand);
        s.last = kKindNone;
    } else {
        uint32_t cur = ctz64(m.ev);
        s.carry = popc64(m.cand & below(cur));
        uint32_t known = 0;
        bool inside = false;
        for (uint64_t r = m.ev; r != 0; r &= r - 1) {
            const uint32_t e = ctz64(r);
            if (inside) known += popc64(m.cand & below(e) & ~below(cur));
            inside = (m.ev_start >> e) & 1ull;
            cur = e + 1;
        }
        if (inside) known += popc64(m.cand & ~below(cur));
        s.known = known;
        s.last = inside ? kKindStart : kKindStop;
    }
}

/* where the emit pass writes to */
struct EmitTarget {
    hbs_nal_entry* index;
    uint64_t index_cap;
    RunHeader* hdr;
};

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
HBS_D uint64_t emit_block(const uint8_t* blk, uint64_t g0, const BlockMarks& m, bool inside,
                          uint64_t nal_ord, uint64_t rbsp_pos, const EmitTarget& tgt)
{
    uint64_t inside_mask = 0;
    uint32_t cur = 0;
    uint64_t ord = nal_ord;

    for (uint64_t r = m.ev; r != 0; r &= r - 1) {
        const uint32_t e = ctz64(r);
        if (inside) {
            inside_mask |= below(e) & ~below(cur);
            /* NAL ord-1 ends where the terminator begins (h264_nal.c:74) */
            const uint64_t k = ord - 1;
            const uint8_t* t = blk + e - 2;                   /* first byte of the terminator */
            if (k < tgt.index_cap) {
                tgt.index[k].end = g0 + e - 2;
                if (t[-3] == 0 && t[-2] == 0 && t[-1] == 3)   /* h264_nal.c:170-173 */
                    atomic_or_status(&tgt.index[k], HBS_ST_TRAILING03);
            }
            if (t[-3] == 0 && t[-2] == 0 && t[-1] == 1)       /* empty NAL: loop of hevc_analyze.c:135 stops */
                atomic_min_u64(&tgt.hdr->first_empty, k);
        }
        if ((m.ev_start >> e) & 1ull) {
            const uint64_t k = ord++;
            if (k < tgt.index_cap) {
                tgt.index[k].start = g0 + e + 1;              /* h264_nal.c:61-62 */
                tgt.index[k].rbsp_off = rbsp_pos + popc64(m.cand & inside_mask);
            } else {
                flag_error(tgt.hdr, (uint32_t)(-HBS_E_CAPACITY));
            }
            inside = true;
        } else {
            inside = false;
        }
        cur = e + 1;
    }
    if (inside) inside_mask |= ~below(cur);

    for (uint64_t r = m.err & inside_mask; r != 0; r &= r - 1) {
        const uint32_t pos = ctz64(r);
        const uint64_t k = nal_ord + popc64(m.ev_start & below(pos)) - 1;
        if (k < tgt.index_cap) atomic_or_status(&tgt.index[k], HBS_ST_ERROR);
    }
    return m.cand & inside_mask;
}

/*
 * Output-driven compaction.  The tile's kept bytes, in order, go to
 * arena[kept_base, kept_base + tile_kept).  Output is produced in 16-byte words
 * aligned in the ARENA; word `wi` of the tile covers tile ranks
 * [16*wi - ob, 16*wi - ob + 16) with ob = kept_base & 15.  rank[b] is the tile
 * rank of block b's first kept byte (rank[256] = tile_kept), keep[b] its keep
 * mask, raw the tile image.  Returns the number of valid bytes and their
 * position inside the word via lo/hi (bytes [lo,hi) of the word are ours).
 */
struct GatherOut { uint32_t w[4]; uint32_t lo, hi; };

HBS_D GatherOut gather_word(const uint8_t* raw_tile /* image of stream byte tile_base */,
                            const uint32_t* rank, const uint64_t* keep,
                            uint32_t wi, uint32_t ob, uint32_t tile_kept)
{
    GatherOut g;
    const int32_t r_first = (int32_t)(16u * wi) - (int32_t)ob;     /* rank of byte 0 of the word */
    const int32_t r_lo = r_first < 0 ? 0 : r_first;
    const int32_t r_hi = (r_first + 16 > (int32_t)tile_kept) ? (int32_t)tile_kept : r_first + 16;
    g.lo = (uint32_t)(r_lo - r_first);
    g.hi = (uint32_t)(r_hi - r_first);
    g.w[0] = g.w[1] = g.w[2] = g.w[3] = 0;
    if (r_hi <= r_lo) { g.hi = g.lo; return g; }

    /* block holding rank r_lo: rank[b] <= r_lo < rank[b+1]; rank[b] <= 64 b */
    uint32_t b = (uint32_t)r_lo >> 6;
    if (rank[b + 1] <= (uint32_t)r_lo) {
        uint32_t lo_b = b + 1, hi_b = kThreads - 1;                /* binary search, rare */
        while (lo_b < hi_b) {
            const uint32_t mid = (lo_b + hi_b) >> 1;
            if (rank[mid + 1] <= (uint32_t)r_lo) lo_b = mid + 1; else hi_b = mid;
        }
        b = lo_b;
    }
    const uint32_t skip = (uint32_t)r_lo - rank[b];                 /* kept bytes of block b before ours */
    uint64_t km = keep[b];

    /* fast path: a full word whose 16 source bytes are contiguous */
    if (g.lo == 0 && g.hi == 16) {
        bool contiguous = false;
        uint32_t src = 0;
        if (km == ~0ull) {
            src = 64u * b + skip;
            contiguous = (skip <= 48) || (b + 1 < (uint32_t)kThreads && keep[b + 1] == ~0ull);
        }
        if (contiguous) {
            const uint32_t* a = reinterpret_cast<const uint32_t*>(raw_tile + (src & ~3u));
            const uint32_t sh = src & 3u;
            const uint32_t d0 = a[0], d1 = a[1], d2 = a[2], d3 = a[3], d4 = a[4];
            g.w[0] = alignbyte(d1, d0, sh);
            g.w[1] = alignbyte(d2, d1, sh);
            g.w[2] = alignbyte(d3, d2, sh);
            g.w[3] = alignbyte(d4, d3, sh);
            return g;
        }
    }

    /* general path: walk the keep masks byte by byte */
    for (uint32_t i = 0; i < skip; ++i) km &= km - 1;
    for (uint32_t o = g.lo; o < g.hi; ++o) {
        while (km == 0 && b + 1 < (uint32_t)kThreads) { ++b; km = keep[b]; }
        if (km == 0) break;                                   /* cannot happen: ranks are consistent */
        const uint32_t pos = ctz64(km);
        km &= km - 1;
        const uint32_t v = raw_tile[64u * b + pos];
        g.w[o >> 2] |= v << (8u * (o & 3u));
    }
    return g;
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
HBS_D void tail_fixup(RunHeader* hdr, hbs_nal_entry* index, uint64_t index_cap,
                      uint8_t* rbsp, uint64_t rbsp_cap,
                      const uint8_t* tail, uint64_t n, hbs_summary* sum)
{
    uint64_t found = hdr->final_nals;
    uint64_t kept = hdr->final_kept;
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

    hdr->final_nals = found;
    hdr->final_kept = kept;
    sum->nal_count = count;
    sum->nal_found = found;
    sum->rbsp_bytes = kept;
    sum->stream_bytes = n;
    sum->stop_reason = stop;
    sum->error = -(int32_t)hdr->error;
    sum->reserved[0] = sum->reserved[1] = sum->reserved[2] = 0;
}

/* rbsp_len of NAL k from the packed arena offsets (grid-stride over NALs) */
HBS_D void fill_rbsp_len(const RunHeader* hdr, hbs_nal_entry* index, uint64_t index_cap, uint64_t k)
{
    const uint64_t found = hdr->final_nals;
    if (k >= found || k >= index_cap) return;
    const uint64_t next = (k + 1 < found && k + 1 < index_cap) ? index[k + 1].rbsp_off : hdr->final_kept;
    index[k].rbsp_len = (uint32_t)(next - index[k].rbsp_off);
    /* a rejected NAL reports no consumed size (h264_nal.c:158,166 return before :197) */
    if (index[k].status & HBS_ST_ERROR) index[k].status &= ~HBS_ST_TRAILING03;
}

} // namespace hbs
#endif
