/*
 * hbs_elems.h -- what the event-sparse scan kernels share (hbs_scan4.hip: scan + extract in one pass;
 * hbs_scan5.hip: the index-only pair of passes): wave-level helpers over the tile algebra, the decoupled
 * look-back over tile descriptors (one wavefront, 256 predecessors per step), and the two halves of an
 * element -- the exact window rules on one flagged 16-byte chunk, then index entries and kept bytes once
 * the state carried into the tile is known.  Device code only.
 */
#ifndef HBS_ELEMS_H
#define HBS_ELEMS_H

#include "hbs_wave.h"
#include "hbs_sparse.h"
#include "hbs_scan.h"

namespace hbs {

__device__ __forceinline__ uint32_t lanes_below(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__device__ __forceinline__ TileAgg agg_shfl_up(const TileAgg& a, int d)
{
    TileAgg t;
    t.cnt = __shfl_up(a.cnt, d, 64); t.known = __shfl_up(a.known, d, 64);
    t.sig = __shfl_up(a.sig, d, 64); t.last = __shfl_up(a.last, d, 64);
    return t;
}

/* the aggregate of the lane in front (lane 0: the identity) */
__device__ __forceinline__ TileAgg agg_prev_lane(const TileAgg& a)
{
    TileAgg t;
    t.cnt = dpp_or_zero<kDppWaveShr1, 0xF>(a.cnt); t.known = dpp_or_zero<kDppWaveShr1, 0xF>(a.known);
    t.sig = dpp_or_zero<kDppWaveShr1, 0xF>(a.sig); t.last = dpp_or_zero<kDppWaveShr1, 0xF>(a.last);
    return t;
}

/* inclusive scan with combine(): lane l <- elements of lanes 0..l in order.  Six DPP steps (hbs_wave.h): lanes without a source
 * read the identity, and combine(identity, a) = a. */
template <int kCtrl, int kRowMask>
__device__ __forceinline__ TileAgg agg_dpp(const TileAgg& a)
{
    TileAgg t;
    t.cnt = dpp_or_zero<kCtrl, kRowMask>(a.cnt); t.known = dpp_or_zero<kCtrl, kRowMask>(a.known);
    t.sig = dpp_or_zero<kCtrl, kRowMask>(a.sig); t.last = dpp_or_zero<kCtrl, kRowMask>(a.last);
    return t;
}
__device__ __forceinline__ TileAgg wave_scan_combine(TileAgg a, int /*lane*/)
{
    a = combine(agg_dpp<kDppRowShr1, 0xF>(a), a);
    a = combine(agg_dpp<kDppRowShr2, 0xF>(a), a);
    a = combine(agg_dpp<kDppRowShr4, 0xF>(a), a);
    a = combine(agg_dpp<kDppRowShr8, 0xF>(a), a);
    a = combine(agg_dpp<kDppBcast15, 0xA>(a), a);
    a = combine(agg_dpp<kDppBcast31, 0xC>(a), a);
    return a;
}

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int l)
{
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32) |
           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
}

#ifndef HBS_LOOKBACK_GROUPS
#define HBS_LOOKBACK_GROUPS 4
#endif
/* Four 16-byte descriptor loads that bypass the non-coherent cache levels (sc1: the writer is on
 * another XCD), issued together and awaited together.  A descriptor's two 8-byte words are each
 * self-validating, so reading them with one 16-byte load is as good as two 8-byte atomics. */
__device__ __forceinline__ void load_desc4(u32x4& d0, u32x4& d1, u32x4& d2, u32x4& d3,
                                           const unsigned long long* p0, const unsigned long long* p1,
                                           const unsigned long long* p2, const unsigned long long* p3)
{
    asm volatile("global_load_dwordx4 %0, %4, off sc1\n\t"
                 "global_load_dwordx4 %1, %5, off sc1\n\t"
                 "global_load_dwordx4 %2, %6, off sc1\n\t"
                 "global_load_dwordx4 %3, %7, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3)
                 : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
                 : "memory");
}

/* ... eight of them (round 5: a window of 512 tiles) */
__device__ __forceinline__ void load_desc8(u32x4* d, const unsigned long long* const* p)
{
    asm volatile("global_load_dwordx4 %0, %8, off sc1\n\t"
                 "global_load_dwordx4 %1, %9, off sc1\n\t"
                 "global_load_dwordx4 %2, %10, off sc1\n\t"
                 "global_load_dwordx4 %3, %11, off sc1\n\t"
                 "global_load_dwordx4 %4, %12, off sc1\n\t"
                 "global_load_dwordx4 %5, %13, off sc1\n\t"
                 "global_load_dwordx4 %6, %14, off sc1\n\t"
                 "global_load_dwordx4 %7, %15, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7])
                 : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])
                 : "memory");
}

/*
 * Decoupled look-back by ONE wavefront, 64 * kGroups predecessors per step (kGroups = 4 or 8).  Lane l reads the descriptors
 * of the tiles at distance l, 64 + l, 128 + l, ... in front of win_hi: each of the loads covers 64 consecutive descriptors,
 * 1 KiB, eight cache lines.  The groups are folded nearest first up to the nearest tile that already has its prefix.
 * Returns false on timeout/abort.  Called by every lane of wavefront 0.
 * (Round 5, HBS_LOOKBACK_GROUPS: with 512 tiles in flight that move nearly in step, the nearest tile that has its prefix is
 * 300-500 tiles back: two steps of 256, two round trips at loaded latency -- ~4 us of a tile's 34.)
 */
/* first half: the tile's aggregate goes out (tile 0 publishes its prefix straight away, in the second half) */
__device__ __forceinline__ void look_back_publish(unsigned long long* desc, uint64_t tile, const TileAgg& mine, int lane)
{
    if (tile != 0 && lane == 0) {
        st_desc3(&desc[2 * tile], pack_agg0(mine));
        st_desc3(&desc[2 * tile + 1], pack_agg1(mine));
    }
}

/* second half: fold the tiles in front, publish the inclusive prefix.  May run long after the first (an experiment of round 2
 * did other work in between, so that the predecessors had published by the time it asked). */
template <int kGroups>
__device__ __forceinline__ bool look_back_resolve(unsigned long long* desc, uint64_t tile, const TileAgg& mine,
                                                  RunHeader* hdr, int lane, Prefix& excl, uint32_t& dbg_iters, uint32_t& dbg_stalls)
{
    static_assert(kGroups == 4 || kGroups == 8, "four or eight groups of 64 descriptors");
    dbg_iters = 0; dbg_stalls = 0;
    bool ok = true;
    excl.kept = 0; excl.nals = 0; excl.inside = 0;
    if (tile != 0) {
        TileAgg acc = agg_identity();                 /* tiles between the window and `tile` */
        int64_t win_hi = (int64_t)tile - 1;
        uint32_t spins = 0;
        uint64_t w0[kGroups], w1[kGroups];
        bool fresh = true;                            /* the window moved: read all of it */
        for (;;) {
            ++dbg_iters;
            /* A descriptor that has been seen ready stays usable (an aggregate can only turn into
             * a prefix): while waiting, only lanes that still miss one read again. */
            bool need = fresh;
#pragma unroll
            for (int j = 0; j < kGroups; ++j) {
                const uint32_t s0 = (uint32_t)(w0[j] & 3u), s1 = (uint32_t)(w1[j] & 3u);
                need = need || !((s0 == s1) && (s0 != kDescEmpty));
            }
            if (need) {
                const int64_t t0 = win_hi - lane;
                u32x4 d[kGroups];
                const unsigned long long* p[kGroups];
#pragma unroll
                for (int j = 0; j < kGroups; ++j) {
                    const int64_t t = t0 - 64 * j;
                    p[j] = &desc[2 * (t > 0 ? t : 0)];
                }
                if constexpr (kGroups == 4) load_desc4(d[0], d[1], d[2], d[3], p[0], p[1], p[2], p[3]);
                else load_desc8(d, p);
#pragma unroll
                for (int j = 0; j < kGroups; ++j) {
                    w0[j] = ((uint64_t)d[j].y << 32) | d[j].x;
                    w1[j] = ((uint64_t)d[j].w << 32) | d[j].z;
                    if (t0 - 64 * j < 0) { w0[j] = kDescPrefix; w1[j] = kDescPrefix; }   /* virtual tile -1: empty prefix */
                }
            }
            fresh = false;
            /* nearest group that holds a prefix, with everything in front of it ready */
            int j0 = -1, lstar = 64;
            bool stall = false;
            uint64_t pw0 = 0, pw1 = 0;
#pragma unroll
            for (int j = 0; j < kGroups; ++j) {
                const uint32_t s0 = (uint32_t)(w0[j] & 3u), s1 = (uint32_t)(w1[j] & 3u);
                const bool ready = (s0 == s1) && (s0 != kDescEmpty);
                const uint64_t m_ready = __ballot(ready);
                const uint64_t m_pre = __ballot(ready && s0 == kDescPrefix);
                if (j0 < 0 && !stall) {
                    if (m_pre != 0) {
                        const int ls = (int)__builtin_ctzll(m_pre);
                        const uint64_t front = (1ull << ls) - 1ull;
                        if ((m_ready & front) != front) stall = true;
                        else { j0 = j; lstar = ls; pw0 = w0[j]; pw1 = w1[j]; }
                    } else if (m_ready != ~0ull) {
                        stall = true;
                    }
                }
            }
            if (stall) {
                ++dbg_stalls;
                bool aborted = false;
                if ((spins & 63u) == 63u)
                    aborted = __hip_atomic_load(&hdr->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
                if (++spins > (1u << 20) || aborted) { ok = false; break; }
                continue;               /* the round trip of the next poll is delay enough */
            }
            TileAgg total = acc;
#pragma unroll
            for (int j = 0; j < kGroups; ++j) {
                if (j0 < 0 || j <= j0) {
                    const TileAgg win = window_fold3(unpack_agg(w0[j], w1[j]), (j == j0) ? lstar : 64, lane);
                    total = combine(win, total);
                }
            }
            if (j0 >= 0) {
                const Prefix p = unpack_pre(readlane_u64(pw0, lstar), readlane_u64(pw1, lstar));
                excl = fold(p, total);
                break;
            }
            acc = total;
            win_hi -= 64 * kGroups;
            fresh = true;
        }
    }
    if (lane == 0) {
        if (ok) {
            const Prefix incl = fold(excl, mine);
            st_desc3(&desc[2 * tile], pack_pre0(incl));
            st_desc3(&desc[2 * tile + 1], pack_pre1(incl));
        } else {
            __hip_atomic_store(&hdr->abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicMax(&hdr->error, (uint32_t)(-HBS_E_TIMEOUT));
        }
    }
    return ok;
}

__device__ __forceinline__ bool look_back4(unsigned long long* desc, uint64_t tile, const TileAgg& mine,
                                           RunHeader* hdr, int lane, Prefix& excl, uint32_t& dbg_iters, uint32_t& dbg_stalls)
{
    look_back_publish(desc, tile, mine, lane);
    return look_back_resolve<HBS_LOOKBACK_GROUPS>(desc, tile, mine, hdr, lane, excl, dbg_iters, dbg_stalls);
}

__device__ __forceinline__ Prefix prefix_uniform4(const Prefix& p)
{
    Prefix r;
    r.kept = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(p.kept >> 32)) << 32) |
             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)p.kept);
    r.nals = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(p.nals >> 32)) << 32) |
             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)p.nals);
    r.inside = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.inside);
    return r;
}

/* the seven dwords around chunk g0 of the stream, for an element */
__device__ __forceinline__ void elem_load(ElemView& v, const uint8_t* __restrict__ stream, uint64_t g0, uint64_t n, bool padded)
{
    if (g0 >= 8 && (padded || g0 + 20 <= n)) {
        const uint32_t* p = reinterpret_cast<const uint32_t*>(stream + g0);
        const u32x4 q = *reinterpret_cast<const u32x4*>(p);
        v.xpp = p[-2]; v.xp = p[-1]; v.xn = p[4];
        v.x0 = q.x; v.x1 = q.y; v.x2 = q.z; v.x3 = q.w;
    } else {
        const u32x4 q = load_chunk_guarded(stream, g0, n);
        v.xpp = load_dword_guarded(stream, (int64_t)g0 - 8, n);
        v.xp = load_dword_guarded(stream, (int64_t)g0 - 4, n);
        v.xn = load_dword_guarded(stream, (int64_t)g0 + 16, n);
        v.x0 = q.x; v.x1 = q.y; v.x2 = q.z; v.x3 = q.w;
    }
    v.stream = stream; v.g0 = g0; v.n = n;
}

__device__ __forceinline__ TileAgg agg_readlane(const TileAgg& a, int l)
{
    TileAgg r;
    r.cnt = (uint32_t)__builtin_amdgcn_readlane((int)a.cnt, l); r.known = (uint32_t)__builtin_amdgcn_readlane((int)a.known, l);
    r.sig = (uint32_t)__builtin_amdgcn_readlane((int)a.sig, l); r.last = (uint32_t)__builtin_amdgcn_readlane((int)a.last, l);
    return r;
}

#ifndef HBS_ELEM_BYTE_STORES
#define HBS_ELEM_BYTE_STORES 0
#endif
/* what a flagged lane leaves in LDS for the thread that will handle its chunk as an element */
struct Deposit { uint32_t xpp, xp, x0, x1, x2, x3, xn, chunk; };

/* lane L (a compile-time constant) of v <- the wave-uniform value s.  The s_nop covers gfx950's wait states between a
 * VALU instruction that writes an SGPR (the v_cmp of a ballot) and a VALU instruction reading it, which the compiler
 * cannot insert across an asm statement. */
template <int L>
__device__ __forceinline__ void write_lane_c(uint32_t& v, uint32_t s)
{
    asm volatile("s_nop 1\n\tv_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(s), "i"(L));
}

/* everything one lane of wavefront 0 knows about its element */
struct Elem {
    ElemView v;
    ChunkMarks m;
    BlockSum s;
    ElemClasses cls;       /* where its bytes [-8, 20) hold 00 / 01 / 03 */
    uint32_t gap;          /* bytes of the gap in front of it            */
    uint32_t chunk;        /* its chunk number in the tile               */
};

/* Second half for an element once the tile's carried state is known: index entries, its own kept
 * bytes, and the segment word for the chunks behind it.  e = tile aggregate in front of its gap. */
__device__ __forceinline__ void elem_emit(const Elem& el, const TileAgg& e, const Prefix& excl, bool can_store, uint8_t* out,
                                          const EmitTarget& tgt, uint32_t* seg_slot)
{
    const ElemStart st = elem_start(e, el.gap, excl.inside);
    const uint32_t keep = emit_chunk_fast(el.cls, el.v.g0, el.m, st.inside, excl.nals + e.cnt, excl.kept + st.kept, tgt);
    const uint32_t nk = (uint32_t)__builtin_popcount(keep);
    if (can_store && keep != 0u) {
        if (keep == 0xFFFFu) {
            u32x4 q; q.x = el.v.x0; q.y = el.v.x1; q.z = el.v.x2; q.w = el.v.x3;
            reinterpret_cast<Unaligned16_3*>(out + st.kept)->v = q;
        } else {
#if HBS_ELEM_BYTE_STORES
            /* a chunk with holes: every kept byte goes out by itself, to its rank (16 predicated byte stores and no
             * loop: a loop runs as long as its longest lane, and the workgroup waits for this wavefront) */
            uint8_t* const o = out + st.kept;
            const uint32_t d[4] = {el.v.x0, el.v.x1, el.v.x2, el.v.x3};
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if ((keep >> i) & 1u) o[__builtin_popcount(keep & ((1u << i) - 1u))] = (uint8_t)(d[i >> 2] >> (8 * (i & 3)));
#else
            uint64_t lo, hi;
            const uint32_t cn = compact_chunk_regs(el.v.x0, el.v.x1, el.v.x2, el.v.x3, keep, lo, hi);
            store_pieces(out + st.kept, lo, hi, cn);
#endif
        }
    }
    const bool after = (el.s.last != kKindNone) ? (el.s.last == kKindStart) : st.inside;
    *seg_slot = seg_pack((int32_t)el.chunk, st.kept + nk, after);
}

/* ---- a whole KiB row as 64 elements (dense tiles of hbs_scan4.hip, dense rows of hbs_scan5.hip) --------------------- */
struct DenseRow {
    Elem el;                /* this lane's chunk of the row */
    bool row_has_event;
};

/* row r of a wavefront's segment, this lane's chunk as an element.  qp / qc / qn: previous, current, next row. */
__device__ __forceinline__ void dense_row(DenseRow& d, const u32x4& qp, const u32x4& qc, const u32x4& qn, int r, int nrows,
                                          uint32_t before, uint32_t before2, uint32_t after,
                                          const uint8_t* src, uint64_t wseg, uint64_t n, uint32_t chunk0, int lane)
{
    const uint32_t e_prev_w = r == 0 ? before : (uint32_t)__builtin_amdgcn_readlane((int)qp.w, 63);
    const uint32_t e_prev_z = r == 0 ? before2 : (uint32_t)__builtin_amdgcn_readlane((int)qp.z, 63);
    const uint32_t e_next_x = r == nrows - 1 ? after : (uint32_t)__builtin_amdgcn_readlane((int)qn.x, 0);
    d.el.v.xpp = from_prev_lane(qc.z, e_prev_z);
    d.el.v.xp = from_prev_lane(qc.w, e_prev_w);
    d.el.v.x0 = qc.x; d.el.v.x1 = qc.y; d.el.v.x2 = qc.z; d.el.v.x3 = qc.w;
    d.el.v.xn = from_next_lane(qc.x, e_next_x);
    d.el.v.stream = src; d.el.v.g0 = wseg + 1024ull * (uint64_t)r + 16ull * (uint64_t)lane; d.el.v.n = n;
    elem_walk(d.el.v, d.el.m, d.el.s, d.el.cls);
    d.el.gap = 0;
    d.el.chunk = chunk0 + 64u * (uint32_t)r + (uint32_t)lane;
    d.row_has_event = __ballot(d.el.s.last != kKindNone) != 0ull;
}

/* A row of a dense tile, asked two things only: does a terminator (00 00 00 or 00 00 01) end in bytes [0, 18) of any of its
 * chunks, and if none does, how many of its bytes are kept?  Without a terminator no chunk of the row holds an event, so
 * the row's aggregate is a gap of "kept bytes" (elem_walk: cnt 0, known 0, last none, carry = the bytes that are not
 * emulation prevention bytes) -- and that needs two byte-flag words per dword instead of elem_walk's class masks and walk:
 * ~100 instructions a row against ~300.  Rows of 00 00 03 padding (cabac_zero_words) are exactly this case, and the time a
 * dense tile takes to publish its aggregate is what every tile behind it waits for (mixed streams).
 * The test is conservative (any 00 00 00 counts, not only the first of a run); a row that fails it takes the exact walk.
 * Valid for rows that end at least 64 bytes in front of the stream's end (the caller checks).
 * kept = kept bytes of this lane's chunk; returns true when some lane of the wavefront saw a terminator. */
template <bool kErrToo>
__device__ __forceinline__ bool dense_row_quick(const u32x4& qp, const u32x4& qc, const u32x4& qn, int r, int nrows,
                                                uint32_t before, uint32_t after, uint32_t& kept)
{
    const uint32_t e_prev_w = r == 0 ? before : (uint32_t)__builtin_amdgcn_readlane((int)qp.w, 63);
    const uint32_t e_next_x = r == nrows - 1 ? after : (uint32_t)__builtin_amdgcn_readlane((int)qn.x, 0);
    const uint32_t xp = from_prev_lane(qc.w, e_prev_w), xn = from_next_lane(qc.x, e_next_x);
    /* 0x80 per byte: zero, one, three */
    const uint32_t zp = zero_bytes(xp), z0 = zero_bytes(qc.x), z1 = zero_bytes(qc.y), z2 = zero_bytes(qc.z), z3 = zero_bytes(qc.w), zn = zero_bytes(xn);
    const uint32_t o0 = zero_bytes(qc.x ^ 0x01010101u), o1 = zero_bytes(qc.y ^ 0x01010101u), o2 = zero_bytes(qc.z ^ 0x01010101u),
                   o3 = zero_bytes(qc.w ^ 0x01010101u), on = zero_bytes(xn ^ 0x01010101u);
    const uint32_t t0 = zero_bytes(qc.x ^ 0x03030303u), t1 = zero_bytes(qc.y ^ 0x03030303u), t2 = zero_bytes(qc.z ^ 0x03030303u),
                   t3 = zero_bytes(qc.w ^ 0x03030303u);
    /* "the two bytes in front are zero", per byte: the zero flags moved up by one and by two bytes */
#define HBS_ZZ(cur, prev) (__builtin_amdgcn_alignbyte((cur), (prev), 3) & __builtin_amdgcn_alignbyte((cur), (prev), 2))
    const uint32_t zz0 = HBS_ZZ(z0, zp), zz1 = HBS_ZZ(z1, z0), zz2 = HBS_ZZ(z2, z1), zz3 = HBS_ZZ(z3, z2), zzn = HBS_ZZ(zn, z3);
    const uint32_t term = (zz0 & (z0 | o0)) | (zz1 & (z1 | o1)) | (zz2 & (z2 | o2)) | (zz3 & (z3 | o3)) | (zzn & (zn | on) & 0x00008080u);
    const uint32_t epb = (uint32_t)__builtin_popcount(zz0 & t0) + (uint32_t)__builtin_popcount(zz1 & t1) +
                         (uint32_t)__builtin_popcount(zz2 & t2) + (uint32_t)__builtin_popcount(zz3 & t3);
    kept = 16u - epb;
    uint32_t bad = term;
    if (kErrToo) {
        /* what sets HBS_ST_ERROR (elem_walk: m.err): 00 00 02, or 00 00 03 followed by a byte above 3 -- kErrToo: such a row is
         * not "quick" either (the index-only emit pass skips quick rows altogether: no terminator, no entry; no error, no mark) */
        const uint32_t w0 = zero_bytes(qc.x ^ 0x02020202u), w1 = zero_bytes(qc.y ^ 0x02020202u), w2 = zero_bytes(qc.z ^ 0x02020202u),
                       w3 = zero_bytes(qc.w ^ 0x02020202u);
        /* 0x80 where the NEXT byte is above 3 */
        const uint32_t g0 = ~zero_bytes(qc.x & 0xFCFCFCFCu) & 0x80808080u, g1 = ~zero_bytes(qc.y & 0xFCFCFCFCu) & 0x80808080u,
                       g2 = ~zero_bytes(qc.z & 0xFCFCFCFCu) & 0x80808080u, g3 = ~zero_bytes(qc.w & 0xFCFCFCFCu) & 0x80808080u,
                       gn = ~zero_bytes(xn & 0xFCFCFCFCu) & 0x80808080u;
        const uint32_t n0 = __builtin_amdgcn_alignbyte(g1, g0, 1), n1 = __builtin_amdgcn_alignbyte(g2, g1, 1),
                       n2 = __builtin_amdgcn_alignbyte(g3, g2, 1), n3 = __builtin_amdgcn_alignbyte(gn, g3, 1);
        bad |= (zz0 & (w0 | (t0 & n0))) | (zz1 & (w1 | (t1 & n1))) | (zz2 & (w2 | (t2 & n2))) | (zz3 & (w3 | (t3 & n3)));
    }
#undef HBS_ZZ
    return __ballot(bad != 0u) != 0ull;
}


__device__ __forceinline__ u32x4 dense_fetch(const uint8_t* src, uint64_t wseg, int r, int lane)
{
    return *reinterpret_cast<const u32x4*>(src + wseg + 1024ull * (uint64_t)r + 16ull * (uint64_t)lane);    /* the padded copy / the stream: always there */
}


} // namespace hbs
#endif
