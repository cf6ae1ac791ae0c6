/*
 * hbs_scan7.hip -- K12, event-sparse form, two half tiles per workgroup in flight.
 *
 * Same contract, tile algebra, descriptors, element code and workgroup shape as hbs_scan4.hip (4 fat wavefronts, 2
 * workgroups per CU, wavefront 0 does a tile's serial work; reference loop find_nal_unit + nal_to_rbsp,
 * h264_nal.c:38-76 / :147-200, driven as in hevc_analyze.c:135-177).  What differs is the schedule.  In hbs_scan4.hip a
 * tile's phases follow one another -- fetch, flags, elements, look-back, emit, copy -- and the phase table of round 2
 * (profiles/r02) says where the time goes: the look-back is 20 k of a tile's 89 k cycles, three quarters of it polls that
 * fail because the tiles in front publish their aggregates at the same moment this one asks for them, and the fetch (18 k)
 * cannot start before the copy has drained.  Here a wavefront's 48 rows are TWO sets of 24 (half tiles A and B, each a tile
 * of its own for ticket, descriptor and look-back), worked in the order
 *
 *      flags(A) elements(A)->publish   flags(B) elements(B)->publish   resolve(A) emit(A) copy(A) fetch(A')   resolve(B) emit(B) copy(B) fetch(B')
 *
 * so that between a half tile's publish and its resolve lies the other half's flag pass (the predecessors' aggregates have
 * arrived by then: one poll), and between its fetch and its flag pass lies the other half's resolve / emit / copy (its loads
 * fly meanwhile).
 */
#include <hip/hip_runtime.h>
#include <utility>
#include "hbs_wave.h"
#include "hbs_sparse.h"
#include "hbs_scan.h"
#include "hbs_elems.h"

namespace hbs {

constexpr int k7Rows          = 24;                      /* rows of 1 KiB per wavefront per half tile */
constexpr int k7Waves         = 4;
constexpr int k7Threads       = 64 * k7Waves;
constexpr int k7WaveBytes     = k7Rows * 1024;
constexpr int k7TileBytes     = k7Waves * k7WaveBytes;   /* 96 KiB */
constexpr int k7ChunksPerTile = k7TileBytes / kChunk;    /* 6144   */
constexpr int k7ParkRows      = 24;                      /* rows of wavefront 0 that wait in LDS while it runs the element code */
constexpr int k7TailLead      = 16;
constexpr uint32_t k7DenseElems = 64;                    /* past one batch of elements a half tile is walked by rows, all four wavefronts (hbs_scan4.hip: dense tiles) */
static_assert(k7TileBytes >= kTileBytes, "the descriptor workspace is sized for kTileBytes tiles");
static_assert(k7TailLead + k7TileBytes + 64 <= k4TailBytes, "the padded last-tile copy shares hbs_scan4's workspace");
constexpr uint32_t k7NoTile = 0xFFFFFFFFu;
constexpr int kDepCap = 64;

template <class F, int... Is>
__device__ __forceinline__ void rows7_apply(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void for_n(F&& f) { rows7_apply(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

/* one half tile's share of a wavefront */
struct Half7 {
    u32x4 q[k7Rows];
    uint32_t before, before2, after;
    uint32_t fm_lo, fm_hi;             /* lane r: flag mask of row r                         */
    uint32_t local_pre;                /* lane r: elements of my rows in front of row r      */
    uint64_t rowmask;
    uint32_t wb1, wb2, wb3, nflag, wave_base;
    uint32_t tile;                     /* k7NoTile: no more tiles for this set              */
    uint32_t dense;
};

struct LdsHalf7 {
    uint32_t wave_tot[k7Waves];
    uint16_t list[k7ChunksPerTile];
    uint32_t seg[k4ElemPass + 1];
    Deposit dep[k7Waves][kDepCap];
    unsigned long long ex_kept, ex_nals;
    uint32_t ex_inside, ex_ok;
    TileAgg tagg;                      /* the tile's aggregate, from its publish to its resolve */
    TileAgg wagg[k7Waves];             /* dense half tiles: per-wavefront aggregates            */
};
struct Lds7 {
    LdsHalf7 h[2];
    u32x4 park[k7ParkRows][64];
    uint32_t ticket[2];
};

struct Geo7 { uint64_t base, tile_end, wseg; const uint8_t* src; bool last_tile; };
__device__ __forceinline__ Geo7 geo7(uint32_t tile, int wv, const uint8_t* __restrict__ stream, const uint8_t* __restrict__ tail, uint64_t num_tiles)
{
    Geo7 g;
    g.base = (uint64_t)tile * (uint64_t)k7TileBytes;
    g.tile_end = g.base + (uint64_t)k7TileBytes;
    g.wseg = g.base + (uint64_t)(wv * k7WaveBytes);
    g.last_tile = (uint64_t)tile == num_tiles - 1;
    g.src = g.last_tile ? reinterpret_cast<const uint8_t*>(reinterpret_cast<uintptr_t>(tail) + (uintptr_t)k7TailLead - (uintptr_t)g.base) : stream;
    return g;
}

__device__ __forceinline__ void half_fetch(Half7& H, const Geo7& g, const uint8_t* __restrict__ stream, uint64_t n, int wv, int lane)
{
    const u32x4* p = reinterpret_cast<const u32x4*>(g.src + g.wseg) + lane;
#pragma unroll
    for (int r = 0; r < k7Rows; ++r) H.q[r] = stream_load16(p + r * 64);
    H.before = (g.wseg >= 4) ? stream_load4(g.src + g.wseg - 4) : 0xFFFFFFFFu;
    H.before2 = (g.wseg >= 8) ? stream_load4(g.src + g.wseg - 8) : 0xFFFFFFFFu;
    H.after = (g.last_tile || wv != k7Waves - 1) ? stream_load4(g.src + g.wseg + k7WaveBytes)
                                                 : load_dword_guarded(stream, (int64_t)(g.wseg + k7WaveBytes), n);
}

/* flag masks of my rows, four rows per branch (hbs_scan4.hip); deposits for the element thread */
__device__ __forceinline__ uint32_t half_flags(Half7& H, LdsHalf7& L, const Geo7& g, uint64_t n, int wv, int lane)
{
    uint32_t fm_lo = 0, fm_hi = 0, wslot = 0;
    for_n<k7Rows / 4>([&](auto gc) {
        constexpr int g0 = 4 * decltype(gc)::value;
        uint32_t xp[4], xn[4];
        bool f[4];
        uint64_t fmask[4];
        for_n<4>([&](auto kc) {
            constexpr int k = decltype(kc)::value, r = g0 + k;
            const uint32_t e_prev = (r == 0) ? H.before : (uint32_t)__builtin_amdgcn_readlane((int)H.q[r ? r - 1 : 0].w, 63);
            const uint32_t e_next = (r == k7Rows - 1) ? H.after : (uint32_t)__builtin_amdgcn_readlane((int)H.q[r + 1 < k7Rows ? r + 1 : r].x, 0);
            xp[k] = from_prev_lane(H.q[r].w, e_prev);
            xn[k] = from_next_lane(H.q[r].x, e_next);
            f[k] = chunk_flag(xp[k], H.q[r].x, H.q[r].y, H.q[r].z, H.q[r].w, xn[k]);
            fmask[k] = __ballot(f[k]);
        });
        if ((fmask[0] | fmask[1] | fmask[2] | fmask[3]) != 0) {
            for_n<4>([&](auto kc) {
                constexpr int k = decltype(kc)::value, r = g0 + k;
                if (fmask[k] != 0) {
                    write_lane_c<r>(fm_lo, (uint32_t)fmask[k]);
                    write_lane_c<r>(fm_hi, (uint32_t)(fmask[k] >> 32));
                    const uint32_t e_prev_z = (r == 0) ? H.before2 : (uint32_t)__builtin_amdgcn_readlane((int)H.q[r ? r - 1 : 0].z, 63);
                    const uint32_t xpp = from_prev_lane(H.q[r].z, e_prev_z);
                    const uint32_t slot = wslot + lanes_below(fmask[k]);
                    if (f[k] && slot < (uint32_t)kDepCap) {
                        Deposit d;
                        d.xpp = xpp; d.xp = xp[k]; d.x0 = H.q[r].x; d.x1 = H.q[r].y; d.x2 = H.q[r].z; d.x3 = H.q[r].w; d.xn = xn[k];
                        d.chunk = (uint32_t)(64 * (k7Rows * wv + r) + lane);
                        L.dep[wv][slot] = d;
                    }
                    wslot += (uint32_t)__builtin_popcountll(fmask[k]);
                }
            });
        }
    });
    if (g.last_tile && (n & 15ull) != 0 && n > g.wseg && n < g.wseg + (uint64_t)k7WaveBytes) {
        /* the chunk cut by the stream end is always an element */
        const uint32_t cut = (uint32_t)(n - g.wseg) >> 4;
        const int cr = (int)(cut >> 6), cl = (int)(cut & 63u);
        const uint64_t have = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)fm_hi, cr) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)fm_lo, cr);
        if (!((have >> cl) & 1ull)) {
            if (lane == 0 && wslot < (uint32_t)kDepCap) L.dep[wv][wslot].chunk = 0xFFFFFFFFu;
            if (lane == cr) { if (cl < 32) fm_lo |= 1u << cl; else fm_hi |= 1u << (cl - 32); }
        }
    }
    H.fm_lo = fm_lo; H.fm_hi = fm_hi;
    const uint32_t cnt = (lane < k7Rows) ? (uint32_t)__builtin_popcount(fm_lo) + (uint32_t)__builtin_popcount(fm_hi) : 0u;
    const uint32_t inc = wave_incl_scan32(cnt, lane);
    H.local_pre = inc - cnt;
    H.rowmask = __ballot(cnt != 0u);
    return (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
}

/* element i of the half tile (lane = i mod 64 of wavefront 0) */
__device__ __forceinline__ TileAgg elem_make7(Elem& el, const LdsHalf7& L, const Half7& H, uint32_t i, const Geo7& g, uint64_t n)
{
    const uint32_t c = L.list[i];
    const uint64_t prev_end = (i > 0) ? g.base + 16ull * ((uint32_t)L.list[i - 1] + 1u) : g.base;
    const uint32_t ew = (i >= H.wb1 ? 1u : 0u) + (i >= H.wb2 ? 1u : 0u) + (i >= H.wb3 ? 1u : 0u);
    const uint32_t ej = i - (ew == 0u ? 0u : ew == 1u ? H.wb1 : ew == 2u ? H.wb2 : H.wb3);
    const bool have_dep = ej < (uint32_t)kDepCap && L.dep[ew][ej < (uint32_t)kDepCap ? ej : 0u].chunk == c;
    if (have_dep) {
        const Deposit& d = L.dep[ew][ej];
        el.v.xpp = d.xpp; el.v.xp = d.xp; el.v.x0 = d.x0; el.v.x1 = d.x1; el.v.x2 = d.x2; el.v.x3 = d.x3; el.v.xn = d.xn;
        el.v.stream = g.src; el.v.g0 = g.base + 16ull * c; el.v.n = n;
    } else {
        elem_load(el.v, g.src, g.base + 16ull * c, n, g.last_tile);
    }
    elem_walk(el.v, el.m, el.s, el.cls);
    el.gap = span_bytes(prev_end, el.v.g0, n);
    el.chunk = c;
    return elem_agg(el.gap, el.s);
}

__device__ __forceinline__ void park7(Lds7& l, const Half7& H, int lane)
{
#pragma unroll
    for (int i = 0; i < k7ParkRows; ++i) l.park[i][lane] = H.q[k7Rows - k7ParkRows + i];
}
__device__ __forceinline__ void unpark7(const Lds7& l, Half7& H, int lane)
{
#pragma unroll
    for (int i = 0; i < k7ParkRows; ++i) H.q[k7Rows - k7ParkRows + i] = l.park[i][lane];
}

/* rows of a dense half tile: aggregate / emission, as hbs_scan4.hip's dense tiles (rows read again, a rolled loop) */
__device__ __noinline__ TileAgg dense_aggregate7(const Geo7 g, uint64_t n, uint32_t before, uint32_t before2, uint32_t after, uint32_t chunk0, int lane)
{
    TileAgg acc = agg_identity();
    u32x4 qp = dense_fetch(g.src, g.wseg, 0, lane), qc = qp, qn;
#pragma unroll 1
    for (int r = 0; r < k7Rows; ++r) {
        qn = dense_fetch(g.src, g.wseg, r + 1 < k7Rows ? r + 1 : r, lane);
        DenseRow d;
        dense_row(d, qp, qc, qn, r, k7Rows, before, before2, after, g.src, g.wseg, n, chunk0, lane);
        if (!d.row_has_event) {
            acc = combine(acc, gap_agg(wave_sum32(d.el.s.carry)));
        } else {
            const TileAgg ea = wave_scan_combine(elem_agg(0u, d.el.s), lane);
            acc = combine(acc, agg_readlane(ea, 63));
        }
        qp = qc; qc = qn;
    }
    return acc;
}
__device__ __noinline__ void dense_emit7(const Geo7 g, uint64_t n, uint32_t before, uint32_t before2, uint32_t after, uint32_t chunk0, int lane,
                                            TileAgg acc0, const Prefix excl, bool can_store, uint8_t* out, const EmitTarget tgt, uint32_t* scratch_word)
{
    TileAgg acc = acc0;
    u32x4 qp = dense_fetch(g.src, g.wseg, 0, lane), qc = qp, qn;
#pragma unroll 1
    for (int r = 0; r < k7Rows; ++r) {
        qn = dense_fetch(g.src, g.wseg, r + 1 < k7Rows ? r + 1 : r, lane);
        DenseRow d;
        dense_row(d, qp, qc, qn, r, k7Rows, before, before2, after, g.src, g.wseg, n, chunk0, lane);
        const TileAgg ea = wave_scan_combine(elem_agg(0u, d.el.s), lane);
        TileAgg up = agg_shfl_up(ea, 1);
        if (lane == 0) up = agg_identity();
        const TileAgg e = combine(acc, up);
        acc = combine(acc, agg_readlane(ea, 63));
        if (d.el.v.g0 < n) elem_emit(d.el, e, excl, can_store, out, tgt, scratch_word);
        qp = qc; qc = qn;
    }
}

/* ---- flags + elements -> the half tile's aggregate goes out ---------------------------------------------- */
__device__ __forceinline__ void half_flags_elements_publish(Half7& H, Lds7& l, int s, const uint8_t* __restrict__ stream, const uint8_t* __restrict__ tail,
                                                            uint64_t n, uint64_t num_tiles, unsigned long long* __restrict__ desc, int wv, int lane)
{
    LdsHalf7& L = l.h[s];
    const Geo7 g = geo7(H.tile, wv, stream, tail, num_tiles);
    const uint32_t mine = half_flags(H, L, g, n, wv, lane);
    if (lane == 0) L.wave_tot[wv] = mine;
    __syncthreads();
    const uint32_t wt0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.wave_tot[0]), wt1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.wave_tot[1]);
    const uint32_t wt2 = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.wave_tot[2]), wt3 = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.wave_tot[3]);
    H.wb1 = wt0; H.wb2 = wt0 + wt1; H.wb3 = H.wb2 + wt2; H.nflag = H.wb3 + wt3;
    H.wave_base = (wv == 0) ? 0u : (wv == 1) ? H.wb1 : (wv == 2) ? H.wb2 : H.wb3;
#ifdef HBS7_NO_DENSE
    H.dense = 0u;
#else
    H.dense = H.nflag > k7DenseElems ? 1u : 0u;
#endif
    if (H.dense) {
        /* a dense half tile is read again by rows: its registers are free from here */
#pragma unroll
        for (int r = 0; r < k7Rows; ++r) H.q[r] = u32x4{0u, 0u, 0u, 0u};
        const TileAgg wa = dense_aggregate7(g, n, H.before, H.before2, H.after, (uint32_t)(64 * k7Rows * wv), lane);
        if (lane == 0) L.wagg[wv] = wa;
        __syncthreads();
        if (wv == 0) {
            TileAgg t = agg_identity();
#pragma unroll
            for (int w = 0; w < k7Waves; ++w) t = combine(t, L.wagg[w]);
            look_back_publish(desc, (uint64_t)H.tile, t, lane);
            if (lane == 0) L.tagg = t;
        }
        return;
    }
    /* the chunk list, in stream order */
    for (uint64_t rm = H.rowmask; rm != 0ull; rm &= rm - 1ull) {
        const int r = __builtin_ctzll(rm);
        const uint32_t rp = H.wave_base + (uint32_t)__builtin_amdgcn_readlane((int)H.local_pre, r);
        const uint64_t f = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)H.fm_hi, r) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)H.fm_lo, r);
        if ((f >> lane) & 1ull) L.list[rp + lanes_below(f)] = (uint16_t)(64 * (k7Rows * wv + r) + lane);
    }
    __syncthreads();
    if (wv == 0) {
        /* the element code needs ~100 registers of its own: part of this half's rows wait in LDS (the other half's loads may be in flight) */
        park7(l, H, lane);
        Elem el;
        TileAgg ea = agg_identity();
        if ((uint32_t)lane < H.nflag) ea = elem_make7(el, L, H, (uint32_t)lane, g, n);
        ea = wave_scan_combine(ea, lane);
        const TileAgg acc = agg_readlane(ea, 63);
        const uint64_t last_end = (H.nflag > 0) ? g.base + 16ull * ((uint32_t)L.list[H.nflag - 1] + 1u) : g.base;
        const TileAgg t = combine(acc, gap_agg(span_bytes(last_end, g.tile_end, n)));
        look_back_publish(desc, (uint64_t)H.tile, t, lane);
        if (lane == 0) L.tagg = t;
        unpark7(l, H, lane);
    }
}

/* ---- resolve, the elements' second half, copy.  false: a look-back timed out, the workgroup gives up -------- */
__device__ __forceinline__ bool half_resolve_emit_copy(Half7& H, Lds7& l, int s, const uint8_t* __restrict__ stream, const uint8_t* __restrict__ tail,
                                                       uint64_t n, uint64_t num_tiles, uint8_t* __restrict__ rbsp, uint64_t rbsp_cap,
                                                       unsigned long long* __restrict__ desc, RunHeader* __restrict__ hdr, const EmitTarget& tgt, int wv, int lane)
{
    LdsHalf7& L = l.h[s];
    const Geo7 g = geo7(H.tile, wv, stream, tail, num_tiles);
    if (wv == 0) {
        TileAgg tagg;
        tagg.cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.tagg.cnt); tagg.known = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.tagg.known);
        tagg.sig = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.tagg.sig); tagg.last = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.tagg.last);
        Prefix ex;
        uint32_t it, stl;
        const bool ok = look_back_resolve(desc, (uint64_t)H.tile, tagg, hdr, lane, ex, it, stl);
        const uint32_t tile_kept = tagg.known + (ex.inside ? tagg.sig : 0u);
        const bool can = rbsp != nullptr && ex.kept + tile_kept <= rbsp_cap;
        if (lane == 0) {
            L.ex_kept = ex.kept; L.ex_nals = ex.nals; L.ex_inside = ex.inside;
            L.ex_ok = !ok ? 0u : (rbsp != nullptr && !can) ? 2u : 1u;
            L.seg[0] = seg_pack(-1, 0u, ex.inside != 0u);
            if (ok && rbsp != nullptr && !can) atomicMax(&hdr->error, (uint32_t)(-HBS_E_CAPACITY));
            if (ok && g.last_tile) {
                const Prefix incl = fold(ex, tagg);
                hdr->final_kept = incl.kept; hdr->final_nals = incl.nals; hdr->final_inside = incl.inside;
            }
        }
        /* the first batch of elements (nearly always the only one) follows at once: one barrier for both */
        if (ok && !H.dense && H.nflag != 0u) {
            park7(l, H, lane);
            Elem el;
            const uint32_t i = (uint32_t)lane;
            TileAgg ea = agg_identity();
            if (i < H.nflag) ea = elem_make7(el, L, H, i, g, n);
            ea = wave_scan_combine(ea, lane);
            TileAgg up = agg_shfl_up(ea, 1);
            if (lane == 0) up = agg_identity();
            if (i < H.nflag) elem_emit(el, up, ex, can, rbsp + ex.kept, tgt, &L.seg[lane + 1]);
            unpark7(l, H, lane);
        }
    }
    __syncthreads();
    if (L.ex_ok == 0u) return false;
    Prefix excl;
    {
        Prefix ex;
        ex.kept = L.ex_kept; ex.nals = L.ex_nals; ex.inside = L.ex_inside;
        excl = prefix_uniform4(ex);
    }
    const bool can_store = rbsp != nullptr && L.ex_ok == 1u;
    uint8_t* const out = rbsp + excl.kept;
    if (H.dense) {
#pragma unroll
        for (int r = 0; r < k7Rows; ++r) H.q[r] = u32x4{0u, 0u, 0u, 0u};     /* not live through this branch */
        TileAgg before_me = agg_identity();
#pragma unroll
        for (int w = 0; w < k7Waves; ++w) if (w < wv) before_me = combine(before_me, L.wagg[w]);
        dense_emit7(g, n, H.before, H.before2, H.after, (uint32_t)(64 * k7Rows * wv), lane, before_me, excl, can_store, out, tgt,
                    &L.dep[wv][lane & (kDepCap - 1)].xpp);
        return true;
    }
    const uint32_t whole = (uint32_t)(span_bytes(g.base, g.tile_end, n) >> 4);        /* chunks of the half tile that are complete */
    if (can_store) {
        const uint32_t cc0 = (uint32_t)(64 * k7Rows * wv + lane);
        /* lane j: segment word j (j = 0..63), word 64 apart: a row without elements needs one word, picked with a readlane */
        const uint32_t segv = L.seg[lane];
        const uint32_t seg64 = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.seg[k4ElemPass]);
        for_n<k7Rows>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            const uint32_t cc = cc0 + 64u * r;
            const uint32_t rowpre = H.wave_base + (uint32_t)__builtin_amdgcn_readlane((int)H.local_pre, r);
            if (!((H.rowmask >> r) & 1ull)) {          /* no element in this row: one word for all lanes */
                const uint32_t w = (rowpre == (uint32_t)k4ElemPass) ? seg64 : (uint32_t)__builtin_amdgcn_readlane((int)segv, (int)(rowpre & 63u));
                if (seg_inside(w) && cc < whole) arena_store16(out + (int64_t)seg_bias(w) + 16u * cc, H.q[r]);
            } else {
                const uint64_t f = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)H.fm_hi, r) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)H.fm_lo, r);
                const uint32_t k = rowpre + lanes_below(f);
                if (!((f >> lane) & 1ull) && cc < whole) {
                    const uint32_t w = L.seg[k];
                    if (seg_inside(w)) arena_store16(out + (int64_t)(seg_bias(w) + (int32_t)(16u * cc)), H.q[r]);
                }
            }
        });
    }
    return true;
}

/* the set's next tile: by ticket, when the copy has been issued (a ticket held by a workgroup that has not started the tile is
 * what the look-backs behind it wait for); then its loads */
__device__ __forceinline__ void half_next(Half7& H, Lds7& l, int s, RunHeader* __restrict__ hdr, const uint8_t* __restrict__ stream, const uint8_t* __restrict__ tail,
                                          uint64_t n, uint64_t num_tiles, int wv, int lane)
{
    if (threadIdx.x == 0) l.ticket[s] = atomicAdd(&hdr->ticket, 1u);
    __syncthreads();
    const uint32_t t = (uint32_t)__builtin_amdgcn_readfirstlane((int)l.ticket[s]);
    H.tile = ((uint64_t)t < num_tiles) ? t : k7NoTile;
    if (H.tile != k7NoTile) half_fetch(H, geo7(H.tile, wv, stream, tail, num_tiles), stream, n, wv, lane);
}

__global__ __launch_bounds__(k7Threads, 2)
void k_scan_extract7(const uint8_t* __restrict__ stream, uint64_t n, uint64_t num_tiles,
                     hbs_nal_entry* __restrict__ index, uint64_t index_cap,
                     uint8_t* __restrict__ rbsp, uint64_t rbsp_cap,
                     unsigned long long* __restrict__ desc, RunHeader* __restrict__ hdr, const uint8_t* __restrict__ tail,
                     int gate)
{
    if (gate == kGateIfSparse && probe_dense_dev(hdr)) return;
    __shared__ Lds7 l;
    const int tid0 = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    EmitTarget tgt;
    tgt.index = index; tgt.index_cap = index_cap; tgt.hdr = hdr;
    Half7 A, B;
    int lane = launder_lane(tid0) & 63;
    half_next(A, l, 0, hdr, stream, tail, n, num_tiles, wv, lane);
    half_next(B, l, 1, hdr, stream, tail, n, num_tiles, wv, lane);
    while (A.tile != k7NoTile || B.tile != k7NoTile) {
        lane = launder_lane(tid0) & 63;
        __builtin_amdgcn_s_setprio(3);          /* until the aggregates are out, this workgroup is what its successors wait for */
        if (A.tile != k7NoTile) half_flags_elements_publish(A, l, 0, stream, tail, n, num_tiles, desc, wv, lane);
        if (B.tile != k7NoTile) half_flags_elements_publish(B, l, 1, stream, tail, n, num_tiles, desc, wv, lane);
        __builtin_amdgcn_s_setprio(0);
        lane = launder_lane(tid0) & 63;
        if (A.tile != k7NoTile) {
            if (!half_resolve_emit_copy(A, l, 0, stream, tail, n, num_tiles, rbsp, rbsp_cap, desc, hdr, tgt, wv, lane)) return;
            half_next(A, l, 0, hdr, stream, tail, n, num_tiles, wv, lane);
        }
        if (B.tile != k7NoTile) {
            if (!half_resolve_emit_copy(B, l, 1, stream, tail, n, num_tiles, rbsp, rbsp_cap, desc, hdr, tgt, wv, lane)) return;
            half_next(B, l, 1, hdr, stream, tail, n, num_tiles, wv, lane);
        }
    }
}

int scan7_tile_bytes() { return k7TileBytes; }

int scan7_grid_blocks(int device, int* blocks_per_cu_out)
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return -1;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_scan_extract7, k7Threads, 0) != hipSuccess) return -1;
    if (per_cu < 1) per_cu = 1;
    if (blocks_per_cu_out) *blocks_per_cu_out = per_cu;
    return prop.multiProcessorCount * per_cu;
}

void launch_scan_extract7_kernel(const ScanArgs& a, uint64_t num_tiles, int gate, hipStream_t st)
{
    uint64_t grid = (uint64_t)a.grid_blocks7;
    if (grid > (num_tiles + 1) / 2) grid = (num_tiles + 1) / 2;      /* every workgroup takes two tiles before anything else */
    if (grid < 1) grid = 1;
    k_scan_extract7<<<dim3((unsigned)grid), dim3(k7Threads), 0, st>>>(
        a.stream, a.n, num_tiles, a.index, a.index_cap, a.rbsp, a.rbsp_cap, a.desc, a.hdr, a.tail, gate);
}

} // namespace hbs
