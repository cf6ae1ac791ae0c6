/*
 * hbs_scan.hip -- K12: fused start-code scan + NAL index + RBSP extraction for
 * gfx950 (MI355X), one pass over the stream.
 *
 * Replaces the reference's sequential loop
 *     while (find_nal_unit(p, sz, &s, &e) > 0) { nal_to_rbsp(p+s, ...); p += e; }
 * (hevc_analyze.c:135-177 driving h264_nal.c:38-76 and :147-200) with a
 * chained scan:
 *
 *   - persistent workgroups (512 threads = 8 wave64, 2 per CU), tile i ->
 *     workgroup i mod grid; a tile is 64 KiB fetched with coalesced 16-byte
 *     loads one tile ahead (registers) and staged in a swizzled LDS image;
 *   - each thread classifies 128 contiguous bytes (hbs_tile.h), a workgroup
 *     scan turns that into a tile aggregate {NAL starts, kept bytes as a
 *     function of the carried inside/outside state, state after the tile};
 *   - decoupled look-back over per-tile descriptors (two self-validating
 *     8-byte words, agent-scope relaxed atomics: the data is the flag, so no
 *     fences) yields the exclusive prefix {arena offset, NAL ordinal, state};
 *   - the tile's kept bytes are gathered from LDS into 16-byte words aligned in
 *     the arena and stored coalesced; index entries are scattered by the
 *     threads that own the start / end events.
 *
 * HBM traffic: stream read once (1 B/B), RBSP written once (~1 B/B), index
 * 32 B/NAL, descriptors 16 B per 64 KiB tile.  No MFMA: byte scan, HBM-bound.
 */
#include <hip/hip_runtime.h>
#include "hbs_tile.h"
#include "hbs_scan.h"

#ifndef HBS2_COPY_DEPTH
#define HBS2_COPY_DEPTH -1
#endif

namespace hbs {

/* Diagnostic build only (-DHBS_PHASE_TIMING, tests/tools/phase_timing.py): per-phase
 * shader-clock sums of every workgroup, never part of the shipped library. */
#ifdef HBS_PHASE_TIMING
__device__ unsigned long long g_phase_cycles[1024][8];
#define HBS_T_DECL unsigned long long t_prev = __builtin_amdgcn_s_memtime(), t_acc[8] = {0,0,0,0,0,0,0,0};
#define HBS_T_MARK(i) { const unsigned long long t_now = __builtin_amdgcn_s_memtime(); t_acc[i] += t_now - t_prev; t_prev = t_now; }
#define HBS_T_COUNT(i, v) { t_acc[i] += (v); }
#define HBS_T_FLUSH if (threadIdx.x == 0 && blockIdx.x < 1024) { for (int i = 0; i < 8; ++i) g_phase_cycles[blockIdx.x][i] = t_acc[i]; }
/* instruction-count experiments: drop the rest of the tile after phase i (results are then wrong) */
__device__ int g_stop_after = 99;
#define HBS_T_STOP(i) if (g_stop_after == (i)) break;
#else
#define HBS_T_STOP(i)
#define HBS_T_DECL
#define HBS_T_MARK(i)
#define HBS_T_COUNT(i, v)
#define HBS_T_FLUSH
#endif

constexpr int kSlowCap = 256;        /* deferred general-path words per tile (overflow is handled inline) */

/* one wave's view of its 64-tile look-back window */
struct WaveSlot {
    uint32_t status;                 /* 0 some needed tile not ready, 1 all 64 aggregates, 2 prefix at lane lstar */
    uint32_t abort;
    TileAgg win;                     /* aggregate of the lanes in front of the prefix (or of all 64) */
    uint64_t pre_kept, pre_nals;     /* the prefix found (status 2)                */
    uint32_t pre_inside, pad;
};

struct TileLds {
    alignas(16) uint8_t img[kImageBytes];      /* swizzled tile image (TileView)      */
    uint64_t keep[kBlocks + 1];                /* pass 1: pattern masks; pass 2: keep masks per 64-byte block */
    uint32_t rank[kBlocks + 1];                /* tile rank of each block's first kept byte */
    uint16_t slow[kSlowCap];                   /* chunks with holes, compacted after the whole ones */
    uint32_t slow_cnt;
    uint32_t wave_last[kWaves];
    uint32_t wave_cnt[kWaves];
    uint32_t wave_known[kWaves];
    uint32_t wave_sig[kWaves];
    WaveSlot lb[2][4];
    uint32_t ticket;                           /* tile number handed to this workgroup (dynamic schedules) */
};

/* 16 stream bytes at offset g (may straddle or exceed [0,n)): 0xFF outside */
__device__ __forceinline__ uint4 load16_edge(const uint8_t* __restrict__ s, int64_t g, uint64_t n)
{
    uint32_t w0 = 0xFFFFFFFFu, w1 = 0xFFFFFFFFu, w2 = 0xFFFFFFFFu, w3 = 0xFFFFFFFFu;
#pragma unroll 1
    for (int b = 0; b < 16; ++b) {
        const int64_t q = g + b;
        if (q >= 0 && (uint64_t)q < n) {
            const uint32_t m = ~(0xFFu << (8 * (b & 3)));
            const uint32_t v = (uint32_t)s[q] << (8 * (b & 3));
            if ((b >> 2) == 0) w0 = (w0 & m) | v;
            else if ((b >> 2) == 1) w1 = (w1 & m) | v;
            else if ((b >> 2) == 2) w2 = (w2 & m) | v;
            else w3 = (w3 & m) | v;
        }
    }
    return make_uint4(w0, w1, w2, w3);
}

__device__ __forceinline__ uint4 load16(const uint8_t* __restrict__ s, int64_t g, uint64_t n)
{
    if (g >= 0 && (uint64_t)g + 16 <= n) return *reinterpret_cast<const uint4*>(s + g);
    return load16_edge(s, g, n);
}

constexpr int kChunksPerThread = kThreadBytes / 16;     /* 16-byte loads per thread per tile */

/* The prefetched tile lives in named vector registers (an array here is not
 * promoted out of scratch by hipcc once its fill is conditional). */
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));    /* plain vector: loads/stores stay SSA values */
static_assert(kChunksPerThread == 8, "HBS_REPC lists the chunks of one thread");
#define HBS_REPC(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
struct TileRegs {
#define HBS_DECL(u) u32x4 r##u;
    HBS_REPC(HBS_DECL)
#undef HBS_DECL
};

/* coalesced fetch of one FULL tile into registers: lane-contiguous 16-byte chunks */
__device__ __forceinline__ void fetch_tile(TileRegs& t, const uint8_t* __restrict__ s, uint64_t tile_base, int tid)
{
    const u32x4* p = reinterpret_cast<const u32x4*>(s + tile_base) + tid;
#define HBS_LD(u) t.r##u = p[u * kThreads];
    HBS_REPC(HBS_LD)
#undef HBS_LD
}

/* registers -> swizzled LDS image */
__device__ __forceinline__ void stage_tile(TileLds& l, const TileRegs& t, int tid)
{
#define HBS_ST(u) *reinterpret_cast<u32x4*>(&l.img[TileView::phys(16 * (u * kThreads + tid))]) = t.r##u;
    HBS_REPC(HBS_ST)
#undef HBS_ST
}

/* the (single) partial tile at the end of the stream: straight to LDS, guarded */
__device__ __forceinline__ void stage_tile_edge(TileLds& l, const uint8_t* __restrict__ s, uint64_t tile_base, uint64_t n, int tid)
{
#pragma unroll 1
    for (int u = 0; u < kChunksPerThread; ++u) {
        const int c = u * kThreads + tid;
        *reinterpret_cast<uint4*>(&l.img[TileView::phys(16 * c)]) = load16(s, (int64_t)(tile_base + 16ull * (uint32_t)c), n);
    }
}

/* the 16 bytes either side of the tile */
__device__ __forceinline__ void stage_halo(TileLds& l, const uint8_t* __restrict__ s, uint64_t tile_base, uint64_t n, int tid)
{
    if (tid == 0)
        *reinterpret_cast<uint4*>(&l.img[TileView::phys(-16)]) = load16(s, (int64_t)tile_base - 16, n);
    if (tid == 64)
        *reinterpret_cast<uint4*>(&l.img[TileView::phys(kTileBytes)]) = load16(s, (int64_t)(tile_base + kTileBytes), n);
}

/* Opaque copy of a lane-constant value.  hipcc hoists every address that only
 * depends on threadIdx out of the tile loop and then spills them all around it;
 * laundering the thread id once per phase keeps those addresses phase-local. */
__device__ __forceinline__ int launder(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

enum : uint32_t { kInOutside = 0, kInInside = 1, kInCarry = 2 };

struct ThreadPrefix {
    uint32_t in_state;   /* kIn*: state at the thread's first byte          */
    uint32_t cnt;        /* NAL starts before the thread (tile-relative)    */
    uint32_t known;      /* kept bytes before the thread, state-independent */
    uint32_t sig;        /* kept bytes before the thread if the tile's carry-in is inside */
};

/* workgroup scan of the per-thread summaries -> exclusive prefix per thread + tile aggregate */
__device__ __forceinline__ ThreadPrefix block_scan(TileLds& l, const TileAgg& s, int tid, TileAgg& agg)
{
    const int lane = tid & 63, wv = tid >> 6;
    const uint64_t m_ev = __ballot(s.last != kKindNone);
    const uint64_t m_st = __ballot(s.last == kKindStart);
    const uint64_t lower = m_ev & ((1ull << lane) - 1ull);

    uint32_t in_state = kInCarry;
    if (lower != 0) in_state = (uint32_t)((m_st >> (63 - __builtin_clzll(lower))) & 1ull);
    if (lane == 0) {
        uint32_t wl = kKindNone;
        if (m_ev != 0) wl = ((m_st >> (63 - __builtin_clzll(m_ev))) & 1ull) ? kKindStart : kKindStop;
        l.wave_last[wv] = wl;
    }
    __syncthreads();

    uint32_t carry_kind = kKindNone;
    for (int w = wv - 1; w >= 0; --w) {
        const uint32_t k = l.wave_last[w];
        if (k != kKindNone) { carry_kind = k; break; }
    }
    if (in_state == kInCarry && carry_kind != kKindNone)
        in_state = (carry_kind == kKindStart) ? kInInside : kInOutside;

    const uint32_t known = s.known + (in_state == kInInside ? s.sig : 0u);
    const uint32_t sig = (in_state == kInCarry) ? s.sig : 0u;

    /* per-wave sums are <= 64 * kThreadBytes: scan known|sig packed, cnt separately */
    const uint32_t packed = known | (sig << 16);
    const uint32_t ip = wave_incl_scan(packed, lane);
    const uint32_t ic = wave_incl_scan(s.cnt, lane);
    if (lane == 63) {
        l.wave_known[wv] = ip & 0xFFFFu;
        l.wave_sig[wv] = ip >> 16;
        l.wave_cnt[wv] = ic;
    }
    __syncthreads();

    ThreadPrefix p;
    p.in_state = in_state;
    p.known = (ip & 0xFFFFu) - known;
    p.sig = (ip >> 16) - sig;
    p.cnt = ic - s.cnt;
    uint32_t tk = 0, ts = 0, tc = 0, tl = kKindNone;
#pragma unroll
    for (int w = 0; w < kWaves; ++w) {
        if (w < wv) { p.known += l.wave_known[w]; p.sig += l.wave_sig[w]; p.cnt += l.wave_cnt[w]; }
        tk += l.wave_known[w]; ts += l.wave_sig[w]; tc += l.wave_cnt[w];
        if (l.wave_last[w] != kKindNone) tl = l.wave_last[w];
    }
    agg.known = tk; agg.sig = ts; agg.cnt = tc; agg.last = tl;
    return p;
}

__device__ __forceinline__ uint64_t ld_desc(const unsigned long long* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_desc(unsigned long long* p, uint64_t v)
{
    __hip_atomic_store(p, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

/*
 * Aggregate of the tiles held by lanes [0, lstar) of a look-back window, lane l
 * holding tile (win_hi - l): higher lanes are EARLIER tiles.  Wave-parallel
 * form of folding combine() from lane lstar-1 down to lane 0.
 */
__device__ __forceinline__ TileAgg window_fold(const TileAgg& a, int lstar, int lane)
{
    const uint64_t need = (lstar >= 64) ? ~0ull : ((1ull << lstar) - 1ull);
    const bool mine = lane < lstar;
    const uint64_t m_ev = __ballot(mine && a.last != kKindNone) & need;
    const uint64_t m_st = __ballot(mine && a.last == kKindStart) & need;
    /* state in front of my tile = kind of the nearest earlier tile (higher lane) that has an event */
    const uint64_t above = (lane >= 63) ? 0ull : (m_ev & ~((2ull << lane) - 1ull));
    uint32_t st = kInCarry;
    if (above != 0) st = (uint32_t)((m_st >> __builtin_ctzll(above)) & 1ull);
    uint32_t k = 0, g = 0, c = 0;
    if (mine) {
        k = a.known + (st == kInInside ? a.sig : 0u);
        g = (st == kInCarry) ? a.sig : 0u;
        c = a.cnt;
    }
    TileAgg w;
    w.known = wave_sum(k);
    w.sig = wave_sum(g);
    w.cnt = wave_sum(c);
    w.last = kKindNone;
    if (m_ev != 0) w.last = ((m_st >> __builtin_ctzll(m_ev)) & 1ull) ? kKindStart : kKindStop;
    return w;
}

/*
 * Decoupled look-back over 256 predecessors per step: wave w inspects tiles
 * win_hi - 64w - lane, folds its window, and the four windows are chained
 * through LDS.  Every thread of the workgroup calls this (it contains
 * barriers) and gets the same answer.  desc[2*t], desc[2*t+1] are the two words
 * of tile t.  Returns false on timeout/abort.
 */
__device__ __forceinline__ bool look_back(TileLds& l, unsigned long long* desc, uint64_t tile, const TileAgg& mine,
                                          RunHeader* hdr, int tid, Prefix& excl, uint32_t& dbg_iters, uint32_t& dbg_stalls)
{
    dbg_iters = 0; dbg_stalls = 0;
    const int lane = tid & 63, wv = tid >> 6;
    bool ok = true;
    if (tile == 0) {
        excl.kept = 0; excl.nals = 0; excl.inside = 0;
    } else {
        if (tid == 0) {
            st_desc(&desc[2 * tile], pack_agg0(mine));
            st_desc(&desc[2 * tile + 1], pack_agg1(mine));
        }
        TileAgg acc = {0u, 0u, 0u, kKindNone};       /* tiles between the windows and `tile` */
        int64_t win_hi = (int64_t)tile - 1;           /* nearest tile of the current step      */
        uint32_t spins = 0, par = 0;
        for (;;) {
            const int64_t t = win_hi - (64 * wv + lane);
            uint64_t w0 = kDescPrefix, w1 = kDescPrefix;   /* virtual tile -1: empty prefix   */
            if (t >= 0 && wv < 4) {
                w0 = ld_desc(&desc[2 * t]);
                w1 = ld_desc(&desc[2 * t + 1]);
            }
            const uint32_t s0 = (uint32_t)(w0 & 3u), s1 = (uint32_t)(w1 & 3u);
            const bool ready = (s0 == s1) && (s0 != kDescEmpty);
            const bool is_pre = ready && (s0 == kDescPrefix);
            const uint64_t m_pre = __ballot(is_pre);
            const uint64_t m_ready = __ballot(ready);
            const int lstar = m_pre ? (int)__builtin_ctzll(m_pre) : 64;      /* nearest prefix in my window */
            const uint64_t need = (lstar >= 64) ? ~0ull : ((1ull << lstar) - 1ull);
            const bool win_ok = (m_ready & need) == need;
            const TileAgg win = window_fold(unpack_agg(w0, w1), lstar, lane);
            const Prefix p = unpack_pre(w0, w1);
            if (lane == (lstar & 63) && wv < 4) {
                WaveSlot& sl = l.lb[par][wv];
                sl.status = win_ok ? (lstar < 64 ? 2u : 1u) : 0u;
                /* the abort flag is one line shared by every workgroup: look at it only while stalled for long */
                sl.abort = (wv == 0 && (spins & 63u) == 63u) ? __hip_atomic_load(&hdr->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                sl.win = win;
                sl.pre_kept = p.kept; sl.pre_nals = p.nals; sl.pre_inside = p.inside;
            }
            __syncthreads();
            /* chain the windows, nearest first; identical in every thread */
            bool done = false, stall = false;
            TileAgg a2 = acc;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                if (done || stall) continue;
                const WaveSlot& sl = l.lb[par][w];
                if (sl.status == 0u) { stall = true; continue; }
                a2 = combine(sl.win, a2);
                if (sl.status == 2u) {
                    Prefix q;
                    q.kept = sl.pre_kept; q.nals = sl.pre_nals; q.inside = sl.pre_inside;
                    excl = fold(q, a2);
                    done = true;
                }
            }
            const bool aborted = l.lb[par][0].abort != 0u;
            par ^= 1u;
            ++dbg_iters;
            if (done) break;
            if (stall) {
                ++dbg_stalls;
                if (++spins > (1u << 20) || aborted) { ok = false; break; }
                __builtin_amdgcn_s_sleep(1);
                continue;
            }
            acc = a2;
            win_hi -= 256;
        }
    }
    if (tid == 0) {
        if (ok) {
            const Prefix incl = fold(excl, mine);
            st_desc(&desc[2 * tile], pack_pre0(incl));
            st_desc(&desc[2 * tile + 1], pack_pre1(incl));
        } else {
            __hip_atomic_store(&hdr->abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicMax(&hdr->error, (uint32_t)(-HBS_E_TIMEOUT));
        }
    }
    return ok;
}

/* 16 bytes to a byte-aligned destination (one global_store_dwordx4) */
struct __attribute__((packed, aligned(1))) Unaligned16 { u32x4 v; };
__device__ __forceinline__ void store16_unaligned(uint8_t* dst, const Quad& q)
{
    u32x4 v;
    v.x = q.x; v.y = q.y; v.z = q.z; v.w = q.w;
    reinterpret_cast<Unaligned16*>(dst)->v = v;
}

/* a chunk with removed bytes: compact, then 8/4/2/1-byte pieces */
__device__ __forceinline__ void store_holes(const TileView& v, uint8_t* out, uint32_t c, const ChunkDest& d)
{
    uint64_t lo, hi;
    uint32_t cnt = compact_chunk(v, c, d.sub, lo, hi);
    uint8_t* p = out + d.rank;
    struct __attribute__((packed, aligned(1))) U8 { uint64_t v; };
    struct __attribute__((packed, aligned(1))) U4 { uint32_t v; };
    struct __attribute__((packed, aligned(1))) U2 { uint16_t v; };
    if (cnt & 8u) { reinterpret_cast<U8*>(p)->v = lo; p += 8; lo = hi; }
    if (cnt & 4u) { reinterpret_cast<U4*>(p)->v = (uint32_t)lo; p += 4; lo >>= 32; }
    if (cnt & 2u) { reinterpret_cast<U2*>(p)->v = (uint16_t)lo; p += 2; lo >>= 16; }
    if (cnt & 1u) { *p = (uint8_t)lo; }
}

enum : int { kSchedStriped = 0, kSchedTicketTop = 1, kSchedTicketAfterPrefix = 2 };

/* next unclaimed tile, the same value in every thread (contains a barrier) */
__device__ __forceinline__ uint64_t take_ticket(TileLds& l, RunHeader* hdr, int tid)
{
    if (tid == 0) l.ticket = atomicAdd(&hdr->ticket, 1u);
    __syncthreads();
    return (uint64_t)l.ticket;
}

/* the tile loop of the LDS-image kernel; `first` = the workgroup's first tile under the striped schedule */
__device__ __forceinline__
void scan_tiles(TileLds& l, const uint8_t* __restrict__ stream, uint64_t n, uint64_t num_tiles,
                hbs_nal_entry* __restrict__ index, uint64_t index_cap,
                uint8_t* __restrict__ rbsp, uint64_t rbsp_cap,
                unsigned long long* __restrict__ desc, RunHeader* __restrict__ hdr, int sched)
{
    const int tid0 = threadIdx.x;
    EmitTarget tgt;
    tgt.index = index; tgt.index_cap = index_cap; tgt.hdr = hdr;
    TileView view;
    view.img = l.img;

    const uint64_t full_tiles = n / (uint64_t)kTileBytes;      /* tiles [0, full_tiles) are complete */
    TileRegs nxt;
    /* sched 0: tile i belongs to workgroup i mod grid; 1: tiles are handed out in
     * arrival order at the top of the loop; 2: the next tile is taken (and its
     * loads issued) once the current one has its prefix */
    uint64_t tile = (sched == kSchedStriped) ? (uint64_t)blockIdx.x : take_ticket(l, hdr, tid0);
    if (sched != kSchedTicketTop && tile < full_tiles) fetch_tile(nxt, stream, tile * (uint64_t)kTileBytes, tid0);
    HBS_T_DECL

    while (tile < num_tiles) {
        const uint64_t tile_base = tile * (uint64_t)kTileBytes;
        if (sched == kSchedTicketTop && tile < full_tiles) fetch_tile(nxt, stream, tile_base, launder(tid0));
        uint64_t next_tile = tile + gridDim.x;
        do {
        {
            const int t0 = launder(tid0);
            if (tile < full_tiles) stage_tile(l, nxt, t0);
            else stage_tile_edge(l, stream, tile_base, n, t0);
            stage_halo(l, stream, tile_base, n, t0);
        }
        __syncthreads();
        HBS_T_MARK(0)
        HBS_T_STOP(0)
        const int tid = launder(tid0);

        const int32_t o0 = kThreadBytes * tid;
        const uint64_t g0 = tile_base + (uint64_t)o0;
        const TileAgg mine = classify_thread(view, o0, g0, n, l.keep, tid);
        HBS_T_MARK(1)
        HBS_T_STOP(1)

        TileAgg agg;
        const ThreadPrefix tp = block_scan(l, mine, tid, agg);
        HBS_T_MARK(2)
        HBS_T_STOP(2)

        Prefix excl;
        uint32_t lb_iters, lb_stalls;
        if (!look_back(l, desc, tile, agg, hdr, tid, excl, lb_iters, lb_stalls)) return;
        HBS_T_MARK(3)
        HBS_T_STOP(3)
        HBS_T_COUNT(7, ((unsigned long long)lb_stalls << 32) | lb_iters)
        if (tid == 0 && tile == num_tiles - 1) {
            const Prefix incl = fold(excl, agg);
            hdr->final_kept = incl.kept; hdr->final_nals = incl.nals; hdr->final_inside = incl.inside;
        }
        /* Next tile's HBM reads fly under the emit and the gather.  They are issued
         * behind the look-back because loads return in order: a descriptor read
         * queued behind 16 tile loads would wait for all of them. */
        if (sched == kSchedTicketAfterPrefix) next_tile = take_ticket(l, hdr, launder(tid0));
        if (sched != kSchedTicketTop && next_tile < full_tiles) fetch_tile(nxt, stream, next_tile * (uint64_t)kTileBytes, launder(tid0));

        const uint64_t ex_kept = excl.kept;
        const uint32_t tile_kept = agg.known + (excl.inside ? agg.sig : 0u);
        ThreadStart ts;
        ts.in_state = tp.in_state; ts.cnt = tp.cnt; ts.known = tp.known; ts.sig = tp.sig;
        {
            const int te = launder(tid0);
            emit_thread(view, kThreadBytes * te, tile_base + (uint64_t)(kThreadBytes * te), n, te, ts, excl, l.keep, l.rank, tgt);
        }
        if (tid == 0) l.slow_cnt = 0;
        __syncthreads();
        HBS_T_MARK(4)
        HBS_T_STOP(4)

        if (rbsp != nullptr && tile_kept != 0) {
            if (ex_kept + tile_kept <= rbsp_cap) {
                uint8_t* out = rbsp + ex_kept;
                /* whole chunks now (lane-contiguous: coalesced image reads and arena
                 * stores); the few with holes are queued and done together so that
                 * they do not serialise whole waves */
                const uint32_t c0 = (uint32_t)launder(tid0);
#pragma unroll 2
                for (uint32_t c = c0; c < (uint32_t)(kTileBytes / 16); c += kThreads) {
                    const ChunkDest d = chunk_dest(l.rank, l.keep, c);
                    if (d.sub == 0xFFFFu) {
                        const Quad qd = view.quad((int32_t)(16u * c));
                        store16_unaligned(out + d.rank, qd);
#if HBS2_COPY_DEPTH >= 0
                        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(HBS2_COPY_DEPTH) : "memory");     /* a short memory queue on the CU: hbs_scan4.hip, round 3 */
#endif
                    } else if (d.sub != 0u) {
                        const uint32_t slot = atomicAdd(&l.slow_cnt, 1u);
                        if (slot < (uint32_t)kSlowCap) l.slow[slot] = (uint16_t)c;
                        else store_holes(view, out, c, d);
                    }
                }
                __syncthreads();
                HBS_T_MARK(5)
                const uint32_t nslow = l.slow_cnt < (uint32_t)kSlowCap ? l.slow_cnt : (uint32_t)kSlowCap;
                for (uint32_t i = (uint32_t)launder(tid0); i < nslow; i += kThreads) {
                    const uint32_t c = l.slow[i];
                    store_holes(view, out, c, chunk_dest(l.rank, l.keep, c));
                }
            } else if (tid == 0) {
                atomicMax(&hdr->error, (uint32_t)(-HBS_E_CAPACITY));
            }
        }
        __syncthreads();
        HBS_T_MARK(6)
        } while (0);
        if (sched == kSchedTicketTop) next_tile = take_ticket(l, hdr, launder(tid0));
        tile = next_tile;
    }
    HBS_T_FLUSH
}

__global__ __launch_bounds__(kThreads, 4)
void k_scan_extract(const uint8_t* __restrict__ stream, uint64_t n, uint64_t num_tiles,
                    hbs_nal_entry* __restrict__ index, uint64_t index_cap,
                    uint8_t* __restrict__ rbsp, uint64_t rbsp_cap,
                    unsigned long long* __restrict__ desc, RunHeader* __restrict__ hdr, int sched, int gate)
{
    if (gate_closed(gate, hdr)) return;
    __shared__ TileLds l;
    scan_tiles(l, stream, n, num_tiles, index, index_cap, rbsp, rbsp_cap, desc, hdr, sched);
}

#ifdef HBS_PHASE_TIMING
extern "C" int hbs_debug_set_stop(int phase)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stop_after), &phase, sizeof(int));
}
extern "C" int hbs_debug_phase_cycles(unsigned long long* host_out /* [1024][8] */)
{
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_phase_cycles), sizeof(unsigned long long) * 1024 * 8);
}
#endif

/* the run header is initialised by k_scan_prologue (hbs_scan4.hip); k_scan_small below does the same in front of its tile */

/* Behind the tiles, ONE launch (round 4; they were two, k_tail_fixup then k_fill_rbsp_len, each waiting for the one before):
 * thread 0 applies the end-of-stream rules and writes the summary; everybody derives rbsp_len (and drops TRAILING03 from
 * rejected NALs) for the entries whose successor the tiles already wrote.  The last entry the tiles found, and one the
 * end-of-stream rules append, are thread 0's -- it is the only thread that touches them, and nothing here writes
 * hdr->final_nals / final_kept, which every thread reads. */
constexpr int kFinishBlocksMin = 128, kFinishBlocksMax = 8192;   /* sized by the entries there can be (four a thread): 128 workgroups took 40 us over
                                                                    the 2 M entries of a 2 GiB stream of 1 KiB NALs, 89 us at 512 bytes */
__global__ __launch_bounds__(256)
void k_scan_finish(const uint8_t* __restrict__ stream, uint64_t n,
                   hbs_nal_entry* index, uint64_t index_cap,
                   uint8_t* rbsp, uint64_t rbsp_cap, RunHeader* hdr, hbs_summary* sum, AheadCtl* ahead_ctl)
{
    /* the call's last launch: the count-ahead's workspace (hbs_scan4.hip) is the next call's from here on */
    if (ahead_ctl && blockIdx.x == 0 && threadIdx.x == 0) { ahead_ctl->listed = 0u; ahead_ctl->call += 1ull; }
    const uint64_t found0 = hdr->final_nals;
    const uint64_t lim = found0 < index_cap ? found0 : index_cap;
    /* a wavefront takes 64 consecutive entries: each lane loads the second half of its entry (rbsp_off, rbsp_len, status) in
     * one piece, the next entry's rbsp_off comes from the lane above (lane 63 loads it) */
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t k0 = wave * 64u; k0 + 1 < lim; k0 += nwaves * 64u) {
        const uint64_t k = k0 + (uint64_t)lane;
        const bool in = k + 1 < lim;
        u32x4 h = u32x4{0u, 0u, 0u, 0u};
        if (k < lim) h = reinterpret_cast<const u32x4*>(index + k)[1];
        const uint64_t off = ((uint64_t)h.y << 32) | h.x;
        uint32_t nlo = (uint32_t)__shfl_down((int)h.x, 1, 64), nhi = (uint32_t)__shfl_down((int)h.y, 1, 64);
        if (lane == 63 && in) { const uint64_t o = index[k + 1].rbsp_off; nlo = (uint32_t)o; nhi = (uint32_t)(o >> 32); }
        if (in) {
            const int32_t st = (int32_t)h.w;
            uint2 o;
            o.x = (uint32_t)((((uint64_t)nhi << 32) | nlo) - off);
            o.y = (uint32_t)(((st & HBS_ST_ERROR) && (st & HBS_ST_TRAILING03)) ? (st & ~HBS_ST_TRAILING03) : st);
            *reinterpret_cast<uint2*>(&index[k].rbsp_len) = o;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        uint8_t tail[8];
        for (int i = 0; i < 8; ++i) {
            const int64_t q = (int64_t)n - 8 + i;
            tail[i] = (q >= 0) ? stream[q] : (uint8_t)0xFF;
        }
        const TailOut t = tail_fixup(hdr, index, index_cap, rbsp, rbsp_cap, tail, n, sum, false);
        if (lim > 0) fill_rbsp_len_v(index, index_cap, lim - 1, t.found, t.kept);
        if (t.found > t.found0) fill_rbsp_len_v(index, index_cap, t.found0, t.found, t.kept);
    }
}

/* A stream of at most one tile (the legacy single-NAL symbols call with a few KiB): everything the
 * call does -- header, index clear, the tile, the end-of-stream fix-up, rbsp_len -- in ONE launch
 * of one workgroup instead of nine launches. */
__device__ __forceinline__ void init_header(RunHeader* hdr)
{
    hdr->final_kept = 0; hdr->final_nals = 0; hdr->final_inside = 0;
    hdr->error = 0; hdr->first_empty = ~0ull; hdr->abort_flag = 0; hdr->ticket = 0;
    hdr->probe_chunks = 0; hdr->probe_flagged = 0;
}

__global__ __launch_bounds__(kThreads, 4)
void k_scan_small(const uint8_t* __restrict__ stream, uint64_t n,
                  hbs_nal_entry* index, uint64_t index_cap, uint8_t* rbsp, uint64_t rbsp_cap,
                  unsigned long long* __restrict__ desc, RunHeader* hdr, hbs_summary* sum)
{
    __shared__ TileLds l;
    const int tid = threadIdx.x;
    if (tid == 0) init_header(hdr);
    {
        unsigned long long* q = reinterpret_cast<unsigned long long*>(index);
        for (uint64_t i = (uint64_t)tid; i < index_cap * (sizeof(hbs_nal_entry) / 8); i += kThreads) q[i] = 0ull;
    }
    __threadfence();
    __syncthreads();
    if (n) scan_tiles(l, stream, n, 1, index, index_cap, rbsp, rbsp_cap, desc, hdr, kSchedStriped);
    __threadfence();
    __syncthreads();
    if (tid == 0) {
        uint8_t tail[8];
        for (int i = 0; i < 8; ++i) {
            const int64_t q = (int64_t)n - 8 + i;
            tail[i] = (q >= 0) ? stream[q] : (uint8_t)0xFF;
        }
        tail_fixup(hdr, index, index_cap, rbsp, rbsp_cap, tail, n, sum);
    }
    __threadfence();
    __syncthreads();
    const uint64_t found = *reinterpret_cast<volatile uint64_t*>(&hdr->final_nals);
    for (uint64_t k = (uint64_t)tid; k < found; k += kThreads) fill_rbsp_len(hdr, index, index_cap, k);
}

bool scan_takes_small_path(uint64_t n, uint64_t index_cap, int variant)
{
    return variant == 0 && n <= (uint64_t)kTileBytes && index_cap <= 16384;
}

/* ---- host side ------------------------------------------------------------ */

int scan_grid_blocks(int device, int* blocks_per_cu_out)
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return -1;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_scan_extract, kThreads, 0) != hipSuccess) return -1;
    /* All workgroups must be co-resident (tile i waits on tiles < i).  LDS
     * (78 KiB per workgroup) admits two per CU; never ask for more than the
     * occupancy query grants. */
    if (per_cu > 2) per_cu = 2;
    if (per_cu < 1) per_cu = 1;
    if (blocks_per_cu_out) *blocks_per_cu_out = per_cu;
    return prop.multiProcessorCount * per_cu;
}

hipError_t launch_scan_extract(const ScanArgs& a, hipStream_t st)
{
    hipError_t e;
    const bool automatic = a.variant == 0;
    if (scan_takes_small_path(a.n, a.index_cap, a.variant)) {
        if (a.ev_begin) { e = hipEventRecord(a.ev_begin, st); if (e != hipSuccess) return e; }
        k_scan_small<<<1, dim3(kThreads), 0, st>>>(a.stream, a.n, a.index, a.index_cap, a.rbsp, a.rbsp_cap, a.desc, a.hdr, a.summary);
        if (a.ev_end) { e = hipEventRecord(a.ev_end, st); if (e != hipSuccess) return e; }
        return hipGetLastError();
    }
    /* no arena asked for: the streaming index-only kernel takes the place of the event-sparse one */
    /* (from 1 GiB up: below, the event-sparse kernel without an arena is quicker -- 0.126 against 0.139 ms at 512 MiB, 0.233
     * against 0.227 at 1 GiB, 0.458 against 0.403 at 2 GiB; scripts/experiments/index_only_by_size.py) */
    const bool index_only = scan_uses_index_only(a.n, a.variant, a.rbsp);
    const int sparse_variant = ((a.variant == 5 && !index_only) || automatic) ? 4 : a.variant;
    const uint64_t tiles4 = (a.n + (uint64_t)scan4_tile_bytes() - 1) / (uint64_t)scan4_tile_bytes();
    const uint64_t tiles6 = (a.n + (uint64_t)scan4r24_tile_bytes() - 1) / (uint64_t)scan4r24_tile_bytes();
    const uint64_t tiles2 = (a.n + (uint64_t)kTileBytes - 1) / (uint64_t)kTileBytes;
    /* the finest tiling any kernel of this call may use sizes the look-back words to clear */
    const uint64_t num_tiles = automatic ? tiles2 : sparse_variant == 4 ? tiles4 : sparse_variant == 6 ? tiles6 : tiles2;
    /* one launch: run header, density probe (automatic mode), padded copy of the stream's last 192 KiB (event-sparse kernels, either
     * geometry), cleared index and look-back words */
    const int tail_tile = ((!index_only && sparse_variant == 4) || sparse_variant == 6 || automatic) ? scan4_tile_bytes() : 0;
    launch_scan_prologue(a, num_tiles * 2, automatic && a.n != 0, tail_tile, st);
    hipError_t first = hipSuccess;
    auto note = [&](hipError_t x) { if (first == hipSuccess && x != hipSuccess) first = x; };
    if (num_tiles) {
        uint64_t grid = (uint64_t)a.grid_blocks;
        if (grid > tiles2) grid = tiles2;
        /* (an error from here on is remembered, not returned at once: the prologue has stamped the count-ahead's workspace with this
         * call's number, and only the finish launch below hands the workspace to the next call -- round 5's advice) */
        if (a.ev_begin) note(hipEventRecord(a.ev_begin, st));
        if (index_only && !automatic) {
            launch_scan_index5(a, kGateNone, st);
        } else if (!automatic && sparse_variant == 4) {
            launch_scan_ahead4(a, tiles4, kGateNone, st);           /* (does nothing unless the call carries the count-ahead's workspace) */
            scan4_launch_kernel(a, tiles4, kGateNone, st);
        } else if (!automatic && sparse_variant == 6) {
            scan4r24_launch_kernel(a, tiles6, kGateNone, st);
        } else if (automatic) {
            /* All the kernels the probe may pick are enqueued; each reads the probe's verdict from the run header and the
             * ones it rules out return at once (no host round trip, the call stays asynchronous, capturable).
             * They share the descriptor array and the ticket: whichever runs finds both untouched. */
            /* (a side stream for the kernels that rule themselves out, forked and joined with events, was tried in round 4: the two
             * event waits cost more than an empty kernel's 4.7 us -- a 1 GiB call 0.427 ms against 0.410) */
            if (index_only) launch_scan_index5(a, kGateIfSparseIdx, st);
            else { launch_scan_ahead4(a, tiles4, kGateIfSparse, st); scan4_launch_kernel(a, tiles4, kGateIfSparse, st); }
            /* dense but regular (NALs of ~120-450 bytes): the 24-row geometry (not behind the streaming index-only kernel, which
             * holds to one element in 9 chunks itself: hbs_common.h) */
            if (!index_only) scan4r24_launch_kernel(a, tiles6, kGateIfMid, st);
            k_scan_extract<<<dim3((unsigned)grid), dim3(kThreads), 0, st>>>(
                a.stream, a.n, tiles2, a.index, a.index_cap, a.rbsp, a.rbsp_cap, a.desc, a.hdr, a.sched, index_only ? kGateIfDenseIdx : kGateIfDense);
        } else {
            k_scan_extract<<<dim3((unsigned)grid), dim3(kThreads), 0, st>>>(
                a.stream, a.n, tiles2, a.index, a.index_cap, a.rbsp, a.rbsp_cap, a.desc, a.hdr, a.sched, kGateNone);
        }
        if (a.ev_end) note(hipEventRecord(a.ev_end, st));
    }
    /* no more entries than start codes fit the stream (3 bytes each) */
    const uint64_t may = a.index_cap < a.n / 3 + 1 ? a.index_cap : a.n / 3 + 1;
    uint64_t fb = (may + 1023) / 1024;
    fb = fb < (uint64_t)kFinishBlocksMin ? (uint64_t)kFinishBlocksMin : fb > (uint64_t)kFinishBlocksMax ? (uint64_t)kFinishBlocksMax : fb;
    k_scan_finish<<<dim3((unsigned)fb), 256, 0, st>>>(a.stream, a.n, a.index, a.index_cap, a.rbsp, a.rbsp_cap, a.hdr, a.summary, a.ahead_ctl);
    note(hipGetLastError());
    return first;
}

} // namespace hbs
