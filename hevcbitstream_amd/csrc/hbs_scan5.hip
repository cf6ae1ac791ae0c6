/*
 * hbs_scan5.hip -- index only (find_nal_unit over a whole stream, no RBSP arena).
 *
 * Without an arena to fill nothing has to stay in registers, so the event-sparse scan turns into
 * a STREAMING kernel: a wavefront takes a tile of 256 KiB by ticket, reads it 16 KiB at a time
 * (the next 16 KiB already in flight), keeps only each chunk's chunk_flag() -- one 64-bit
 * word per KiB, in LDS -- and then treats the flagged chunks exactly as hbs_scan4.hip treats its
 * elements: the window rules on the chunk's bytes [-8, 20) (fetched again from the stream: a few
 * per 64 KiB in coded video, L2 hits), 64 at a time; tile aggregate.  A wavefront is alone in its
 * workgroup, there is no barrier anywhere -- and, since round 3, no look-back either: the tiles'
 * prefixes are formed by two small kernels behind the streaming one and a fourth writes the index
 * entries from the elements the first one recorded (see "two passes" below).
 *
 * Results are those of k_scan_extract4 with rbsp == nullptr, bit for bit (same element code, same
 * end-of-stream fix-up behind it).  Replaces the byte loop of find_nal_unit (reference
 * h264_nal.c:38-76) over a stream, as hbs_scan4.hip does.
 */
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "hbs_elems.h"

namespace hbs {

#ifndef HBS5_ROWS_KERNEL
#define HBS5_ROWS_KERNEL 0     /* 1: the emit pass's row walks (tiles of padding or zeros) as a launch of their own */
#endif
#ifndef HBS5_SPAN_ROWS
#define HBS5_SPAN_ROWS 16
#define HBS5_TILE_ROWS 256
#endif
constexpr int k5SpanRows = HBS5_SPAN_ROWS;                    /* rows of 1 KiB a wavefront flags per step */
constexpr uint64_t k5SpanBytes = (uint64_t)k5SpanRows * 1024u;
/* KiB of stream (= flag words) per tile: a launch parameter since round 5 (`rows`, a whole number of spans, 64 ... 512).  From 4 GiB
 * up it is k5TileRowsLarge, as always; below, the tiles are cut so that every resident wavefront gets the SAME whole number of
 * them (scan5_tile_rows): a 1 GiB stream in 256 KiB tiles is 4096 tiles for 3072 wavefronts -- a third of them took a second
 * tile while the others sat idle, 220 us where the 16 GiB rate says 180. */
constexpr int k5TileRowsLarge = HBS5_TILE_ROWS;
constexpr int k5MaxTileRows = 512, k5MinTileRows = 64;
constexpr int k5MaxWordsPerLane = k5MaxTileRows / 64;         /* lane l owns words [wpl l, wpl l + wpl) of its tile when elements are numbered; wpl = ceil(rows / 64) */
static_assert(k5TileRowsLarge % k5SpanRows == 0 && k5TileRowsLarge <= k5MaxTileRows && k5MinTileRows % k5SpanRows == 0, "tile = whole spans");
__host__ __device__ inline uint32_t words_per_lane5(int rows) { return (uint32_t)((rows + 63) / 64); }

/* ---- the streaming half: flags of one span of k5SpanRows KiB --------------------------------- */

/* one span in registers: its k5SpanRows rows and the dwords just outside */
struct Span5 { u32x4 q[k5SpanRows]; uint32_t before, before2, after; };

__device__ __forceinline__ void span_load(Span5& s, const uint8_t* __restrict__ stream, uint64_t n, uint64_t base, int lane)
{
    if (base + k5SpanBytes <= n) {
#pragma unroll
        for (int r = 0; r < k5SpanRows; ++r) s.q[r] = stream_load16(reinterpret_cast<const u32x4*>(stream + base + 1024u * r + 16u * lane));
    } else {
#pragma unroll
        for (int r = 0; r < k5SpanRows; ++r) s.q[r] = load_chunk_guarded(stream, base + 1024u * r + 16u * lane, n);
    }
    s.before = base >= 4 ? *reinterpret_cast<const uint32_t*>(stream + base - 4) : 0xFFFFFFFFu;
    s.before2 = base >= 8 ? *reinterpret_cast<const uint32_t*>(stream + base - 8) : 0xFFFFFFFFu;
    s.after = load_dword_guarded(stream, (int64_t)(base + k5SpanBytes), n);
}

/* what a flagged lane leaves for the batch that walks its chunk as an element (round 4): the chunk's bytes [-8, 20) and its
 * number in the tile, in a ring of k5Ring entries in LDS.  Until round 4 a batch FETCHED those bytes again from the stream --
 * by then (a tile is 256 KiB and 3072 wavefronts stream at once) they had left the L2: a dependent round trip to HBM per
 * batch at loaded latency, and a sector per element on top of the stream's bytes (12 % at one NAL per KiB). */
constexpr uint32_t k5Ring = 192;
struct Ring5 { uint32_t head, tail, pending; bool spilled; };     /* head, tail: modulo k5Ring */

/* lane r of the result: the flag word of row r of the span at `base` (bit l = chunk l of that row).
 * kDeposit: the flagged lanes also leave a Deposit each, in order, at ring slots tail .. (the first `room` of them);
 * chunk0 = the span's first chunk number in its tile; count = the span's elements. */
template <bool kDeposit>
__device__ __forceinline__ unsigned long long span_flags(const Span5& s, uint64_t n, uint64_t base, uint64_t cut, int lane,
                                                         Deposit* __restrict__ dep, uint32_t chunk0, uint32_t tail, uint32_t room, uint32_t& count)
{
    unsigned long long mine = 0;
    uint32_t cnt = 0;
#pragma unroll
    for (int r = 0; r < k5SpanRows; ++r) {
        const uint32_t e_prev = r == 0 ? s.before : (uint32_t)__builtin_amdgcn_readlane((int)s.q[r ? r - 1 : 0].w, 63);
        const uint32_t e_next = r == k5SpanRows - 1 ? s.after : (uint32_t)__builtin_amdgcn_readlane((int)s.q[r + 1 < k5SpanRows ? r + 1 : r].x, 0);
        const uint32_t xp = from_prev_lane(s.q[r].w, e_prev);
        const uint32_t xn = from_next_lane(s.q[r].x, e_next);
        const uint64_t g0 = base + 1024u * r + 16u * lane;
        /* the chunk cut by the stream end is always an element */
        bool f = g0 < n && chunk_flag(xp, s.q[r].x, s.q[r].y, s.q[r].z, s.q[r].w, xn);
        unsigned long long m = __ballot(f);
        if (__builtin_popcountll(m) > kExactFlagMin) {           /* many zero pairs in this KiB: which of them are followed by a byte <= 3? (as hbs_scan4.hip) */
            f = f && chunk_pattern_any_dev(xp, s.q[r].x, s.q[r].y, s.q[r].z, s.q[r].w, xn);
            m = __ballot(f);
        }
        m |= __ballot(g0 < n && (g0 >> 4) == cut);
        if (lane == r) mine = m;
        if (kDeposit && m != 0ull) {
            const uint32_t e_prev_z = r == 0 ? s.before2 : (uint32_t)__builtin_amdgcn_readlane((int)s.q[r ? r - 1 : 0].z, 63);
            const uint32_t xpp = from_prev_lane(s.q[r].z, e_prev_z);
            const uint32_t rank = cnt + lanes_below(m);
            if (((m >> lane) & 1ull) != 0ull && rank < room) {
                uint32_t slot = tail + rank;
                if (slot >= k5Ring) slot -= k5Ring;
                Deposit d;
                d.xpp = xpp; d.xp = xp; d.x0 = s.q[r].x; d.x1 = s.q[r].y; d.x2 = s.q[r].z; d.x3 = s.q[r].w; d.xn = xn;
                d.chunk = chunk0 + 64u * (uint32_t)r + (uint32_t)lane;
                dep[slot] = d;
            }
            cnt += (uint32_t)__builtin_popcountll(m);
        }
    }
    count = cnt;
    return mine;
}

/* ---- the element half ------------------------------------------------------------------------ */

struct Lds5 {
    unsigned long long words[k5MaxTileRows];   /* the tile's mask: [0, rows); zero up to 64 x words_per_lane5(rows)  */
    uint32_t lane_pre[64];                     /* elements in front of lane l's words               */
    uint32_t seg_dummy[64];                    /* elem_emit leaves a segment word per element: nobody copies here */
};

/* chunk number (in the tile) of element i: the (i - lane_pre[o])-th set bit of owner lane o's words */
__device__ __forceinline__ uint32_t elem_chunk(const Lds5& l, uint32_t i, uint32_t wpl)
{
    /* owner: the last lane whose prefix is <= i */
    uint32_t lo = 0, hi = 63;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (l.lane_pre[mid] <= i) lo = mid; else hi = mid - 1;
    }
    uint32_t k = i - l.lane_pre[lo];
    uint32_t w = lo * wpl;
    unsigned long long x = l.words[w];
    for (;;) {
        const uint32_t c = (uint32_t)__builtin_popcountll(x);
        if (k < c) break;
        k -= c;
        x = l.words[++w];
    }
    for (; k; --k) x &= x - 1;                                /* drop the k lowest set bits */
    return w * 64u + (uint32_t)__builtin_ctzll(x);
}

/* elements [i0, i0 + 64) of the tile, one per lane: the exact rules on each, then their combination in order.
 * prev_chunk_end = end of the element in front of element i0 (stream offset), carried from batch to batch. */
__device__ __forceinline__ TileAgg make_batch(Elem& el, const Lds5& l, uint32_t i0, uint32_t nelem, int lane,
                                              const uint8_t* __restrict__ stream, uint64_t base, uint64_t n, uint64_t& prev_end, uint32_t wpl)
{
    const uint32_t i = i0 + (uint32_t)lane;
    const bool have = i < nelem;
    uint32_t c = 0;
    if (have) c = elem_chunk(l, i, wpl);
    /* the element in front of mine: the lane below, or what the previous batch left */
    const uint32_t c_prev = (uint32_t)__shfl_up((int)c, 1, 64);
    const uint64_t my_prev_end = lane == 0 ? prev_end : base + 16ull * ((uint64_t)c_prev + 1u);
    TileAgg ea = agg_identity();
    el.gap = 0; el.chunk = 0;
    if (have) {
        elem_load(el.v, stream, base + 16ull * c, n, false);
        elem_walk(el.v, el.m, el.s, el.cls);
        el.gap = span_bytes(my_prev_end, el.v.g0, n);
        el.chunk = c;
        ea = elem_agg(el.gap, el.s);
    }
    /* the last element of this batch, for the next one */
    const uint32_t cnt = nelem - i0 < 64u ? nelem - i0 : 64u;
    const uint32_t c_last = (uint32_t)__shfl((int)c, (int)(cnt - 1u), 64);
    prev_end = base + 16ull * ((uint64_t)c_last + 1u);
    return ea;
}

/* the same for `cnt` (<= 64) elements whose bytes wait in the ring, from slot `head` on */
__device__ __forceinline__ TileAgg ring_batch(Elem& el, const Deposit* __restrict__ dep, uint32_t head, uint32_t cnt, int lane,
                                              const uint8_t* __restrict__ stream, uint64_t base, uint64_t n, uint64_t& prev_end)
{
    const bool have = (uint32_t)lane < cnt;
    uint32_t slot = head + (have ? (uint32_t)lane : 0u);
    if (slot >= k5Ring) slot -= k5Ring;
    const Deposit& d = dep[slot];
    const uint32_t c = d.chunk;
    const uint32_t c_prev = (uint32_t)__shfl_up((int)c, 1, 64);
    const uint64_t my_prev_end = lane == 0 ? prev_end : base + 16ull * ((uint64_t)c_prev + 1u);
    TileAgg ea = agg_identity();
    el.gap = 0; el.chunk = 0;
    if (have) {
        el.v.xpp = d.xpp; el.v.xp = d.xp; el.v.x0 = d.x0; el.v.x1 = d.x1; el.v.x2 = d.x2; el.v.x3 = d.x3; el.v.xn = d.xn;
        el.v.stream = stream; el.v.g0 = base + 16ull * c; el.v.n = n;
        elem_walk(el.v, el.m, el.s, el.cls);
        el.gap = span_bytes(my_prev_end, el.v.g0, n);
        el.chunk = c;
        ea = elem_agg(el.gap, el.s);
    }
    const uint32_t c_last = (uint32_t)__shfl((int)c, (int)(cnt - 1u), 64);
    prev_end = base + 16ull * ((uint64_t)c_last + 1u);
    return ea;
}

/* ---- tiles dense in elements: a row at a time ------------------------------------------------------------------
 * Past two batches the elements of a tile are walked 64 at a time, each batch finding its chunks in the flag words and
 * fetching their bytes one element per lane: a tile inside 00 00 03 padding or zero stuffing (65536 elements per MiB)
 * took milliseconds that way, and the look-backs of every tile behind it waited -- an index-only scan of a stream with
 * 1 % of such bytes ran 14.6 times slower than without (scripts/mixed_time.py).  When the flagged rows of a tile hold 16
 * or more elements each on average, the tile is walked by ROWS instead: every KiB row that has a flag is one batch, all 64
 * of its chunks elements (no gaps inside the row), read with one coalesced load that is issued a row ahead. */
/* The rows of a tile again, in order, kRowsAhead at a time (the next batch's loads in flight while this one is walked: with
 * one row in flight every row cost a memory round trip): f(r, previous row, row, next row) for the rows that hold a flag. */
constexpr int kRowsAhead = 8;
static_assert(k5SpanRows % kRowsAhead == 0, "whole batches of rows in any tile");
struct RowEdges5 { uint32_t before, before2, after; };
__device__ __forceinline__ RowEdges5 rows_edges(const uint8_t* __restrict__ stream, uint64_t n, uint64_t base, uint64_t tile_end)
{
    RowEdges5 e;
    e.before = load_dword_guarded(stream, (int64_t)base - 4, n);
    e.before2 = load_dword_guarded(stream, (int64_t)base - 8, n);
    e.after = load_dword_guarded(stream, (int64_t)tile_end, n);
    return e;
}
template <class F>
__device__ __forceinline__ void tile_rows(const uint8_t* __restrict__ stream, uint64_t n, uint64_t base, int rows, int lane, const Lds5& l, F&& f)
{
    const bool whole = base + 1024ull * (uint64_t)rows <= n;     /* every row of the tile is there: plain loads */
    auto fetch = [&](int r) -> u32x4 {
        const uint64_t g = base + 1024ull * (uint64_t)(r < rows ? r : rows - 1) + 16ull * (uint64_t)lane;
        return whole ? *reinterpret_cast<const u32x4*>(stream + g) : load_chunk_guarded(stream, g, n);
    };
    u32x4 cur[kRowsAhead + 2];                                    /* rows b - 1 .. b + kRowsAhead */
#pragma unroll
    for (int i = 0; i <= kRowsAhead; ++i) cur[i + 1] = fetch(i);
    cur[0] = cur[1];
#pragma unroll 1
    for (int b = 0; b < rows; b += kRowsAhead) {
        u32x4 nxt[kRowsAhead];
#pragma unroll
        for (int i = 0; i < kRowsAhead; ++i) nxt[i] = fetch(b + kRowsAhead + 1 + i);
#pragma unroll
        for (int i = 0; i < kRowsAhead; ++i)
            if (l.words[b + i] != 0ull) f(b + i, cur[i], cur[i + 1], cur[i + 2]);
        cur[0] = cur[kRowsAhead]; cur[1] = cur[kRowsAhead + 1];
#pragma unroll
        for (int i = 0; i < kRowsAhead; ++i) cur[i + 2] = nxt[i];
    }
}

/* One flagged row.  1: no terminator ends in it (kErrToo: and nothing that marks an error): all it adds is `quick_bytes`
 * state-dependent bytes (the gap in front of it + its kept bytes), found with dense_row_quick's ~100 instructions instead of
 * the walk's ~300 -- rows of 00 00 03 padding; 2: d holds the row's 64 elements */
template <bool kErrToo>
__device__ __forceinline__ int row_visit(DenseRow& d, const u32x4& qp, const u32x4& qc, const u32x4& qn, int r, int rows, const RowEdges5& e,
                                         const uint8_t* __restrict__ stream, uint64_t n, uint64_t base, int lane, uint64_t& prev_end, uint32_t& quick_bytes)
{
    const uint64_t row_lo = base + 1024ull * (uint64_t)r;
    const uint32_t gap0 = span_bytes(prev_end, row_lo, n);
    prev_end = row_lo + 1024ull;
    uint32_t kept;
    if (row_lo + 1024ull + 64ull <= n && !dense_row_quick<kErrToo>(qp, qc, qn, r, rows, e.before, e.after, kept)) {
        quick_bytes = gap0 + wave_sum32(kept);
        return 1;
    }
    dense_row(d, qp, qc, qn, r, rows, e.before, e.before2, e.after, stream, base, n, 0u, lane);
    d.el.gap = lane == 0 ? gap0 : 0u;
    return 2;
}

/* ---- two passes, no look-back (round 3) ---------------------------------------------------------------------------
 * Until round 3 this was ONE kernel: a wavefront streamed its tile, walked its elements, published the tile's aggregate,
 * looked back over the tiles in front (hbs_elems.h) and wrote its index entries.  With nothing held in registers the
 * look-back looked cheap, but a wavefront that waits for its predecessors is a wavefront that does not read, and tiles
 * finish in convoys: without the look-back (wrong results, timing only) the same kernel took 2.65 ms instead of 2.97 on
 * the 16 GiB bench stream, the pure read of its geometry 2.53 (profiles/r03/idx_ablation.txt, ceiling3_read.txt).
 * An index-only scan does not need one: nothing a tile does while it streams depends on the tiles in front of it.  So:
 *
 *   k_index5_stream   tile by ticket: flag words, elements, the tile's aggregate -- and the elements themselves (what the
 *                     emit half needs of each: 32 bytes) into the workspace, up to 2048 per tile (the bench stream has ~145 per
 *                     MiB).  No waiting anywhere.
 *   k_index5_chunks   aggregate of every 64 consecutive tiles (one wavefront each)
 *   k_index5_prefix   one wavefront: the prefix in front of every chunk of 64 tiles, 64 chunks per step
 *   k_index5_emit     one wavefront per tile: its prefix (chunk prefix + the tiles of its chunk in front of it), then the
 *                     index entries from the recorded elements.  A tile with more elements than are recorded, or one
 *                     that was walked by rows, is streamed and walked again here, with the prefix known: rare, and
 *                     nobody waits for it.
 *
 * The passes behind the first move ~0.5 % of the stream's bytes.  Same results as before, bit for bit. */
struct Rec5 { uint32_t chunk, gap, pa, pb, pc, z, e1, e3; };             /* one element, as the emit half needs it */
static_assert(sizeof(Rec5) == 32, "two 16-byte stores per element");
struct Pre5 { unsigned long long kept, nals; uint32_t inside, pad; };     /* a Prefix in memory */
__host__ __device__ inline uint32_t rec_cap5(int rows) { return 8u * (uint32_t)rows; }   /* elements recorded per tile: eight a KiB (25 % of the stream's size as workspace).  A tile with more is streamed twice:
                                                                             two a KiB until round 4 (NALs of 512 bytes passed it in every other tile), four until round 6 (NALs of 384 held, of 256 -- 4.8 elements
                                                                             a KiB -- did not: 0.22 of peak, against 0.39 with eight; 192-byte NALs 0.19 -> 0.33) */
constexpr uint32_t k5Rewalk = 0xFFFFFFFFu;                                /* nrec: the emit pass walks the tile again */
/* ... a tile that was walked by rows (a stretch of padding, of zeros): in parts of kPartRows rows, by different wavefronts
 * (round 5).  One wavefront took ~400 us over the 256 rows of such a tile, and k_index5_emit took as long as its slowest tile:
 * 0.47 ms instead of 0.07 on the bench's mixed stream (1 % of it in 640 KiB stretches), all of the index-only scan's 1.16 x.
 * The stream pass leaves the aggregate in front of every part where the tile's records would be (it has none) and lists the
 * tile; the tile's own wavefront takes part 0, and all wavefronts of the launch share the other parts of all listed tiles once
 * their own tile is done. */
constexpr uint32_t k5RewalkParts = 0xFFFFFFFEu;
constexpr uint32_t k5RewalkWhole = 0xFFFFFFFDu;                           /* ... and one that only the emit pass found dense (not expected): all its rows by its own wavefront */
constexpr int kPartRows = 32;
static_assert(kPartRows % k5SpanRows == 0 && k5MaxTileRows / kPartRows <= 16, "a part is whole spans; its aggregates fit the tile's record space many times over");
constexpr int k5ChunkTiles = 64;

struct Ws5 {
    TileAgg* tagg;       /* [tiles]                  */
    uint32_t* nrec;      /* [tiles]                  */
    uint32_t* rwlist;    /* [tiles]: tiles walked by rows, in the order the stream pass met them (RunHeader::rewalk_count of them) */
    TileAgg* cagg;       /* [chunks]                 */
    Pre5* cpre;          /* [chunks]                 */
    Rec5* rec;           /* [tiles][rec_cap5(rows)]  */
};
__host__ __device__ inline uint64_t ws5_chunks(uint64_t tiles) { return (tiles + k5ChunkTiles - 1) / k5ChunkTiles; }
__host__ __device__ inline Ws5 ws5_carve(void* base, uint64_t tiles, int rows)
{
    uint8_t* p = static_cast<uint8_t*>(base);
    Ws5 w;
    const uint64_t ch = ws5_chunks(tiles);
    w.rec = reinterpret_cast<Rec5*>(p); p += tiles * rec_cap5(rows) * sizeof(Rec5);
    w.tagg = reinterpret_cast<TileAgg*>(p); p += ((tiles * sizeof(TileAgg) + 255) & ~255ull);
    w.cagg = reinterpret_cast<TileAgg*>(p); p += ((ch * sizeof(TileAgg) + 255) & ~255ull);
    w.cpre = reinterpret_cast<Pre5*>(p); p += ((ch * sizeof(Pre5) + 255) & ~255ull);
    w.nrec = reinterpret_cast<uint32_t*>(p); p += ((tiles * sizeof(uint32_t) + 255) & ~255ull);
    w.rwlist = reinterpret_cast<uint32_t*>(p);
    return w;
}
uint64_t scan5_workspace_bytes(uint64_t stream_bytes)
{
    /* whatever the tile height of the call will be: the records are eight a KiB (plus one tile's worth of rounding at the largest
     * height), the per-tile and per-chunk arrays are counted at the smallest */
    const uint64_t rec = (stream_bytes / 1024u + 2u * (uint64_t)k5MaxTileRows) * 8u * sizeof(Rec5);
    const uint64_t tiles = stream_bytes / (1024u * (uint64_t)k5MinTileRows) + 2, ch = ws5_chunks(tiles);
    return rec + ((tiles * sizeof(TileAgg) + 255) & ~255ull) + ((ch * sizeof(TileAgg) + 255) & ~255ull) +
           ((ch * sizeof(Pre5) + 255) & ~255ull) + 2 * ((tiles * sizeof(uint32_t) + 255) & ~255ull) + 256;
}

/* the streaming half of a tile: its flag words into l.words, one span at a time, the next span's loads in flight meanwhile */
__device__ __forceinline__ void tile_words(Lds5& l, const uint8_t* __restrict__ stream, uint64_t n, uint64_t base, int rows, uint64_t cut, int lane)
{
    Span5 cur, nxt;
    const int nsp = rows / k5SpanRows;
    if (base < n) span_load(nxt, stream, n, base, launder_lane(lane));
#pragma unroll 1
    for (int sp = 0; sp < nsp; ++sp) {
        const uint64_t sbase = base + (uint64_t)sp * k5SpanBytes;
        cur = nxt;
        if (sp + 1 < nsp && sbase + k5SpanBytes < n) span_load(nxt, stream, n, sbase + k5SpanBytes, launder_lane(lane));
        unsigned long long w = 0;
        uint32_t cnt;
        if (sbase < n) w = span_flags<false>(cur, n, sbase, cut, launder_lane(lane), nullptr, 0u, 0u, 0u, cnt);
        if (lane < k5SpanRows) l.words[sp * k5SpanRows + lane] = w;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

/* ... and, for the streaming kernel, the flagged chunks' bytes into the ring; whenever 64 of them wait, on_batch(head, 64) walks
 * them (the next span's loads are in flight meanwhile).  A span that does not fit the ring's free slots spills the tile:
 * nothing more is deposited or walked, and the caller does the tile's elements the old way, from its flag words. */
template <class F>
__device__ __forceinline__ Ring5 tile_words_ring(Lds5& l, Deposit* __restrict__ dep, const uint8_t* __restrict__ stream, uint64_t n,
                                                 uint64_t base, int rows, uint64_t cut, int lane, F&& on_batch)
{
    Span5 cur, nxt;
    Ring5 rg; rg.head = 0; rg.tail = 0; rg.pending = 0; rg.spilled = false;
    const int nsp = rows / k5SpanRows;
    if (base < n) span_load(nxt, stream, n, base, launder_lane(lane));
#pragma unroll 1
    for (int sp = 0; sp < nsp; ++sp) {
        const uint64_t sbase = base + (uint64_t)sp * k5SpanBytes;
        cur = nxt;
        if (sp + 1 < nsp && sbase + k5SpanBytes < n) span_load(nxt, stream, n, sbase + k5SpanBytes, launder_lane(lane));
        unsigned long long w = 0;
        uint32_t cnt = 0;
        if (sbase < n) w = span_flags<true>(cur, n, sbase, cut, launder_lane(lane), dep, (uint32_t)(sp * k5SpanRows * 64), rg.tail,
                                            rg.spilled ? 0u : k5Ring - rg.pending, cnt);
        if (lane < k5SpanRows) l.words[sp * k5SpanRows + lane] = w;
        cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt);
        if (!rg.spilled) {
            if (rg.pending + cnt > k5Ring) rg.spilled = true;
            else {
                rg.pending += cnt;
                rg.tail += cnt; if (rg.tail >= k5Ring) rg.tail -= k5Ring;
                if (rg.pending >= 64u) {
                    __builtin_amdgcn_wave_barrier();
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
                    while (rg.pending >= 64u) {
                        on_batch(rg.head, 64u);
                        rg.head += 64u; if (rg.head >= k5Ring) rg.head -= k5Ring;
                        rg.pending -= 64u;
                    }
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return rg;
}

/* elements of the tile and whether it is walked by rows (16 or more elements per flagged row: zero stuffing, padding) */
__device__ __forceinline__ uint32_t tile_census(Lds5& l, int lane, uint32_t wpl, bool& by_rows)
{
    uint32_t mycnt = 0, myrows = 0;
#pragma unroll
    for (int j = 0; j < k5MaxWordsPerLane; ++j) {
        if ((uint32_t)j < wpl) {
            const unsigned long long w = l.words[(uint32_t)lane * wpl + (uint32_t)j];      /* (words past the tile's rows are zero) */
            mycnt += (uint32_t)__builtin_popcountll(w);
            myrows += w != 0ull ? 1u : 0u;
        }
    }
    const uint32_t inc = wave_incl_scan32(mycnt, lane);
    l.lane_pre[lane] = inc - mycnt;
    const uint32_t nelem = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const uint32_t flagged_rows = wave_sum32(myrows);
    by_rows = nelem > 128u && nelem >= 16u * flagged_rows;
    return nelem;
}

__device__ __forceinline__ void rec_store(Rec5* r, const Elem& el, uint64_t base)
{
    const ElemPacked p = elem_pack(el.m, el.s);
    u32x4 a, b;
    a.x = (uint32_t)((el.v.g0 - base) >> 4); a.y = el.gap; a.z = p.a; a.w = p.b;
    b.x = p.c; b.y = el.cls.z; b.z = el.cls.e1; b.w = el.cls.e3;
    u32x4* q = reinterpret_cast<u32x4*>(r);
    q[0] = a; q[1] = b;
}
__device__ __forceinline__ void rec_load(const Rec5* r, Elem& el, uint64_t base, uint64_t n)
{
    const u32x4* q = reinterpret_cast<const u32x4*>(r);
    const u32x4 a = q[0], b = q[1];
    ElemPacked p; p.a = a.z; p.b = a.w; p.c = b.x;
    elem_unpack(p, el.m, el.s);
    el.cls.z = b.y; el.cls.e1 = b.z; el.cls.e3 = b.w;
    el.chunk = a.x; el.gap = a.y;
    el.v.g0 = base + 16ull * a.x; el.v.n = n; el.v.stream = nullptr;
    el.v.xpp = el.v.xp = el.v.x0 = el.v.x1 = el.v.x2 = el.v.x3 = el.v.xn = 0;      /* bytes: only a copy would want them */
}

/* kRows: the tile height when it is the large streams' (a compile-time constant again: as a launch parameter -- round 5 -- the words
 * per lane, the record capacity and every loop bound over a tile's rows lived in SGPRs, 160 of them spilled where round 4's kernel
 * spilled 80, and the uniform 16 GiB scan lost 4-6 %); 0: any height, `rows_arg` (calls below 1.5 GiB) */
template <int kRows>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3)))
void k_index5_stream(const uint8_t* __restrict__ stream, uint64_t n, uint64_t num_tiles, int rows_arg, int strided, void* __restrict__ ws, RunHeader* __restrict__ hdr, int gate)
{
    if (gate_closed(gate, hdr)) return;
    __shared__ Lds5 l;
    __shared__ Deposit ring[k5Ring];
    const int rows = kRows ? kRows : rows_arg;
    const Ws5 w5 = ws5_carve(ws, num_tiles, rows);
    const int lane0 = threadIdx.x;
    const uint64_t cut = (n & 15ull) ? (n >> 4) : ~0ull;
    const uint64_t k5TileBytes = 1024ull * (uint64_t)rows;
    const uint32_t k5RecCap = rec_cap5(rows), wpl = words_per_lane5(rows);
    for (int i = rows + lane0; i < k5MaxTileRows; i += 64) l.words[i] = 0ull;     /* the census reads whole words-per-lane groups */
    /* Tiles.  Large streams (from 4 GiB): by ticket, as always -- twenty and more tiles per wavefront, dynamic balance.  Smaller
     * ones (`strided`): tile = workgroup number, + gridDim.x per round, no ticket at all.  Until round 5 every tile came by ticket:
     * 3072 wavefronts asked ONE address for their first ticket in the kernel's first microsecond and for a last, failing one in
     * its last; read-modify-writes on one address are served one after the other -- tens of microseconds of a 1 GiB call.
     * Nothing here waits for another tile, so the order tiles are started in is free, and the launcher cuts the stream so that
     * every wavefront gets the same number of tiles (scan5_geometry). */
    uint64_t tile = blockIdx.x;
    if (!strided) { uint32_t t0 = 0; if (lane0 == 0) t0 = atomicAdd(&hdr->ticket, 1u); tile = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)t0); }
    for (;;) {
        const int lane = launder_lane(lane0);
        if (tile >= num_tiles) break;
        const uint64_t base = tile * k5TileBytes;
        const uint64_t tile_end = base + k5TileBytes;
        TileAgg acc = agg_identity();
        uint64_t prev_end = base;
        uint32_t nwalked = 0;
        auto on_batch = [&](uint32_t head, uint32_t cnt) {
            Elem el;
            TileAgg ea = ring_batch(el, ring, head, cnt, lane, stream, base, n, prev_end);
            if ((uint32_t)lane < cnt && nwalked + (uint32_t)lane < k5RecCap) rec_store(&w5.rec[tile * k5RecCap + nwalked + (uint32_t)lane], el, base);
            ea = wave_scan_combine(ea, lane);
            acc = combine(acc, agg_readlane(ea, 63));
            nwalked += cnt;
        };
        const Ring5 rg = tile_words_ring(l, ring, stream, n, base, rows, cut, lane, on_batch);
        bool by_rows;
        const uint32_t nelem = tile_census(l, lane, wpl, by_rows);
        const uint32_t npass = (nelem + 63u) >> 6;
        if (!by_rows && !rg.spilled) {
            if (rg.pending) on_batch(rg.head, rg.pending);           /* the rest: fewer than 64 */
        } else if (by_rows) {
            acc = agg_identity();                                     /* (what was walked before the tile turned out dense is dropped) */
            prev_end = base;
            const RowEdges5 edges = rows_edges(stream, n, base, tile_end);
            /* the aggregate in front of every part of kPartRows rows, for the emit pass (k5RewalkParts): taken in front of the first
             * row visited at or behind the part's first (rows without a flagged chunk are not visited: their bytes are a gap) */
            TileAgg* const parts = reinterpret_cast<TileAgg*>(&w5.rec[tile * k5RecCap]);
            int next_part = 1;
            auto parts_upto = [&](int r) {
                while (next_part * kPartRows <= r && next_part * kPartRows < rows) {
                    const TileAgg sn = combine(acc, gap_agg(span_bytes(prev_end, base + 1024ull * (uint64_t)(kPartRows * next_part), n)));
                    if (lane == 0) parts[next_part - 1] = sn;
                    ++next_part;
                }
            };
            tile_rows(stream, n, base, rows, lane, l, [&](int r, const u32x4& qp, const u32x4& qc, const u32x4& qn) {
                DenseRow d;
                uint32_t quick = 0;
                parts_upto(r);
                if (row_visit<false>(d, qp, qc, qn, r, rows, edges, stream, n, base, lane, prev_end, quick) == 1) { acc = combine(acc, gap_agg(quick)); return; }
                const uint32_t gap0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.el.gap);
                if (!d.row_has_event) {
                    acc = combine(acc, gap_agg(gap0 + wave_sum32(d.el.s.carry)));      /* nothing but state-dependent bytes */
                } else {
                    const TileAgg ea = wave_scan_combine(elem_agg(d.el.gap, d.el.s), lane);
                    acc = combine(acc, agg_readlane(ea, 63));
                }
            });
            parts_upto(rows);
            if (lane == 0) w5.rwlist[atomicAdd(&hdr->rewalk_count, 1u)] = (uint32_t)tile;
        } else {                                                      /* spilled: the tile's elements from its flag words, their bytes from the stream */
            acc = agg_identity();
            prev_end = base;
            const bool record = nelem <= k5RecCap;
#pragma unroll 1
            for (uint32_t p = 0; p < npass; ++p) {
                Elem el;
                TileAgg ea = make_batch(el, l, 64u * p, nelem, lane, stream, base, n, prev_end, wpl);
                if (record && 64u * p + (uint32_t)lane < nelem) rec_store(&w5.rec[tile * k5RecCap + 64u * p + (uint32_t)lane], el, base);
                ea = wave_scan_combine(ea, lane);
                acc = combine(acc, agg_readlane(ea, 63));
            }
        }
        const TileAgg tagg = combine(acc, gap_agg(span_bytes(prev_end, tile_end, n)));
        if (lane == 0) {
            w5.tagg[tile] = tagg;
            w5.nrec[tile] = by_rows ? k5RewalkParts : nelem > k5RecCap ? k5Rewalk : nelem;
        }
        __builtin_amdgcn_wave_barrier();                       /* l is reused by the next tile */
        if (strided) tile += gridDim.x;
        else {
            uint32_t tk = 0;
            if (lane == 0) tk = atomicAdd(&hdr->ticket, 1u);
            tile = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)tk);
        }
    }
}

__device__ __forceinline__ TileAgg agg_load_or_identity(const TileAgg* a, uint64_t i, uint64_t count)
{
    TileAgg r = agg_identity();
    if (i < count) r = a[i];
    return r;
}

__global__ __launch_bounds__(64)
void k_index5_chunks(uint64_t num_tiles, int rows, void* __restrict__ ws, const RunHeader* __restrict__ hdr, int gate)
{
    if (gate_closed(gate, hdr)) return;
    const Ws5 w5 = ws5_carve(ws, num_tiles, rows);
    const int lane = threadIdx.x;
    const uint64_t c = blockIdx.x;
    const TileAgg a = wave_scan_combine(agg_load_or_identity(w5.tagg, c * k5ChunkTiles + (uint64_t)lane, num_tiles), lane);
    if (lane == 63) w5.cagg[c] = a;
}

__global__ __launch_bounds__(64)
void k_index5_prefix(uint64_t num_tiles, int rows, void* __restrict__ ws, const RunHeader* __restrict__ hdr, int gate)
{
    if (gate_closed(gate, hdr)) return;
    const Ws5 w5 = ws5_carve(ws, num_tiles, rows);
    const int lane = threadIdx.x;
    const uint64_t chunks = ws5_chunks(num_tiles);
    Prefix carry; carry.kept = 0; carry.nals = 0; carry.inside = 0;
    TileAgg nxt = agg_load_or_identity(w5.cagg, (uint64_t)lane, chunks);
#pragma unroll 1
    for (uint64_t c0 = 0; c0 < chunks; c0 += 64) {
        const TileAgg mine = nxt;
        nxt = agg_load_or_identity(w5.cagg, c0 + 64 + (uint64_t)lane, chunks);        /* the next step's, under this step's scan */
        const TileAgg inc = wave_scan_combine(mine, lane);
        TileAgg ex = agg_shfl_up(inc, 1);
        if (lane == 0) ex = agg_identity();
        const Prefix p = fold(carry, ex);
        if (c0 + (uint64_t)lane < chunks) {
            Pre5 o; o.kept = p.kept; o.nals = p.nals; o.inside = p.inside; o.pad = 0;
            w5.cpre[c0 + (uint64_t)lane] = o;
        }
        carry = prefix_uniform4(fold(carry, agg_readlane(inc, 63)));
    }
}

/* k_index5_chunks and k_index5_prefix in ONE launch of one workgroup, for streams of at most k5FusedChunks chunks (4096 tiles: every
 * call below ~1.5 GiB, where a launch is 2 % of the call): wavefront w forms the aggregates of chunks w, w + 16, ..., then wavefront
 * 0 their exclusive prefixes. */
constexpr int k5FusedWaves = 16, k5FusedChunks = 64;
__global__ __launch_bounds__(64 * k5FusedWaves)
void k_index5_chunks_prefix(uint64_t num_tiles, int rows, void* __restrict__ ws, const RunHeader* __restrict__ hdr, int gate)
{
    if (gate_closed(gate, hdr)) return;
    __shared__ TileAgg cagg[k5FusedChunks];
    const Ws5 w5 = ws5_carve(ws, num_tiles, rows);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t chunks = ws5_chunks(num_tiles);
    for (uint64_t c = (uint64_t)wv; c < chunks; c += k5FusedWaves) {
        const TileAgg a = wave_scan_combine(agg_load_or_identity(w5.tagg, c * k5ChunkTiles + (uint64_t)lane, num_tiles), lane);
        if (lane == 63) cagg[c] = a;
    }
    __syncthreads();
    if (wv != 0) return;
    TileAgg mine = agg_identity();
    if ((uint64_t)lane < chunks) mine = cagg[lane];
    const TileAgg inc = wave_scan_combine(mine, lane);
    TileAgg ex = agg_shfl_up(inc, 1);
    if (lane == 0) ex = agg_identity();
    Prefix zero; zero.kept = 0; zero.nals = 0; zero.inside = 0;
    const Prefix p = fold(zero, ex);
    if ((uint64_t)lane < chunks) {
        Pre5 o; o.kept = p.kept; o.nals = p.nals; o.inside = p.inside; o.pad = 0;
        w5.cpre[lane] = o;
    }
}

/* the prefix in front of a tile: in front of its chunk of 64 tiles, then the tiles of the chunk in front of it (every lane calls) */
__device__ __forceinline__ Prefix tile_prefix5(const Ws5& w5, uint64_t tile, int lane)
{
    const uint64_t c = tile / k5ChunkTiles, t0 = c * k5ChunkTiles;
    TileAgg mine = agg_identity();
    if (t0 + (uint64_t)lane < tile) mine = w5.tagg[t0 + (uint64_t)lane];
    const TileAgg infront = agg_readlane(wave_scan_combine(mine, lane), 63);
    const Pre5 cp = w5.cpre[c];
    Prefix ex; ex.kept = cp.kept; ex.nals = cp.nals; ex.inside = cp.inside;
    return prefix_uniform4(fold(ex, infront));
}

/* What the emit pass walks by rows (ONE place in the code, and since round 6 a function of its own: inlined, the walk's registers
 * were the kernel's -- 162 VGPRs and 193 spilled SGPRs where round 4's kernel had 143 and 33 -- on a path that a launch without
 * such tiles never takes): first the wavefront's own job, if its tile has one -- part 0 of a tile the stream pass walked by rows
 * (`own_rows` rows of it) --, then the parts 1 .. of ALL such tiles, shared by all wavefronts of the launch once their own tile is
 * done (job j = part 1 + j % (nparts - 1) of list entry j / (nparts - 1); this wavefront takes j = its number, + the grid, ...:
 * consecutive wavefronts the parts of one tile).  (The first form had 4 096 extra helper wavefronts for the parts: 1.4 us more on
 * every call.) */
__device__ __forceinline__
void emit_row_jobs5(Lds5& l, const uint8_t* __restrict__ stream, uint64_t n, uint64_t num_tiles, hbs_nal_entry* __restrict__ index, uint64_t index_cap,
                    int rows, void* __restrict__ ws, RunHeader* __restrict__ hdr, int own_rows, uint64_t jobs)
{
    const Ws5 w5 = ws5_carve(ws, num_tiles, rows);
    const int lane = threadIdx.x;
    const uint64_t k5TileBytes = 1024ull * (uint64_t)rows;
    const uint32_t k5RecCap = rec_cap5(rows);
    const uint64_t cut = (n & 15ull) ? (n >> 4) : ~0ull;
    const int nparts = (rows + kPartRows - 1) / kPartRows;
    EmitTarget tgt;
    tgt.index = index; tgt.index_cap = index_cap; tgt.hdr = hdr;
    uint64_t job = blockIdx.x;
#pragma unroll 1
    for (;;) {
        uint64_t tile;
        int p, prows;
        if (own_rows != 0) { tile = blockIdx.x; p = 0; prows = own_rows; own_rows = 0; }
        else {
            if (job >= jobs) break;
            tile = (uint64_t)w5.rwlist[job / (uint64_t)(nparts - 1)];
            p = 1 + (int)(job % (uint64_t)(nparts - 1));
            prows = rows - kPartRows * p < kPartRows ? rows - kPartRows * p : kPartRows;
            job += num_tiles;
        }
        const uint64_t pbase = tile * k5TileBytes + 1024ull * (uint64_t)(kPartRows * p);
        if (pbase >= n) continue;
        const Prefix excl = tile_prefix5(w5, tile, lane);
        TileAgg accb = agg_identity();                              /* the aggregate in front of the part inside its tile */
        if (p != 0) accb = reinterpret_cast<const TileAgg*>(&w5.rec[tile * k5RecCap])[p - 1];
        /* `prows` rows from `pbase` on, by rows */
        for (int i = prows + lane; i < k5MaxTileRows; i += 64) l.words[i] = 0ull;
        tile_words(l, stream, n, pbase, prows, cut, lane);
        uint64_t prev_end = pbase;
        const RowEdges5 edges = rows_edges(stream, n, pbase, pbase + 1024ull * (uint64_t)prows);
        tile_rows(stream, n, pbase, prows, lane, l, [&](int r, const u32x4& qp, const u32x4& qc, const u32x4& qn) {
            DenseRow d;
            uint32_t quick = 0;
            if (row_visit<true>(d, qp, qc, qn, r, prows, edges, stream, n, pbase, lane, prev_end, quick) == 1) { accb = combine(accb, gap_agg(quick)); return; }   /* nothing to write for such a row */
            const TileAgg ea = wave_scan_combine(elem_agg(d.el.gap, d.el.s), lane);
            TileAgg up = agg_prev_lane(ea);
            if (lane == 0) up = agg_identity();
            const TileAgg eb = combine(accb, up);
            accb = combine(accb, agg_readlane(ea, 63));
            if (d.el.v.g0 < n) elem_emit(d.el, eb, excl, false, nullptr, tgt, &l.seg_dummy[lane]);
        });
        __builtin_amdgcn_wave_barrier();                           /* l is reused by the next part */
    }
}

template <int kRows>
__global__ __launch_bounds__(64)
void k_index5_emit(const uint8_t* __restrict__ stream, uint64_t n, uint64_t num_tiles,
                   hbs_nal_entry* __restrict__ index, uint64_t index_cap, int rows_arg, void* __restrict__ ws, RunHeader* __restrict__ hdr, int gate)
{
    if (gate_closed(gate, hdr)) return;
    __shared__ Lds5 l;
    const int rows = kRows ? kRows : rows_arg;
    const Ws5 w5 = ws5_carve(ws, num_tiles, rows);
    const int lane = threadIdx.x;
    const uint64_t k5TileBytes = 1024ull * (uint64_t)rows;
    const uint32_t k5RecCap = rec_cap5(rows), wpl = words_per_lane5(rows);
    const uint64_t cut = (n & 15ull) ? (n >> 4) : ~0ull;
    EmitTarget tgt;
    tgt.index = index; tgt.index_cap = index_cap; tgt.hdr = hdr;
    const int nparts = (rows + kPartRows - 1) / kPartRows;
    /* (a launch without tiles walked by rows has no jobs: nothing but one load of the count) */
    const uint64_t jobs = (uint64_t)hdr->rewalk_count * (uint64_t)(nparts - 1);
    int own_rows = 0;
    {
        const uint64_t tile = blockIdx.x;
        const uint64_t base = tile * k5TileBytes;
        const uint32_t nrec = w5.nrec[tile];
        if (tile == num_tiles - 1) {
            const Prefix incl = fold(tile_prefix5(w5, tile, lane), w5.tagg[tile]);       /* (every lane) */
            if (lane == 0) { hdr->final_kept = incl.kept; hdr->final_nals = incl.nals; hdr->final_inside = incl.inside; }
        }
        if (nrec == k5RewalkParts) {                               /* walked by rows: my part is the first */
            own_rows = rows < kPartRows ? rows : kPartRows;
        } else if (nrec != k5Rewalk) {
            const Prefix excl = tile_prefix5(w5, tile, lane);
            TileAgg acc = agg_identity();
#pragma unroll 1
            for (uint32_t p0 = 0; p0 < nrec; p0 += 64u) {
                Elem el;
                const bool have = p0 + (uint32_t)lane < nrec;
                TileAgg ea = agg_identity();
                if (have) { rec_load(&w5.rec[tile * k5RecCap + p0 + (uint32_t)lane], el, base, n); ea = elem_agg(el.gap, el.s); }
                ea = wave_scan_combine(ea, lane);
                TileAgg up = agg_prev_lane(ea);
                if (lane == 0) up = agg_identity();
                const TileAgg e = combine(acc, up);
                acc = combine(acc, agg_readlane(ea, 63));
                if (have) elem_emit(el, e, excl, false, nullptr, tgt, &l.seg_dummy[lane]);
            }
        } else {
            /* the tile again, the prefix known: its elements 64 at a time (more of them than are recorded) */
            for (int i = rows + lane; i < k5MaxTileRows; i += 64) l.words[i] = 0ull;
            tile_words(l, stream, n, base, rows, cut, lane);
            bool by_rows;
            const uint32_t nelem = tile_census(l, lane, wpl, by_rows);
            if (by_rows) {                                          /* (the stream pass's census said otherwise: not expected) */
                own_rows = rows;
                if (HBS5_ROWS_KERNEL && lane == 0) w5.nrec[tile] = k5RewalkWhole;
            } else {
                const Prefix excl = tile_prefix5(w5, tile, lane);
                TileAgg accb = agg_identity();
                uint64_t prev_end = base;
                const uint32_t npass = (nelem + 63u) >> 6;
#pragma unroll 1
                for (uint32_t p = 0; p < npass; ++p) {
                    Elem el;
                    TileAgg ea = make_batch(el, l, 64u * p, nelem, lane, stream, base, n, prev_end, wpl);
                    ea = wave_scan_combine(ea, lane);
                    TileAgg up = agg_prev_lane(ea);
                    if (lane == 0) up = agg_identity();
                    const TileAgg eb = combine(accb, up);
                    accb = combine(accb, agg_readlane(ea, 63));
                    if (64u * p + (uint32_t)lane < nelem) elem_emit(el, eb, excl, false, nullptr, tgt, &l.seg_dummy[lane]);
                }
            }
        }
    }
#if !HBS5_ROWS_KERNEL
    if (own_rows == 0 && jobs == 0) return;
    __builtin_amdgcn_wave_barrier();                               /* l is reused */
    emit_row_jobs5(l, stream, n, num_tiles, index, index_cap, rows, ws, hdr, own_rows, jobs);
#endif
}

#if HBS5_ROWS_KERNEL
/* the row jobs as a launch of their own behind k_index5_emit (same grid: a wavefront per tile; all leave at once when no tile
 * was walked by rows) */
__global__ __launch_bounds__(64)
void k_index5_emit_rows(const uint8_t* __restrict__ stream, uint64_t n, uint64_t num_tiles,
                        hbs_nal_entry* __restrict__ index, uint64_t index_cap, int rows, void* __restrict__ ws, RunHeader* __restrict__ hdr, int gate)
{
    if (gate_closed(gate, hdr)) return;
    __shared__ Lds5 l;
    const Ws5 w5 = ws5_carve(ws, num_tiles, rows);
    const int nparts = (rows + kPartRows - 1) / kPartRows;
    const uint64_t jobs = (uint64_t)hdr->rewalk_count * (uint64_t)(nparts - 1);
    const uint32_t nrec = w5.nrec[blockIdx.x];
    const int own_rows = nrec == k5RewalkParts ? (rows < kPartRows ? rows : kPartRows) : nrec == k5RewalkWhole ? rows : 0;
    if (own_rows == 0 && jobs == 0) return;
    emit_row_jobs5(l, stream, n, num_tiles, index, index_cap, rows, ws, hdr, own_rows, jobs);
}
#endif

/* ---- host side ---------------------------------------------------------------------------- */

/* Tile height (KiB), grid and schedule for a stream of n bytes on at most `waves` resident wavefronts.  From 4 GiB up: tiles of
 * k5TileRowsLarge by ticket on every wavefront.  Below: the fewest rounds r such that a tile fits the LDS mask (k5MaxTileRows),
 * the tile height that gives EVERY resident wavefront exactly r tiles, grid = tiles / r, wavefront w takes tiles w, w + grid, ...
 * (Fewer, larger tiles -- 2048 wavefronts x 512 KiB on 1 GiB -- measured 3 % faster still on ~10 KiB NALs and 20 % slower on
 * 0.5-1 KiB NALs, where the element batches make the kernel VALU-bound and a third fewer wavefronts is a third less of that:
 * the host cannot know which it is.) */
Geo5 scan5_geometry(uint64_t n, uint64_t waves)
{
    Geo5 g;
    g.rows = k5TileRowsLarge; g.strided = 0; g.grid = waves;
    if (n >= (4ull << 30) || waves == 0) { g.tiles = (n + 1024ull * (uint64_t)g.rows - 1) / (1024ull * (uint64_t)g.rows); if (g.grid > g.tiles) g.grid = g.tiles; return g; }
    const uint64_t cap = waves * 1024ull * (uint64_t)k5MaxTileRows;
    const uint64_t rounds = n ? (n + cap - 1) / cap : 1;
    if (rounds >= 2) {
        /* More than one tile a wavefront (1.5 - 4 GiB): a WORKGROUP per tile of k5TileRowsLarge rows, the grid = the tiles, and the
         * hardware's dispatcher deals them out as wavefronts leave -- dynamic balance without a ticket (round 5).  On uniform
         * streams the same time as whole rounds of taller tiles (2 GiB: 0.420-0.428 ms either way); a stream with stretches that
         * are walked by rows (the bench's mixed stream) 1.14-1.19 x its uniform time instead of 1.46-1.54 x: the wavefront that
         * has such a tile is no longer the one the launch waits for with a second tile still to do.  With ONE tile a wavefront
         * (below 1.5 GiB) a workgroup per smaller tile measured 1.5-3 % slower on uniform streams (BASELINE's config 2), 1.24-1.28 x
         * instead of 1.65-1.71 x on the mixed one: not taken (scripts/experiments/README.md). */
        g.tiles = (n + 1024ull * (uint64_t)g.rows - 1) / (1024ull * (uint64_t)g.rows);
        g.grid = g.tiles;
        g.strided = 1;
        return g;
    }
    const uint64_t per = (n + waves * rounds - 1) / (waves * rounds);
    uint64_t rows = (per + k5SpanBytes - 1) / k5SpanBytes * (uint64_t)k5SpanRows;
    if (rows < (uint64_t)k5MinTileRows) rows = k5MinTileRows;
    if (rows > (uint64_t)k5MaxTileRows) rows = k5MaxTileRows;
    g.rows = (int)rows;
    g.tiles = (n + 1024ull * rows - 1) / (1024ull * rows);
    const uint64_t r2 = g.tiles ? (g.tiles + waves - 1) / waves : 1;
    g.grid = g.tiles ? (g.tiles + r2 - 1) / r2 : 0;
    g.strided = 1;
    return g;
}
int scan5_tile_rows(uint64_t n, uint64_t waves) { return scan5_geometry(n, waves).rows; }

void launch_scan_index5(const ScanArgs& a, int gate, hipStream_t st)
{
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    static const int per_cu = [] {                             /* what is resident: 168 VGPRs per lane (held there by amdgpu_waves_per_eu: 7 per-tile values live in scratch) -> 3 wavefronts per SIMD (8 or 16 per CU measured slower) */
        const char* e = getenv("HBS5_WAVES_PER_CU");            /* debugging aid */
        return e && atoi(e) > 0 && atoi(e) <= 32 ? atoi(e) : 12;
    }();
    static const int env_rows = [] {                           /* debugging aid: a tile height for every call */
        const char* e = getenv("HBS5_TILE_ROWS");
        const int v = e ? atoi(e) : 0;
        return (v >= k5MinTileRows && v <= k5MaxTileRows && v % k5SpanRows == 0) ? v : 0;
    }();
    uint64_t waves = (uint64_t)cus * (uint64_t)per_cu;
    /* hbs_ctx_reserve_workgroups: a reserved slot is four wavefronts' worth of registers (one 256-thread workgroup of the
     * event-sparse kernel); these one-wavefront workgroups fill every SIMD otherwise */
    if (a.spare_wgs > 0) waves = waves > 4ull * (uint64_t)a.spare_wgs + 64 ? waves - 4ull * (uint64_t)a.spare_wgs : 64;
    Geo5 g = scan5_geometry(a.n, waves);
    static const int env_ticket = [] { const char* e = getenv("HBS5_FORCE_TICKET"); return e ? atoi(e) : 0; }();   /* debugging aid: 256-row tiles by ticket at any size (round 4's schedule) */
    if (env_ticket) {
        g.rows = k5TileRowsLarge; g.strided = 0;
        g.tiles = (a.n + 1024ull * (uint64_t)g.rows - 1) / (1024ull * (uint64_t)g.rows);
        g.grid = waves < g.tiles ? waves : g.tiles;
    }
    if (env_rows) {                                            /* (the debugging aid: that height, by ticket) */
        g.rows = env_rows; g.strided = 0;
        g.tiles = (a.n + 1024ull * (uint64_t)env_rows - 1) / (1024ull * (uint64_t)env_rows);
        g.grid = waves < g.tiles ? waves : g.tiles;
    }
    const int rows = g.rows;
    const uint64_t num_tiles = g.tiles;
    if (num_tiles == 0) return;
    if (rows == k5TileRowsLarge) k_index5_stream<k5TileRowsLarge><<<dim3((unsigned)(g.grid < 1 ? 1 : g.grid)), dim3(64), 0, st>>>(a.stream, a.n, num_tiles, rows, g.strided, a.ws5, a.hdr, gate);
    else k_index5_stream<0><<<dim3((unsigned)(g.grid < 1 ? 1 : g.grid)), dim3(64), 0, st>>>(a.stream, a.n, num_tiles, rows, g.strided, a.ws5, a.hdr, gate);
    if (ws5_chunks(num_tiles) <= (uint64_t)k5FusedChunks) {
        k_index5_chunks_prefix<<<dim3(1), dim3(64 * k5FusedWaves), 0, st>>>(num_tiles, rows, a.ws5, a.hdr, gate);
    } else {
        k_index5_chunks<<<dim3((unsigned)ws5_chunks(num_tiles)), dim3(64), 0, st>>>(num_tiles, rows, a.ws5, a.hdr, gate);
        k_index5_prefix<<<dim3(1), dim3(64), 0, st>>>(num_tiles, rows, a.ws5, a.hdr, gate);
    }
    if (rows == k5TileRowsLarge) k_index5_emit<k5TileRowsLarge><<<dim3((unsigned)num_tiles), dim3(64), 0, st>>>(a.stream, a.n, num_tiles, a.index, a.index_cap, rows, a.ws5, a.hdr, gate);
    else k_index5_emit<0><<<dim3((unsigned)num_tiles), dim3(64), 0, st>>>(a.stream, a.n, num_tiles, a.index, a.index_cap, rows, a.ws5, a.hdr, gate);
#if HBS5_ROWS_KERNEL
    k_index5_emit_rows<<<dim3((unsigned)num_tiles), dim3(64), 0, st>>>(a.stream, a.n, num_tiles, a.index, a.index_cap, rows, a.ws5, a.hdr, gate);
#endif
}

} // namespace hbs
