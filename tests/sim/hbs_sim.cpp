/*
 * hbs_sim.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * Compiles the product's per-tile device logic (hevcbitstream_amd/csrc/
 * hbs_tile.h, the code the gfx950 kernels run per thread) with g++ and steps it
 * on the CPU, one "thread" at a time, tiles in stream order.  The workgroup
 * scan and the look-back of hbs_scan.hip are replaced by their sequential
 * meaning (prefix sums / fold).  This lets `-m "not gpu"` tests fuzz the window
 * rules, the end-of-stream rules and the gather against the oracle without a
 * GPU.  It is never linked into, loaded by, or shipped with the product.
 */
#define HBS_HOST_SIM 1
#include <cstring>
#include <vector>
#include "../../hevcbitstream_amd/csrc/hbs_tile.h"

using namespace hbs;

namespace {

uint8_t byte_at(const uint8_t* s, int64_t q, uint64_t n) { return (q >= 0 && (uint64_t)q < n) ? s[q] : 0xFF; }

struct SimTile {
    alignas(16) uint8_t img[kImageBytes];
    uint64_t keep[kBlocks + 1];
    uint32_t rank[kBlocks + 1];
};

} // namespace

extern "C" int sim_index_extract(const uint8_t* stream, uint64_t n,
                                 hbs_nal_entry* index, uint64_t index_cap,
                                 uint8_t* rbsp, uint64_t rbsp_cap, hbs_summary* sum)
{
    RunHeader hdr;
    memset(&hdr, 0, sizeof(hdr));
    hdr.first_empty = ~0ull;
    if (index_cap) memset(index, 0, index_cap * sizeof(hbs_nal_entry));
    EmitTarget tgt{index, index_cap, &hdr};

    const uint64_t num_tiles = (n + kTileBytes - 1) / kTileBytes;
    Prefix run{0, 0, 0};
    static SimTile t;
    static TileAgg sums[kThreads];
    TileView view;
    view.img = t.img;

    for (uint64_t tile = 0; tile < num_tiles; ++tile) {
        const uint64_t tile_base = tile * (uint64_t)kTileBytes;
        memset(t.img, 0xCD, sizeof(t.img));                       /* poison: only staged bytes may be read */
        for (int i = -16; i < kTileBytes + 16; ++i)
            t.img[TileView::phys(i)] = byte_at(stream, (int64_t)tile_base + i, n);

        for (int tid = 0; tid < kThreads; ++tid)
            sums[tid] = classify_thread(view, kThreadBytes * tid, tile_base + (uint64_t)kThreadBytes * tid, n, t.keep, tid);

        /* sequential meaning of block_scan() */
        ThreadStart ts[kThreads];
        uint32_t state = 2 /* carry */, k = 0, s = 0, c = 0;
        TileAgg agg{0, 0, 0, kKindNone};
        for (int tid = 0; tid < kThreads; ++tid) {
            ts[tid].in_state = state; ts[tid].known = k; ts[tid].sig = s; ts[tid].cnt = c;
            k += sums[tid].known + (state == 1 ? sums[tid].sig : 0u);
            s += (state == 2) ? sums[tid].sig : 0u;
            c += sums[tid].cnt;
            if (sums[tid].last != kKindNone) { state = (sums[tid].last == kKindStart) ? 1u : 0u; agg.last = sums[tid].last; }
        }
        agg.known = k; agg.sig = s; agg.cnt = c;

        /* cross-check the aggregate algebra used by the look-back */
        {
            TileAgg viaCombine{0, 0, 0, kKindNone};
            for (int tid = 0; tid < kThreads; ++tid) viaCombine = combine(viaCombine, sums[tid]);
            if (viaCombine.cnt != agg.cnt || viaCombine.known != agg.known || viaCombine.sig != agg.sig || viaCombine.last != agg.last) return -100;
            const TileAgg rt = unpack_agg(pack_agg0(agg), pack_agg1(agg));
            if (rt.cnt != agg.cnt || rt.known != agg.known || rt.sig != agg.sig || rt.last != agg.last) return -101;
        }

        const Prefix excl = run;
        run = fold(run, agg);
        {
            const Prefix rt = unpack_pre(pack_pre0(run), pack_pre1(run));
            if (rt.kept != run.kept || rt.nals != run.nals || rt.inside != run.inside) return -102;
        }
        const uint32_t tile_kept = agg.known + (excl.inside ? agg.sig : 0u);

        /* ascending thread order: thread t reads keep[4t..4t+4] before anyone wrote them */
        for (int tid = 0; tid < kThreads; ++tid)
            emit_thread(view, kThreadBytes * tid, tile_base + (uint64_t)kThreadBytes * tid, n, tid, ts[tid], excl, t.keep, t.rank, tgt);
        t.rank[kBlocks] = tile_kept;

        if (rbsp != nullptr && tile_kept != 0) {
            if (excl.kept + tile_kept <= rbsp_cap) {
                uint8_t* out = rbsp + excl.kept;
                for (uint32_t c = 0; c < (uint32_t)(kTileBytes / 16); ++c) {
                    const ChunkDest d = chunk_dest(t.rank, t.keep, c);
                    if (d.sub == 0) continue;
                    if (d.sub == 0xFFFFu) {
                        for (int i = 0; i < 16; ++i) out[d.rank + i] = (uint8_t)view.byte((int32_t)(16 * c + i));
                    } else {
                        uint64_t lo, hi;
                        const uint32_t cnt = compact_chunk(view, c, d.sub, lo, hi);
                        for (uint32_t i = 0; i < cnt; ++i) out[d.rank + i] = (uint8_t)(((i < 8) ? lo : hi) >> (8 * (i & 7)));
                    }
                }
            } else {
                flag_error(&hdr, (uint32_t)(-HBS_E_CAPACITY));
            }
        }
    }
    hdr.final_kept = run.kept; hdr.final_nals = run.nals; hdr.final_inside = run.inside;

    uint8_t tail[8];
    for (int i = 0; i < 8; ++i) tail[i] = byte_at(stream, (int64_t)n - 8 + i, n);
    tail_fixup(&hdr, index, index_cap, rbsp, rbsp_cap, tail, n, sum);
    for (uint64_t k = 0; k < hdr.final_nals; ++k) fill_rbsp_len(&hdr, index, index_cap, k);
    return 0;
}
