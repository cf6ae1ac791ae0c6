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
    alignas(16) uint8_t raw[kHalo + kTileBytes + kHalo];
    uint64_t keep[kThreads];
    uint32_t rank[kThreads + 1];
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
    static BlockMarks marks[kThreads];
    static BlockSum sums[kThreads];

    for (uint64_t tile = 0; tile < num_tiles; ++tile) {
        const uint64_t tile_base = tile * (uint64_t)kTileBytes;
        for (int i = -kHalo; i < kTileBytes + kHalo; ++i)
            t.raw[kHalo + i] = byte_at(stream, (int64_t)tile_base + i, n);

        for (int tid = 0; tid < kThreads; ++tid)
            classify_block(&t.raw[kHalo + kBlockBytes * tid], tile_base + (uint64_t)kBlockBytes * tid, n, marks[tid], sums[tid]);

        /* sequential meaning of block_scan() */
        uint32_t in_state[kThreads], pk[kThreads], ps[kThreads], pc[kThreads];
        uint32_t state = 2 /* carry */, k = 0, s = 0, c = 0;
        TileAgg agg{0, 0, 0, kKindNone};
        for (int tid = 0; tid < kThreads; ++tid) {
            in_state[tid] = state; pk[tid] = k; ps[tid] = s; pc[tid] = c;
            k += sums[tid].known + (state == 1 ? sums[tid].carry : 0u);
            s += (state == 2) ? sums[tid].carry : 0u;
            c += sums[tid].cnt;
            if (sums[tid].last != kKindNone) { state = (sums[tid].last == kKindStart) ? 1u : 0u; agg.last = sums[tid].last; }
        }
        agg.known = k; agg.sig = s; agg.cnt = c;

        /* cross-check the aggregate algebra used by the look-back */
        {
            TileAgg viaCombine{0, 0, 0, kKindNone};
            for (int tid = 0; tid < kThreads; ++tid) {
                TileAgg b{sums[tid].cnt, sums[tid].known, sums[tid].carry, sums[tid].last};
                viaCombine = combine(viaCombine, b);
            }
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

        for (int tid = 0; tid < kThreads; ++tid) {
            const bool inside = (in_state[tid] == 1) || (in_state[tid] == 2 && excl.inside);
            const uint32_t rank0 = pk[tid] + (excl.inside ? ps[tid] : 0u);
            t.keep[tid] = emit_block(&t.raw[kHalo + kBlockBytes * tid], tile_base + (uint64_t)kBlockBytes * tid, marks[tid],
                                     inside, excl.nals + pc[tid], excl.kept + rank0, tgt);
            t.rank[tid] = rank0;
        }
        t.rank[kThreads] = tile_kept;

        if (rbsp != nullptr && tile_kept != 0) {
            if (excl.kept + tile_kept <= rbsp_cap) {
                const uint32_t ob = (uint32_t)(excl.kept & 15ull);
                const uint32_t nwords = (ob + tile_kept + 15u) >> 4;
                uint8_t* out = rbsp + (excl.kept - ob);
                for (uint32_t wi = 0; wi < nwords; ++wi) {
                    const GatherOut g = gather_word(&t.raw[kHalo], t.rank, t.keep, wi, ob, tile_kept);
                    for (uint32_t o = g.lo; o < g.hi; ++o) out[16ull * wi + o] = (uint8_t)(g.w[o >> 2] >> (8u * (o & 3u)));
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
