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
#include "../../hevcbitstream_amd/csrc/hbs_emit.h"

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

/* K3 per-segment logic stepped in NAL order (see hbs_emit.hip for the wave-level driver) */
extern "C" int64_t sim_emit_annexb(const uint8_t* rbsp, const hbs_nal_entry* idx, uint64_t n, int gap_mode,
                                   uint8_t* out, uint64_t out_cap, hbs_nal_entry* idx_out)
{
    uint64_t base = 0;
    for (uint64_t k = 0; k < n; ++k) {
        const uint64_t begin = idx[k].rbsp_off;
        const uint32_t len = idx[k].rbsp_len;
        const uint64_t gap = (gap_mode == 1) ? synth_gap(k) : idx[k].start - (k ? idx[k - 1].end : 0ull);
        const uint64_t nal_start = base + gap;
        if (nal_start + len + len / 2 + 2 > out_cap) return -1;
        for (uint64_t i = base; i + 1 < nal_start; ++i) out[i] = 0;
        if (gap) out[nal_start - 1] = 1;
        uint64_t dst = nal_start;
        const uint32_t nseg = (len + kSegBytes - 1) / kSegBytes;
        for (uint32_t s = 0; s < nseg; ++s) {
            const uint64_t sb = begin + (uint64_t)s * kSegBytes;
            const uint64_t se = (s + 1 == nseg) ? begin + len : sb + kSegBytes;
            const uint32_t c = count_segment(rbsp, begin, sb, se);
            emit_segment(rbsp, begin, sb, se, out + dst);
            dst += (se - sb) + c;
        }
        if (idx_out) { idx_out[k] = idx[k]; idx_out[k].start = nal_start; idx_out[k].end = dst; idx_out[k].status = 0; }
        base = dst;
    }
    return (int64_t)base;
}

extern "C" int64_t sim_synth_rbsp(uint64_t seed, uint64_t n, int mode, uint8_t* rbsp, hbs_nal_entry* idx)
{
    uint64_t off = 0;
    for (uint64_t k = 0; k < n; ++k) {
        const uint32_t len = synth_rbsp_len(seed, k);
        for (uint32_t w = 0; w < (len + 7) / 8; ++w) {
            const uint64_t x = synth_rbsp_word(seed, k, w, len, mode);
            for (uint32_t i = 0; i < 8 && 8 * w + i < len; ++i) rbsp[off + 8 * w + i] = (uint8_t)(x >> (8 * i));
        }
        idx[k].start = idx[k].end = 0; idx[k].rbsp_off = off; idx[k].rbsp_len = len; idx[k].status = 0;
        off += len;
    }
    return (int64_t)off;
}
