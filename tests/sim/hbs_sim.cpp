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
#include "../../hevcbitstream_amd/csrc/hbs_chunk.h"
#include "../../hevcbitstream_amd/csrc/hbs_sparse.h"
#include "../../hevcbitstream_amd/csrc/hbs_emit.h"
#include "../../hevcbitstream_amd/csrc/hbs_parse.h"
#include "../../hevcbitstream_amd/csrc/hbs_parse_fix.h"
#include "../../hevcbitstream_amd/csrc/hbs_parse_ext.h"
#include "../../hevcbitstream_amd/csrc/hbs_ingest.h"

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

/* k3_tiles' dense tiles: the chunk algebra of hbs_emit.h (dz_fast4, dz_one_start4, dz_map) stepped chunk by chunk over `nchunks`
 * chunks of 16 bytes -- what k3_dense_tile's first half adds up, with its wavefronts' ballots and shuffles replaced by a loop.
 * start_at[c]: the byte of chunk c at which a NAL begins (-1: none; at most one a chunk, as the closed form asks), gaps[c] its
 * start code's bytes.  tot[h] = bytes that go in when the first chunk is entered with count h, st[h] = the count behind the last. */
extern "C" void sim_dz_walk(const uint8_t* bytes, uint64_t nchunks, const int8_t* start_at, const uint32_t* gaps, uint32_t* tot, uint32_t* st)
{
    using namespace hbs;
    for (uint32_t h = 0; h < 3; ++h) { tot[h] = 0; st[h] = h; }
    for (uint64_t c = 0; c < nchunks; ++c) {
        uint32_t w[4];
        for (int d = 0; d < 4; ++d)
            w[d] = (uint32_t)bytes[16 * c + 4 * d] | ((uint32_t)bytes[16 * c + 4 * d + 1] << 8) | ((uint32_t)bytes[16 * c + 4 * d + 2] << 16) |
                   ((uint32_t)bytes[16 * c + 4 * d + 3] << 24);
        const DzFast f = start_at[c] >= 0 ? dz_one_start4(w[0], w[1], w[2], w[3], (uint32_t)start_at[c], gaps[c]) : dz_fast4(w[0], w[1], w[2], w[3]);
        for (uint32_t h = 0; h < 3; ++h) {
            const uint32_t ci = st[h];
            tot[h] += ci == 0 ? f.i0 : ci == 1 ? f.i1 : f.i2;
            st[h] = f.reset ? f.out : dz_map(ci);
        }
    }
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
        /* as the kernels do it: 16-byte chunks, the byte-exact rules only where chunk_flag() says
         * two adjacent zeros may sit in front of one of the chunk's bytes */
        auto nal_dword = [&](int64_t o) {          /* bytes of this NAL only; 0xFF outside */
            uint32_t v = 0;
            for (int i = 0; i < 4; ++i) {
                const int64_t q = o + i;
                v |= (uint32_t)((q >= 0 && q < (int64_t)len) ? rbsp[begin + q] : 0xFF) << (8 * i);
            }
            return v;
        };
        for (uint32_t off = 0; off < len; off += 16) {
            const uint64_t sb = begin + off, se = begin + (off + 16 < len ? off + 16 : len);
            const bool f = chunk_flag(nal_dword((int64_t)off - 4), nal_dword(off), nal_dword(off + 4), nal_dword(off + 8),
                                      nal_dword(off + 12), nal_dword(off + 16));
            const uint32_t c = count_segment(rbsp, begin, sb, se);
            {   /* the register form the kernels run must say the same: count, mask, bytes */
                uint32_t lc = off ? lead_count4(nal_dword((int64_t)off - 4)) : 0u;
                if (lc == kLeadUnknown) lc = lead_count(rbsp, begin, sb);
                if (lc != lead_count(rbsp, begin, sb)) return -3;
                const uint32_t nb = (uint32_t)(se - sb);
                const uint32_t w0 = nal_dword(off), w1 = nal_dword(off + 4), w2 = nal_dword(off + 8), w3 = nal_dword(off + 12);
                const uint32_t m = insert_mask16(w0, w1, w2, w3, nb, lc);
                if ((uint32_t)__builtin_popcount(m) != c) return -3;
                uint8_t a[32], b[32];
                const uint32_t na = emit_chunk16(a, w0, w1, w2, w3, nb, m);
                emit_segment(rbsp, begin, sb, se, b);
                if (na != nb + c || memcmp(a, b, na) != 0) return -3;
            }
            if (!f) {
                if (c != 0) return -2;             /* the test must be conservative */
                memcpy(out + dst, rbsp + sb, se - sb);
            } else {
                emit_segment(rbsp, begin, sb, se, out + dst);
            }
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

/* K4 in stream order: plan, context resolution and the per-NAL parse of hbs_parse.hip */
static TraceRec* g_sim_trace = nullptr;       /* set by sim_parse_trace around sim_parse_headers */
static uint32_t g_sim_trace_cap = 0;
static uint32_t* g_sim_trace_count = nullptr;

static uint8_t* g_sim_state_out = nullptr;     /* when set: the derived tables behind the last NAL go here (sizeof(RpsTables)) */
static int g_sim_state_ok = 0;
extern "C" void sim_parse_set_state_out(uint8_t* p) { g_sim_state_out = p; }
extern "C" int sim_parse_state_ok() { return g_sim_state_ok; }
extern "C" uint64_t sim_rps_tables_bytes() { return sizeof(RpsTables); }
static int g_sim_fix_mode = 0;                 /* 0: the batch parse alone; 1: + the exact re-walk when a slice raised the flag; 2: + always */
static int g_sim_fix_stats[3];                 /* a slice raised the flag / slices walked again / chains that were too deep */
extern "C" void sim_parse_set_fix(int mode) { g_sim_fix_mode = mode; }
extern "C" void sim_parse_fix_stats(int* out) { out[0] = g_sim_fix_stats[0]; out[1] = g_sim_fix_stats[1]; out[2] = g_sim_fix_stats[2]; }

template <int kMode>
static int64_t sim_parse_impl(const uint8_t* rbsp, const hbs_nal_entry* idx, uint64_t n,
                              ParsedNal* parsed, uint8_t* structs, uint64_t structs_cap)
{
    static std::vector<uint8_t> zeros(sizeof(hevc_sps_t) + 64, 0);
    uint64_t run = 0;
    long long cs = -1, cp = -1;
    std::vector<long long> ctx_sps(n), ctx_pps(n);
    std::vector<uint32_t> deps(n + 1, 0u);
    int any_diverged = 0;
    for (uint64_t k = 0; k < n; ++k) {
        ParsedNal p;
        p.rc = -1; p.nal_unit_type = p.nal_layer_id = p.nal_temporal_id_plus1 = -1;
        p.struct_off = ~0ull; p.slice_data_size = 0; p.slice_data_off = 0;
        uint64_t sz = 0;
        if (!(idx[k].status & HBS_ST_ERROR)) {
            nal_header_of(rbsp + idx[k].rbsp_off, idx[k].rbsp_len, p);
            sz = slot_bytes_of(p.nal_unit_type);
        }
        p.struct_off = sz ? run : ~0ull;
        run += sz;
        ctx_sps[k] = cs; ctx_pps[k] = cp;
        if (p.nal_unit_type == HEVC_NAL_UNIT_TYPE_SPS_NUT) cs = (long long)k;
        if (p.nal_unit_type == HEVC_NAL_UNIT_TYPE_PPS_NUT) cp = (long long)k;
        parsed[k] = p;
    }
    if (!structs) return (int64_t)run;
    for (int pass = 0; pass < 2; ++pass)
        for (uint64_t k = 0; k < n; ++k) {
            const int type = parsed[k].nal_unit_type;
            const bool slice = is_slice_type_nal(type);
            const bool pset = type == HEVC_NAL_UNIT_TYPE_VPS_NUT || type == HEVC_NAL_UNIT_TYPE_SPS_NUT || type == HEVC_NAL_UNIT_TYPE_PPS_NUT;
            if (type < 0 || (!slice && !pset) || ((pass == 0) != pset)) continue;
            const uint64_t off = parsed[k].struct_off, slot = slot_bytes_of(type);
            if (off + slot > structs_cap) { parsed[k].struct_off = ~0ull; continue; }
            uint8_t* dst = structs + off;
            memset(dst, 0, slot);
            ParserT<kMode> ps;
            const uint8_t* src = rbsp + idx[k].rbsp_off;
            ps.b.win = src; ps.b.full = src; ps.b.win_bytes = idx[k].rbsp_len < 512u ? idx[k].rbsp_len : 512u;
            ps.b.size = idx[k].rbsp_len; ps.b.pos = 16;
            ps.b.tr = g_sim_trace ? g_sim_trace + k * (uint64_t)g_sim_trace_cap : nullptr; ps.b.tr_cap = g_sim_trace_cap; ps.b.tr_n = 0; ps.b.wbuf = nullptr;
            ps.sps = nullptr; ps.pps = nullptr; ps.init_rows();
            const hevc_sps_t* zero_sps = reinterpret_cast<const hevc_sps_t*>(zeros.data());
            const hevc_pps_t* zero_pps = reinterpret_cast<const hevc_pps_t*>(zeros.data());
            const hevc_sps_t* last_sps = zero_sps;
            const hevc_pps_t* last_pps = zero_pps;
            RpsRow row;
            if (slice) {
                reinterpret_cast<hevc_slice_header_t*>(dst)->collocated_from_l0_flag = 1;
                if (ctx_sps[k] >= 0 && parsed[ctx_sps[k]].struct_off != ~0ull) {
                    last_sps = reinterpret_cast<const hevc_sps_t*>(structs + parsed[ctx_sps[k]].struct_off);
                    ps.sps_rps = reinterpret_cast<const RpsTables*>(structs + parsed[ctx_sps[k]].struct_off + round16(sizeof(hevc_sps_t)));
                }
                if (ctx_pps[k] >= 0 && parsed[ctx_pps[k]].struct_off != ~0ull)
                    last_pps = reinterpret_cast<const hevc_pps_t*>(structs + parsed[ctx_pps[k]].struct_off);
                memset(&row, 0, sizeof(row));
                ps.own = &row;
            } else if (type == HEVC_NAL_UNIT_TYPE_SPS_NUT) {
                ps.out_rps = reinterpret_cast<RpsTables*>(dst + round16(sizeof(hevc_sps_t)));
            }
            const int consumed = (int)(idx[k].end - idx[k].start) - ((idx[k].status & HBS_ST_TRAILING03) ? 1 : 0);
            ParsedNal out = parsed[k];
            parse_one_nal(ps, type, dst, consumed, &out, last_pps, last_sps, zero_pps, zero_sps);
            parsed[k] = out;
            if (slice) { deps[k] = deps_pack(ps.rec_own, ps.rec_ref, ps.rec_read, ps.own_idx); any_diverged |= ps.diverged; }
            if (g_sim_trace_count) g_sim_trace_count[k] = ps.b.tr_n;
        }
    if (g_sim_state_out) {
        /* what the reference holds behind the last NAL (k4_state): every row of the tables from its last writer */
        std::vector<uint32_t> wmask(n + 1, 0u), bsum(n / kFixBlock + 2, 0u);
        for (uint64_t k = 0; k < n; ++k) { wmask[k] = fix_wmask_of(parsed, structs, deps.data(), k); bsum[k / kFixBlock] |= wmask[k]; }
        FixCtx c;
        c.rbsp = rbsp; c.idx = idx; c.n = n; c.parsed = parsed; c.structs = structs; c.structs_cap = structs_cap;
        c.ctx_sps = ctx_sps.data(); c.ctx_pps = ctx_pps.data(); c.zeros = zeros.data(); c.init_sps_slot = nullptr; c.init_pps = nullptr;
        c.deps = deps.data(); c.wmask = wmask.data(); c.bsum = bsum.data();
        std::vector<RpsRow> temps(kFixDepth);
        std::vector<uint8_t> scratch(sizeof(hevc_slice_header_t) + 64);
        RpsTables* t = reinterpret_cast<RpsTables*>(g_sim_state_out);
        g_sim_state_ok = 1;
        for (int r = 0; r < 32; ++r) {
            RowView v;
            if (!fix_resolve_row(c, r, n, temps.data(), reinterpret_cast<hevc_slice_header_t*>(scratch.data()), v)) { g_sim_state_ok = 0; continue; }
            t->NumDeltaPocs[r] = v.nd; t->NumNegativePics[r] = v.nn; t->NumPositivePics[r] = v.np;
            for (int j = 0; j < 32; ++j) { t->DeltaPocS0[r][j] = v.s0[j]; t->UsedByCurrPicS0[r][j] = v.u0[j]; t->DeltaPocS1[r][j] = v.s1[j]; t->UsedByCurrPicS1[r][j] = v.u1[j]; }
        }
    }
    /* the exact re-walk of the slices that need it: hbs_parse_fix.h, as k4_fix_masks / k4_fix_list / k4_fix run it */
    g_sim_fix_stats[0] = any_diverged; g_sim_fix_stats[1] = 0; g_sim_fix_stats[2] = 0;
    if (g_sim_fix_mode && (any_diverged || g_sim_fix_mode == 2)) {          /* mode 2: look for affected slices even when no slice raised the flag (is the flag complete?) */
        std::vector<uint32_t> wmask(n + 1, 0u), bsum(n / kFixBlock + 2, 0u);
        for (uint64_t k = 0; k < n; ++k) { wmask[k] = fix_wmask_of(parsed, structs, deps.data(), k); bsum[k / kFixBlock] |= wmask[k]; }
        FixCtx c;
        c.rbsp = rbsp; c.idx = idx; c.n = n; c.parsed = parsed; c.structs = structs; c.structs_cap = structs_cap;
        c.ctx_sps = ctx_sps.data(); c.ctx_pps = ctx_pps.data(); c.zeros = zeros.data(); c.init_sps_slot = nullptr; c.init_pps = nullptr;
        c.deps = deps.data(); c.wmask = wmask.data(); c.bsum = bsum.data();
        std::vector<uint64_t> list;
        for (uint64_t k = 0; k < n; ++k) if (fix_is_affected(c, k)) list.push_back(k);
        g_sim_fix_stats[1] = (int)list.size();
        std::vector<RpsRow> temps(kFixTemps);
        for (uint64_t k : list) {
            uint32_t tr_n = 0;
            const bool ok = fix_slice<kMode>(c, k, temps.data(), g_sim_trace ? g_sim_trace + k * (uint64_t)g_sim_trace_cap : nullptr, g_sim_trace_cap, &tr_n);
            if (!ok) g_sim_fix_stats[2] += 1;
            else if (g_sim_trace_count) g_sim_trace_count[k] = tr_n;
        }
    }
    return (int64_t)run;
}

extern "C" int64_t sim_parse_headers(const uint8_t* rbsp, const hbs_nal_entry* idx, uint64_t n,
                                     ParsedNal* parsed, uint8_t* structs, uint64_t structs_cap)
{
    return sim_parse_impl<kModeRead>(rbsp, idx, n, parsed, structs, structs_cap);
}

extern "C" int64_t sim_parse_trace(const uint8_t* rbsp, const hbs_nal_entry* idx, uint64_t n, ParsedNal* parsed, uint8_t* structs,
                                   uint64_t structs_cap, TraceRec* trace, uint32_t trace_cap, uint32_t* trace_count)
{
    g_sim_trace = trace; g_sim_trace_cap = trace_cap; g_sim_trace_count = trace_count;
    for (uint64_t k = 0; k < n; ++k) trace_count[k] = 0;
    const int64_t r = sim_parse_impl<kModeTrace>(rbsp, idx, n, parsed, structs, structs_cap);
    g_sim_trace = nullptr; g_sim_trace_cap = 0; g_sim_trace_count = nullptr;
    return r;
}

/* event-sparse variant (hbs_scan4.hip): flag test, elements with gaps, segment words -- tile by
 * tile as the kernel does it, the look-back replaced by a running prefix */
extern "C" int sim4_index_extract(const uint8_t* stream, uint64_t n,
                                  hbs_nal_entry* index, uint64_t index_cap,
                                  uint8_t* rbsp, uint64_t rbsp_cap, hbs_summary* sum)
{
    RunHeader hdr;
    memset(&hdr, 0, sizeof(hdr));
    hdr.first_empty = ~0ull;
    if (index_cap) memset(index, 0, index_cap * sizeof(hbs_nal_entry));
    EmitTarget tgt{index, index_cap, &hdr};
    Prefix run{0, 0, 0};
    auto dword_at = [&](int64_t q) {
        uint32_t v = 0;
        for (int i = 0; i < 4; ++i) v |= (uint32_t)byte_at(stream, q + i, n) << (8 * i);
        return v;
    };
    auto view_at = [&](uint64_t g0) {
        ElemView v;
        v.xpp = dword_at((int64_t)g0 - 8);
        v.xp = dword_at((int64_t)g0 - 4); v.x0 = dword_at((int64_t)g0); v.x1 = dword_at((int64_t)g0 + 4);
        v.x2 = dword_at((int64_t)g0 + 8); v.x3 = dword_at((int64_t)g0 + 12); v.xn = dword_at((int64_t)g0 + 16);
        v.stream = stream; v.g0 = g0; v.n = n;
        return v;
    };
    const uint64_t num_tiles = (n + k4TileBytes - 1) / k4TileBytes;
    std::vector<uint32_t> list, seg;
    std::vector<uint8_t> flagged(k4ChunksPerTile);
    for (uint64_t tile = 0; tile < num_tiles; ++tile) {
        const uint64_t base = tile * (uint64_t)k4TileBytes, tile_end = base + k4TileBytes;
        list.clear();
        for (uint32_t c = 0; c < (uint32_t)k4ChunksPerTile; ++c) {
            const uint64_t g = base + 16ull * c;
            const ElemView v = view_at(g);
            bool f = chunk_flag(v.xp, v.x0, v.x1, v.x2, v.x3, v.xn);
            if (!f && g < n && chunk_patterns(v.xp, v.x0, v.x1, v.x2, v.x3, v.xn) != 0) return -210;   /* the test must be conservative */
            f = f || (g < n && n < g + 16);
            flagged[c] = f;
            if (f) list.push_back(c);
        }
        /* phase 0: tile aggregate */
        TileAgg acc = agg_identity();
        for (size_t i = 0; i < list.size(); ++i) {
            const uint64_t prev_end = i ? base + 16ull * (list[i - 1] + 1u) : base;
            const ElemView v = view_at(base + 16ull * list[i]);
            ChunkMarks m; BlockSum s; ElemClasses cls;
            elem_walk(v, m, s, cls);
            {   /* the bit-parallel walk against the one-pattern-at-a-time rules of hbs_tile.h */
                ChunkMarks mg; BlockSum sg;
                elem_walk_generic(v, mg, sg);
                if (m.cand != mg.cand || m.ev != mg.ev || m.ev_start != mg.ev_start || m.err != mg.err ||
                    s.cnt != sg.cnt || s.known != sg.known || s.carry != sg.carry || s.last != sg.last) return -213;
            }
            acc = combine(acc, elem_agg(span_bytes(prev_end, v.g0, n), s));
        }
        const uint64_t last_end = list.empty() ? base : base + 16ull * (list.back() + 1u);
        const TileAgg tagg = combine(acc, gap_agg(span_bytes(last_end, tile_end, n)));
        const Prefix excl = run;
        run = fold(run, tagg);
        const uint32_t tile_kept = tagg.known + (excl.inside ? tagg.sig : 0u);
        const bool can_store = rbsp != nullptr && excl.kept + tile_kept <= rbsp_cap;
        if (rbsp != nullptr && !can_store) flag_error(&hdr, (uint32_t)(-HBS_E_CAPACITY));
        uint8_t* out = rbsp + excl.kept;
        /* phase 1: elements */
        seg.assign(1, seg_pack(-1, 0u, excl.inside != 0u));
        TileAgg accb = agg_identity();
        for (size_t i = 0; i < list.size(); ++i) {
            const uint64_t prev_end = i ? base + 16ull * (list[i - 1] + 1u) : base;
            const ElemView v = view_at(base + 16ull * list[i]);
            ChunkMarks m; BlockSum s; ElemClasses cls;
            {   /* the kernel parks marks and summary in LDS between the two phases */
                ChunkMarks m0; BlockSum s0;
                elem_walk(v, m0, s0, cls);
                elem_unpack(elem_pack(m0, s0), m, s);
                if (m.cand != m0.cand || m.ev != m0.ev || m.ev_start != m0.ev_start || m.err != m0.err ||
                    s.cnt != s0.cnt || s.known != s0.known || s.carry != s0.carry || s.last != s0.last) return -212;
            }
            const uint32_t gap = span_bytes(prev_end, v.g0, n);
            const ElemStart st = elem_start(accb, gap, excl.inside);
            /* the emit half with class masks against emit_block_t: the generic one runs first, on a copy of the entries it may touch */
            const uint64_t k0 = excl.nals + accb.cnt, lo = k0 ? k0 - 1 : 0, hi = k0 + 18 < index_cap ? k0 + 18 : index_cap;
            std::vector<hbs_nal_entry> scratch;
            for (uint64_t k = lo; k < hi; ++k) scratch.push_back(index[k]);
            if (scratch.empty()) scratch.push_back(hbs_nal_entry{0, 0, 0, 0, 0});
            RunHeader h2 = hdr;
            EmitTarget tgt2{scratch.data() - lo, index_cap, &h2};
            const uint32_t keep_g = (uint32_t)emit_block_t<kChunk, ElemView, uint32_t>(v, 0, v.g0, m, st.inside, k0, excl.kept + st.kept, tgt2);
            const uint32_t keep = emit_chunk_fast(cls, v.g0, m, st.inside, k0, excl.kept + st.kept, tgt);
            if (keep != keep_g || h2.first_empty != hdr.first_empty || h2.error != hdr.error) return -214;
            for (uint64_t k = lo; k < hi; ++k)
                if (memcmp(&scratch[k - lo], &index[k], sizeof(hbs_nal_entry)) != 0) return -215;
            const uint32_t nk = (uint32_t)__builtin_popcount(keep);
            if (can_store && keep) {
                uint64_t lo, hi;
                const uint32_t cnt = compact_chunk_regs(v.x0, v.x1, v.x2, v.x3, keep, lo, hi);
                for (uint32_t b = 0; b < cnt; ++b) out[st.kept + b] = (uint8_t)(((b < 8) ? lo : hi) >> (8 * (b & 7)));
            }
            const bool after = (s.last != kKindNone) ? (s.last == kKindStart) : st.inside;
            seg.push_back(seg_pack((int32_t)list[i], st.kept + nk, after));
            accb = combine(accb, elem_agg(gap, s));
        }
        /* copy of everything else */
        uint32_t k = 0, copied_end = 0;
        for (uint32_t c = 0; c < (uint32_t)k4ChunksPerTile; ++c) {
            if (flagged[c]) { ++k; continue; }
            const uint64_t g = base + 16ull * c;
            if (g + 16 > n) continue;
            const uint32_t w = seg[k];
            if (!seg_inside(w)) continue;
            const int64_t rank = (int64_t)seg_bias(w) + 16 * (int64_t)c;
            if (rank < 0 || (uint64_t)rank + 16 > tile_kept) return -211;
            if (can_store) memcpy(out + rank, stream + g, 16);
            copied_end = (uint32_t)rank + 16;
        }
        (void)copied_end;
    }
    hdr.final_kept = run.kept; hdr.final_nals = run.nals; hdr.final_inside = run.inside;
    uint8_t tail[8];
    for (int i = 0; i < 8; ++i) tail[i] = byte_at(stream, (int64_t)n - 8 + i, n);
    tail_fixup(&hdr, index, index_cap, rbsp, rbsp_cap, tail, n, sum);
    for (uint64_t k = 0; k < hdr.final_nals; ++k) fill_rbsp_len(&hdr, index, index_cap, k);
    return 0;
}

/* register-resident variant (hbs_scan3.hip): the per-chunk logic of hbs_chunk.h in stream order */
extern "C" int sim3_index_extract(const uint8_t* stream, uint64_t n,
                                  hbs_nal_entry* index, uint64_t index_cap,
                                  uint8_t* rbsp, uint64_t rbsp_cap, hbs_summary* sum)
{
    RunHeader hdr;
    memset(&hdr, 0, sizeof(hdr));
    hdr.first_empty = ~0ull;
    if (index_cap) memset(index, 0, index_cap * sizeof(hbs_nal_entry));
    EmitTarget tgt{index, index_cap, &hdr};
    Prefix run{0, 0, 0};
    auto dword_at = [&](int64_t q) {
        uint32_t v = 0;
        for (int i = 0; i < 4; ++i) v |= (uint32_t)byte_at(stream, q + i, n) << (8 * i);
        return v;
    };
    for (uint64_t g0 = 0; g0 < n; g0 += kChunk) {
        RegView v;
        v.xp = dword_at((int64_t)g0 - 4); v.x0 = dword_at((int64_t)g0); v.x1 = dword_at((int64_t)g0 + 4);
        v.x2 = dword_at((int64_t)g0 + 8); v.x3 = dword_at((int64_t)g0 + 12); v.xn = dword_at((int64_t)g0 + 16);
        v.stream = stream; v.g0 = g0; v.n = n;
        const uint32_t pats = chunk_patterns(v.xp, v.x0, v.x1, v.x2, v.x3, v.xn);
        if (!chunk_maybe_pattern(v.xp, v.x0, v.x1, v.x2, v.x3, v.xn) && pats != 0) return -200;   /* filter must be conservative */
        BlockMarks m;
        BlockSum s;
        walk_block_t<kChunk, RegView>(v, 0, g0, n, pats & 0xFFFFu, pats >> 16, m, s);
        const TileAgg a = as_agg(s);
        const uint64_t keep = emit_block_t<kChunk, RegView>(v, 0, g0, m, run.inside != 0, run.nals, run.kept, tgt);
        const Prefix next = fold(run, a);
        if (popc64(keep) != next.kept - run.kept) return -201;
        if (rbsp != nullptr && keep != 0) {
            if (next.kept <= rbsp_cap) {
                uint64_t lo, hi;
                const uint32_t cnt = compact_chunk_regs(v.x0, v.x1, v.x2, v.x3, (uint32_t)keep, lo, hi);
                for (uint32_t i = 0; i < cnt; ++i) rbsp[run.kept + i] = (uint8_t)(((i < 8) ? lo : hi) >> (8 * (i & 7)));
            } else {
                flag_error(&hdr, (uint32_t)(-HBS_E_CAPACITY));
            }
        }
        run = next;
    }
    hdr.final_kept = run.kept; hdr.final_nals = run.nals; hdr.final_inside = run.inside;
    uint8_t tail[8];
    for (int i = 0; i < 8; ++i) tail[i] = byte_at(stream, (int64_t)n - 8 + i, n);
    tail_fixup(&hdr, index, index_cap, rbsp, rbsp_cap, tail, n, sum);
    for (uint64_t k = 0; k < hdr.final_nals; ++k) fill_rbsp_len(&hdr, index, index_cap, k);
    return 0;
}

/* ---- windowed ingest (hbs_ingest.h) over the CPU single-stepper ----------------------------- */
extern "C" int sim4_index_extract(const uint8_t* stream, uint64_t n, hbs_nal_entry* index, uint64_t index_cap,
                                  uint8_t* rbsp, uint64_t rbsp_cap, hbs_summary* sum);
namespace {
struct SimBackend {
    const uint8_t* h_stream;
    uint8_t* h_rbsp;
    uint64_t lead, window, idx_cap;
    std::vector<uint8_t> buf[2], rbsp;
    std::vector<hbs_nal_entry> index;
    uint64_t fresh[2] = {0, 0};
    uint64_t lead_capacity() const { return lead; }
    uint64_t index_capacity() const { return idx_cap; }
    uint64_t fresh_len(int b) const { return fresh[b]; }
    int begin(uint64_t)
    {
        for (int b = 0; b < 2; ++b) buf[b].assign(lead + window + 256, 0xEE);
        rbsp.assign(lead + window + 256, 0);
        index.resize(idx_cap ? idx_cap : 1);
        return 0;
    }
    int upload(int b, uint64_t dst_off, uint64_t src_lo, uint64_t len)
    {
        fresh[b] = len;
        if (len) memcpy(buf[b].data() + dst_off, h_stream + src_lo, len);
        return 0;
    }
    int carry(int from, uint64_t from_off, int to, uint64_t to_off, uint64_t len)
    {
        if (len) memcpy(buf[to].data() + to_off, buf[from].data() + from_off, len);
        return 0;
    }
    int scan(int b, uint64_t off, uint64_t len, hbs_summary* out)
    {
        /* an exact-size copy: the device code must not look past the window */
        std::vector<uint8_t> w(buf[b].begin() + off, buf[b].begin() + off + len);
        return sim4_index_extract(w.data(), len, index.data(), idx_cap, h_rbsp ? rbsp.data() : nullptr, rbsp.size(), out);
    }
    int fetch_index(uint64_t first, uint64_t count, hbs_nal_entry* dst) { memcpy(dst, index.data() + first, count * sizeof(hbs_nal_entry)); return 0; }
    std::vector<hbs_nal_entry> stage_buf = std::vector<hbs_nal_entry>(7);      /* a small, odd staging: the chunking gets exercised */
    hbs_nal_entry* staging() { return stage_buf.data(); }
    uint64_t staging_entries() const { return stage_buf.size(); }
    int fetch_rbsp(uint64_t off, uint64_t len, uint64_t dst_off) { if (len) memcpy(h_rbsp + dst_off, rbsp.data() + off, len); return 0; }
};
}

extern "C" int sim_index_extract_host(const uint8_t* stream, uint64_t n, uint64_t window_bytes, hbs_nal_entry* index, uint64_t index_cap,
                                      uint8_t* rbsp, uint64_t rbsp_cap, hbs_summary* sum)
{
    SimBackend be;
    window_bytes &= ~15ull;
    be.h_stream = stream; be.h_rbsp = rbsp; be.window = window_bytes; be.lead = window_bytes;
    be.idx_cap = (2 * window_bytes) / 32 + 64;
    return hbs::ingest_windowed(be, n, window_bytes, index, index_cap, rbsp != nullptr, rbsp_cap, sum);
}

/* ---- syntax writers (write_one_nal) single-stepped: one NAL from its struct --------------------- */
extern "C" int sim_write_nal(int type, int layer, int tid, uint8_t* slot /* struct; an SPS is followed by its RpsTables */,
                             const uint8_t* sps_slot /* SPS + tables in force, or null */, const uint8_t* pps /* or null */,
                             uint8_t* out, uint32_t cap, WrittenNal* res)
{
    static std::vector<uint8_t> zeros(sizeof(hevc_sps_t) + 64, 0);
    memset(out, 0, cap);
    ParserT<kModeWrite> ps;
    ps.b.win = out; ps.b.full = out; ps.b.win_bytes = 0; ps.b.size = cap; ps.b.pos = 0;
    ps.b.tr = nullptr; ps.b.tr_cap = 0; ps.b.tr_n = 0; ps.b.wbuf = out;
    ps.sps = nullptr; ps.pps = nullptr; ps.init_rows();
    const hevc_sps_t* zero_sps = reinterpret_cast<const hevc_sps_t*>(zeros.data());
    const hevc_pps_t* zero_pps = reinterpret_cast<const hevc_pps_t*>(zeros.data());
    const hevc_sps_t* last_sps = sps_slot ? reinterpret_cast<const hevc_sps_t*>(sps_slot) : zero_sps;
    const hevc_pps_t* last_pps = pps ? reinterpret_cast<const hevc_pps_t*>(pps) : zero_pps;
    RpsRow row;
    if (is_slice_type_nal(type)) {
        if (sps_slot) ps.sps_rps = reinterpret_cast<const RpsTables*>(sps_slot + round16(sizeof(hevc_sps_t)));
        memset(&row, 0, sizeof(row));
        ps.own = &row;
    } else if (type == HEVC_NAL_UNIT_TYPE_SPS_NUT) {
        ps.out_rps = reinterpret_cast<RpsTables*>(slot + round16(sizeof(hevc_sps_t)));
    }
    write_one_nal(ps, type, layer, tid, slot, last_pps, last_sps, zero_pps, zero_sps, res);
    return 0;
}

/* ---- the multi-bit fast paths of the bit reader / writer (hbs_bitfast.h) against bs.h's one bit at a time ---------
 * Random buffers, cursors anywhere (the end of the buffer and beyond included), every width: bits(n), the zero
 * count of ue, put_bits(n, v) must leave the same value, the same cursor and the same bytes as the bit loops they
 * stand in for.  Returns the number of disagreements. */
extern "C" int64_t sim_bitio_check(uint64_t seed, int64_t iterations)
{
    using namespace hbs;
    int64_t bad = 0;
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + 1;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    std::vector<uint8_t> buf(64), wa(64), wb(64);
    for (int64_t it = 0; it < iterations; ++it) {
        const uint32_t size = (uint32_t)(rnd() % 40);
        for (auto& b : buf) { const uint64_t r = rnd(); b = (r & 3) == 0 ? 0 : (uint8_t)(r >> 8); }     /* zero-rich: long ue prefixes */
        const uint32_t pos = (uint32_t)(rnd() % (8 * size + 24));
        const int n = (int)(rnd() % 44);
        BitIOT<kModeRead> a, b;
        a.win = buf.data(); a.full = buf.data(); a.win_bytes = (uint32_t)(rnd() % 2 ? size : 0); a.size = size; a.pos = pos;
        a.tr = nullptr; a.tr_cap = 0; a.tr_n = 0; a.wbuf = nullptr;
        b = a;
        /* bits(n) against n calls of bit() (bs.h:160-169) */
        const uint32_t va = a.bits(n);
        uint32_t vb = 0;
        for (int i = 0; i < n; ++i) vb |= b.bit() << ((uint32_t)(n - i - 1) & 31u);
        if (va != vb || a.pos != b.pos) ++bad;
        /* ue (bs.h:195-207) */
        a.pos = b.pos = pos;
        const uint32_t ua = a.ue_raw();
        int i = 0;
        while ((b.bit() == 0) && (i < 32) && (!b.eof())) ++i;
        uint32_t ub = 0;
        for (int k = 0; k < i; ++k) ub |= b.bit() << ((uint32_t)(i - k - 1) & 31u);
        ub += (1u << (i & 31)) - 1u;
        if (ua != ub || a.pos != b.pos) ++bad;
        /* put_bits(n, v) against n calls of put() (bs.h:240-247) */
        for (size_t k = 0; k < wa.size(); ++k) wa[k] = wb[k] = (uint8_t)rnd();
        BitIOT<kModeWrite> wa_io, wb_io;
        wa_io.win = wa.data(); wa_io.full = wa.data(); wa_io.win_bytes = 0; wa_io.size = size; wa_io.pos = pos;
        wa_io.tr = nullptr; wa_io.tr_cap = 0; wa_io.tr_n = 0; wa_io.wbuf = wa.data();
        wb_io = wa_io; wb_io.win = wb.data(); wb_io.full = wb.data(); wb_io.wbuf = wb.data();
        const uint32_t v = (uint32_t)rnd();
        wa_io.put_bits(n, v);
        for (int k = 0; k < n; ++k) wb_io.put((v >> ((uint32_t)(n - k - 1) & 31u)) & 1u);
        if (wa_io.pos != wb_io.pos || wa != wb) ++bad;
    }
    return bad;
}

/* the opt-in extension's reader (hbs_parse_ext.h) on one NAL's RBSP, as k4_ext runs it per thread */
extern "C" int sim_read_extended_nal(const uint8_t* rbsp, uint32_t size, int consumed, hbs_ext_nal* out, int* nal_unit_type)
{
    hbs::ParsedNal p;
    hbs::nal_header_of(rbsp, size, p);
    *nal_unit_type = p.nal_unit_type;
    if (!hbs::is_extended_nal_type(p.nal_unit_type)) return -2;
    return hbs::read_extended_nal(rbsp, size, p.nal_unit_type, consumed, out);
}
