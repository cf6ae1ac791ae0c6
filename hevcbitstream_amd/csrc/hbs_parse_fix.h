/*
 * hbs_parse_fix.h -- the exact answer for the few slices of a batch whose header depends on more than the SPS in front of them,
 * without walking the batch in order (round 4; until then one such slice sent the whole batch through k4_seq at 81 k NAL/s).
 *
 * What the reference keeps between NALs is ONE set of derived short-term RPS tables, 32 rows (hevc_stream.c:26-32): an SPS
 * writes rows 0 .. num-1 (:61-113 through :1032-1085), a slice that codes its own set writes row num of ITS SPS (:830), and
 * two places read a row: the prediction of a slice's own set from row num - (delta_idx_minus1 + 1) (:1043-1075) and
 * NumPicTotalCurr (:35-59) from row short_term_ref_pic_set_idx (or row num).  The batch parse (k4_parse) gives every slice
 * the rows as its SPS left them plus its own row, which is what the reference computes on every stream the spec allows.
 * It is NOT when the last NAL that wrote a row the slice reads is somebody else: a row index past the SPS's sets, a
 * slice under pps / sps ids other than 0 that wrote its own set into row 0, an IDR that asks for the row a P slice left.
 *
 * The content of a row at a point of the stream is a FUNCTION of the stream, not of a walk: it is what the last NAL in
 * front of that point that wrote the row put there -- an SPS (its snapshot has it), nobody (the caller's initial tables,
 * or zeros), or a slice, whose own row is in turn a function of its bits and of the row it was predicted from at ITS point
 * of the stream.  Which rows a slice writes and reads does not depend on what the rows hold (the indices come from bits in
 * front of every table-dependent length), so every parse records them (deps_pack, hbs_parse.h), and from the records
 *   - fix_wmask_of   says which rows a NAL writes,
 *   - fix_last_writer finds the last writer of a row in front of a NAL (a mask per NAL, an OR per 256 NALs),
 *   - fix_is_affected says whether a slice read a row whose last writer is not the SPS the batch parse assumed,
 *   - fix_slice       evaluates the chain of writers behind each row the slice reads (at most kFixDepth slices deep: each is
 *     walked up to its own set only) and walks the slice again with those rows handed in.
 * Every affected slice is independent of every other one: a lane each.  A chain deeper than kFixDepth reports failure and
 * the batch goes the old way (k4_seq).  Host and device: tests/sim runs the same functions against the oracle.
 */
#ifndef HBS_PARSE_FIX_H
#define HBS_PARSE_FIX_H

#include "hbs_parse.h"

#if defined(__HIPCC__)
#define HBS_FIX_FN __device__ __forceinline__      /* fix_slice calls them from ONE site, in a loop: one copy of the writers' walk */
#else
#define HBS_FIX_FN static
#endif

namespace hbs {

constexpr int kFixDepth = 3;                      /* slices evaluated on the way to one row's content */
constexpr int kFixTemps = 2 * kFixDepth + 1;      /* RpsRows a lane needs: a chain per row it reads, and its own row */
constexpr int kFixBlock = 256;                    /* NALs per summary word */
static_assert(sizeof(RpsTables) <= sizeof(hevc_sps_t), "the zero block of the parse (sizeof(hevc_sps_t) zero bytes) serves as an all-zero set of tables");

struct FixCtx {
    const uint8_t* rbsp; const hbs_nal_entry* idx; uint64_t n;
    ParsedNal* parsed; uint8_t* structs; uint64_t structs_cap;
    const long long* ctx_sps; const long long* ctx_pps;
    const uint8_t* zeros; const uint8_t* init_sps_slot; const uint8_t* init_pps;
    const uint32_t* deps;                         /* per NAL: deps_pack() of a parsed slice, 0 otherwise */
    const uint32_t* wmask;                        /* per NAL: rows it writes */
    const uint32_t* bsum;                         /* per kFixBlock NALs: OR of their masks */
};

/* rows NAL j writes into the reference's tables */
HBS_HD uint32_t fix_wmask_of(const ParsedNal* parsed, const uint8_t* structs, const uint32_t* deps, uint64_t j)
{
    const int t = parsed[j].nal_unit_type;
    if (t == HEVC_NAL_UNIT_TYPE_SPS_NUT) {
        if (parsed[j].struct_off == ~0ull) return 0u;                 /* no room for its struct: it was not parsed at all */
        const int num = reinterpret_cast<const hevc_sps_t*>(structs + parsed[j].struct_off)->num_short_term_ref_pic_sets;
        return num <= 0 ? 0u : (num >= 32 ? ~0u : ((1u << num) - 1u));
    }
    if (is_slice_type_nal(t)) {                                       /* (a slice that was not parsed has no record: deps 0, no row) */
        const int o = deps_own(deps[j]);
        return in32(o) ? (1u << o) : 0u;
    }
    return 0u;
}

/* the last NAL in front of NAL i that wrote row r, -1: none */
HBS_HD long long fix_last_writer(const FixCtx& c, int r, uint64_t i)
{
    const uint32_t bit = 1u << r;
    const uint64_t b0 = i / (uint64_t)kFixBlock;
    for (uint64_t j = i; j > b0 * (uint64_t)kFixBlock; --j)
        if (c.wmask[j - 1] & bit) return (long long)(j - 1);
    for (uint64_t b = b0; b > 0; --b) {
        if (!(c.bsum[b - 1] & bit)) continue;
        for (uint64_t j = b * (uint64_t)kFixBlock; j > (b - 1) * (uint64_t)kFixBlock; --j)
            if (c.wmask[j - 1] & bit) return (long long)(j - 1);
    }
    return -1;
}

/* did the batch parse hand slice i a row whose last writer is somebody else than the SPS in force? */
HBS_HD bool fix_is_affected(const FixCtx& c, uint64_t i)
{
    if (!is_slice_type_nal(c.parsed[i].nal_unit_type)) return false;
    const uint32_t d = c.deps[i];                                      /* 0 for a slice that was not parsed: reads no row */
    /* It counted the pictures of "its own set" without having coded one (an IDR with a P / B slice type): the batch walk read a
     * private, empty row where the reference reads what the last slice that DID code a set left -- in front of the batch too,
     * which no last-writer test in here can see (round 5: a batch that continues a stream, hbs_legacy.c, got zeros there). */
    if ((d >> 18) & 1u) return true;
    const int own = deps_own(d), rows[2] = {deps_ref(d), deps_read(d)};
    for (int q = 0; q < 2; ++q) {
        const int r = rows[q];
        if (r < 0 || (q == 1 && r == own)) continue;                   /* a row it wrote itself, just before */
        const long long w = fix_last_writer(c, r, i);
        if (w != c.ctx_sps[i]) return true;                            /* (-1 == -1: nobody wrote it, and the batch parse used the caller's tables) */
    }
    return false;
}

HBS_HD RowView fix_initial_row(const FixCtx& c, int r)
{
    if (c.init_sps_slot) return view_of_tables(reinterpret_cast<const RpsTables*>(c.init_sps_slot + round16(sizeof(hevc_sps_t))), r);
    return view_of_zeros(reinterpret_cast<const int*>(c.zeros));
}

/* parameter sets in force at NAL k, as the batch parse resolves them */
HBS_HD void fix_context_of(const FixCtx& c, uint64_t k, const hevc_sps_t*& last_sps, const hevc_pps_t*& last_pps)
{
    last_sps = reinterpret_cast<const hevc_sps_t*>(c.zeros);
    last_pps = reinterpret_cast<const hevc_pps_t*>(c.zeros);
    const long long cs = c.ctx_sps[k], cp = c.ctx_pps[k];
    if (cs >= 0) { if (c.parsed[cs].struct_off != ~0ull) last_sps = reinterpret_cast<const hevc_sps_t*>(c.structs + c.parsed[cs].struct_off); }
    else if (c.init_sps_slot) last_sps = reinterpret_cast<const hevc_sps_t*>(c.init_sps_slot);
    if (cp >= 0) { if (c.parsed[cp].struct_off != ~0ull) last_pps = reinterpret_cast<const hevc_pps_t*>(c.structs + c.parsed[cp].struct_off); }
    else if (c.init_pps) last_pps = reinterpret_cast<const hevc_pps_t*>(c.init_pps);
}

/* the own short-term set of slice w into *row: its header walked up to that set, the row it is predicted from handed in */
HBS_FIX_FN void fix_eval_own_row(const FixCtx& c, uint64_t w, int ref_row, const RowView& ref_view, RpsRow* row, hevc_slice_header_t* scratch_sh)
{
    {
        int* z = reinterpret_cast<int*>(row);
        for (uint32_t q = 0; q < (uint32_t)(sizeof(RpsRow) / sizeof(int)); ++q) z[q] = 0;
    }
    const hbs_nal_entry e = c.idx[w];
    ParserT<kModeRead> ps;
    const uint8_t* src = c.rbsp + e.rbsp_off;
    ps.b.win = src; ps.b.full = src; ps.b.win_bytes = 0; ps.b.size = e.rbsp_len; ps.b.pos = 16;
    ps.b.tr = nullptr; ps.b.tr_cap = 0; ps.b.tr_n = 0; ps.b.wbuf = nullptr;
    ps.sps = nullptr; ps.pps = nullptr; ps.init_rows();
    ps.own = row; ps.stop_after_rps = 1;
    ps.sps_rps = reinterpret_cast<const RpsTables*>(c.zeros);       /* the reference's tables always exist; every row this walk reads is handed in */
    if (ref_row >= 0) { ps.ov_idx[0] = ref_row; ps.ov[0] = ref_view; }
    const hevc_sps_t* last_sps; const hevc_pps_t* last_pps;
    fix_context_of(c, w, last_sps, last_pps);
    ps.slice_segment_header(scratch_sh, c.parsed[w].nal_unit_type, last_pps, last_sps,
                            reinterpret_cast<const hevc_pps_t*>(c.zeros), reinterpret_cast<const hevc_sps_t*>(c.zeros));
}

/* what row r holds when NAL i is read.  temps: kFixDepth rows; false: the chain of slices behind it is deeper than that */
HBS_FIX_FN bool fix_resolve_row(const FixCtx& c, int r, uint64_t i, RpsRow* temps, hevc_slice_header_t* scratch_sh, RowView& out)
{
    long long chain[kFixDepth];
    int depth = 0, cr = r;
    uint64_t ci = i;
    RowView cur = view_of_zeros(reinterpret_cast<const int*>(c.zeros));
    for (;;) {
        const long long w = fix_last_writer(c, cr, ci);
        if (w < 0) { cur = fix_initial_row(c, cr); break; }
        if (c.parsed[w].nal_unit_type == HEVC_NAL_UNIT_TYPE_SPS_NUT) {
            cur = view_of_tables(reinterpret_cast<const RpsTables*>(c.structs + c.parsed[w].struct_off + round16(sizeof(hevc_sps_t))), cr);
            break;
        }
        if (depth == kFixDepth) return false;
        chain[depth++] = w;
        const int rr = deps_ref(c.deps[w]);
        if (rr < 0) break;                                              /* coded without prediction: no row behind it */
        cr = rr; ci = (uint64_t)w;
    }
    for (int l = depth - 1; l >= 0; --l) {
        const uint64_t w = (uint64_t)chain[l];
        fix_eval_own_row(c, w, deps_ref(c.deps[w]), cur, &temps[l], scratch_sh);
        cur = view_of_row(&temps[l]);
    }
    out = cur;
    return true;
}

/* Slice i again, exactly.  temps: kFixTemps rows of this lane.  The slice's own struct slot serves as scratch for the writers'
 * headers and is cleared before its own walk (the reference's memset).  false: a chain was too deep, nothing was changed
 * that matters (the batch goes through k4_seq). */
#if defined(__HIPCC__)
#define HBS_FIX_T __device__ __forceinline__
#else
#define HBS_FIX_T static inline
#endif
/* tmp_struct: where a slice WITHOUT a slot in the struct arena (hbs_parse_headers_compact) is walked into: one slice slot owned
 * by the calling lane */
template <int kMode>
HBS_FIX_T bool fix_slice(const FixCtx& c, uint64_t i, RpsRow* temps, TraceRec* trace, uint32_t trace_cap, uint32_t* trace_n, uint8_t* tmp_struct = nullptr)
{
    const uint32_t d = c.deps[i];
    const int own = deps_own(d), ref = deps_ref(d), rd = deps_read(d);
    const int type = c.parsed[i].nal_unit_type;
    uint8_t* const dst = c.parsed[i].struct_off != ~0ull ? c.structs + c.parsed[i].struct_off : tmp_struct;
    hevc_slice_header_t* const sh = reinterpret_cast<hevc_slice_header_t*>(dst);
    const bool rd_foreign = rd >= 0 && rd != own;
    RowView vv[2];
    vv[0] = view_of_zeros(reinterpret_cast<const int*>(c.zeros)); vv[1] = vv[0];
#if defined(__HIPCC__)
#pragma unroll 1
#endif
    for (int q = 0; q < 2; ++q) {                                       /* one call site: one copy of the writers' walk */
        const int r = q == 0 ? ref : (rd_foreign ? rd : -1);
        if (r < 0) continue;
        if (!fix_resolve_row(c, r, i, temps + q * kFixDepth, sh, vv[q])) return false;
    }
    const RowView v0 = vv[0], v1 = vv[1];
    {
        uint32_t* z = reinterpret_cast<uint32_t*>(dst);
        for (uint32_t q = 0; q < (uint32_t)(slot_bytes_of(type) / 4u); ++q) z[q] = 0u;
        int* zr = reinterpret_cast<int*>(&temps[2 * kFixDepth]);
        for (uint32_t q = 0; q < (uint32_t)(sizeof(RpsRow) / sizeof(int)); ++q) zr[q] = 0;
    }
    sh->collocated_from_l0_flag = 1;
    const hbs_nal_entry e = c.idx[i];
    ParserT<kMode> ps;
    const uint8_t* src = c.rbsp + e.rbsp_off;
    ps.b.win = src; ps.b.full = src; ps.b.win_bytes = 0; ps.b.size = e.rbsp_len; ps.b.pos = 16;
    ps.b.tr = trace; ps.b.tr_cap = trace_cap; ps.b.tr_n = 0; ps.b.wbuf = nullptr;
    ps.sps = nullptr; ps.pps = nullptr; ps.init_rows();
    ps.own = &temps[2 * kFixDepth];
    ps.sps_rps = reinterpret_cast<const RpsTables*>(c.zeros);       /* (as above) */
    if (ref >= 0) { ps.ov_idx[0] = ref; ps.ov[0] = v0; }
    if (rd_foreign) { ps.ov_idx[1] = rd; ps.ov[1] = v1; }
    const hevc_sps_t* last_sps; const hevc_pps_t* last_pps;
    fix_context_of(c, i, last_sps, last_pps);
    const int consumed = (int)(e.end - e.start) - ((e.status & HBS_ST_TRAILING03) ? 1 : 0);
    ParsedNal out = c.parsed[i];
    parse_one_nal(ps, type, dst, consumed, &out, last_pps, last_sps,
                  reinterpret_cast<const hevc_pps_t*>(c.zeros), reinterpret_cast<const hevc_sps_t*>(c.zeros));
    c.parsed[i] = out;
    if (trace_n) *trace_n = ps.b.tr_n;
    return true;
}

} // namespace hbs
#endif
