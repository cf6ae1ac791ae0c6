/*
 * hbs_parse.h -- K4: the HEVC header readers as device code (one NAL per lane:
 * a wavefront walks 64 of them in lock step; hbs_parse.hip).  The same walk writes (K5) and traces.
 *
 * Replaces read_hevc_nal_unit's parse half (reference hevc_stream.c:175-239)
 * and the readers it dispatches to: VPS :243-300, SPS :303-401 (+ range ext
 * :404-415), PPS :419-500 (+ range ext :503-521), slice layer :600-617, slice
 * header :782-941, ref_pic_lists_modification :944-966, pred_weight_table
 * :969-1029, st_ref_pic_set :1032-1085 with its derived tables :61-113,
 * VUI :1088-1157, HRD :1160-1218, profile_tier_level :652-755, scaling list
 * :758-779, trailing bits / byte alignment :630-649 -- with the reference's
 * departures from H.265 kept, because a drop-in has to
 * produce the same fields.  The output structs are the ABI-identical ones of
 * include/hevc_stream.h.
 *
 * Differences in shape, not in result:
 *   - the bit reader works on a bit position over the NAL's RBSP (first bytes
 *     staged in LDS) instead of bs.h's pointer + bits_left; its end-of-buffer
 *     rules are those of bs.h:117-207 (zeros past the end, cursor keeps moving,
 *     overrun = a whole byte past the end);
 *   - the parser state the reference keeps in one mutable object (last SPS /
 *     PPS, file-static RPS tables, hevc_stream.c:26-32) is resolved per NAL:
 *     each slice gets the last parameter sets that precede it in the stream
 *     and a private row for its own short-term RPS;
 *   - where the reference indexes out of bounds the same bounded rules as the
 *     test oracle apply (ids other than 0 select an all-zero parameter set, RPS
 *     rows outside 0..31 read as zero, more than 32 entry points are dropped).
 *
 * Compiles for gfx950 and, under tests/sim, for the host.
 */
#ifndef HBS_PARSE_H
#define HBS_PARSE_H

#include "hbs_common.h"
#include "../../include/hevc_stream.h"
#include "hbs_bitfast.h"
namespace hbs {

/* derived short-term RPS tables of one SPS (reference hevc_stream.c:26-32) */
struct RpsTables {
    int NumDeltaPocs[32];
    int NumNegativePics[32];
    int NumPositivePics[32];
    int DeltaPocS0[32][32];
    int UsedByCurrPicS0[32][32];
    int DeltaPocS1[32][32];
    int UsedByCurrPicS1[32][32];
};

/* one row of those tables: what a slice's own st_ref_pic_set derives */
struct RpsRow {
    int NumDeltaPocs, NumNegativePics, NumPositivePics;
    int DeltaPocS0[32], UsedByCurrPicS0[32], DeltaPocS1[32], UsedByCurrPicS1[32];
};

/* one row of the derived tables wherever it lives: a row of an RpsTables, or an RpsRow */
struct RowView { const int* s0; const int* u0; const int* s1; const int* u1; int nd, nn, np; };
HBS_HD RowView view_of_tables(const RpsTables* t, int r)
{
    RowView v;
    v.s0 = t->DeltaPocS0[r]; v.u0 = t->UsedByCurrPicS0[r]; v.s1 = t->DeltaPocS1[r]; v.u1 = t->UsedByCurrPicS1[r];
    v.nd = t->NumDeltaPocs[r]; v.nn = t->NumNegativePics[r]; v.np = t->NumPositivePics[r];
    return v;
}
HBS_HD RowView view_of_row(const RpsRow* q)
{
    RowView v;
    v.s0 = q->DeltaPocS0; v.u0 = q->UsedByCurrPicS0; v.s1 = q->DeltaPocS1; v.u1 = q->UsedByCurrPicS1;
    v.nd = q->NumDeltaPocs; v.nn = q->NumNegativePics; v.np = q->NumPositivePics;
    return v;
}
HBS_HD RowView view_of_zeros(const int* zeros32)        /* a row nobody has written: at least 32 zero ints */
{
    RowView v;
    v.s0 = v.u0 = v.s1 = v.u1 = zeros32; v.nd = v.nn = v.np = 0;
    return v;
}

/* What a slice's walk did with the derived tables, recorded by every parse (hbs_parse_fix.h decides from it whose answer
 * depends on NALs in front of its SPS): bits 0-5 own row written + 1, bits 6-11 row its own set was predicted from + 1,
 * bits 12-17 row num_pic_total_curr read + 1 (0 = none each), bit 18: it read the row of ITS OWN set's index without having coded a set (round 5: the batch walk gave it a
 * private, empty row there).  Which rows a slice touches does not depend on what the rows hold -- the indices are read from the bits in front of any table-dependent length --
 * so the record of a slice that read the wrong content is still right. */
HBS_HD uint32_t deps_pack(int own_written_idx, int ref_row, int read_row, int own_idx = -2)
{
    return (uint32_t)(own_written_idx + 1) | ((uint32_t)(ref_row + 1) << 6) | ((uint32_t)(read_row + 1) << 12) | ((read_row >= 0 && read_row == own_idx && own_written_idx != read_row) ? (1u << 18) : 0u);
}
HBS_HD int deps_own(uint32_t d) { return (int)(d & 63u) - 1; }
HBS_HD int deps_ref(uint32_t d) { return (int)((d >> 6) & 63u) - 1; }
HBS_HD int deps_read(uint32_t d) { return (int)((d >> 12) & 63u) - 1; }

/* per-NAL result record of hbs_parse_headers (layout = hbs_parsed_nal in the public header) */
struct ParsedNal {
    int32_t rc;                    /* what read_hevc_nal_unit returns                         */
    int32_t nal_unit_type, nal_layer_id, nal_temporal_id_plus1;
    uint64_t struct_off;           /* offset of the parsed struct in the struct arena, ~0: none */
    int32_t slice_data_size;       /* h->slice_data->rbsp_size (slices)                       */
    uint32_t slice_data_off;       /* where that payload starts inside the NAL's RBSP         */
};

/* One line of the per-field trace (read_debug_*, hevc_stream.c:2343-3434): which read it was (a
 * site number of this file, see HBS_SITE), the cursor before it, and the value it returned. */
struct TraceRec { uint32_t site; uint32_t pos; int32_t value; };

/* A read site = line of this file * 8 + its ordinal on that line: stable across translation units,
 * and the key of the name table hbs_trace_names.h (regenerated by tests/golden/make_trace_names.py
 * whenever lines move). */
#define HBS_SITE(k) ((uint32_t)(__LINE__ * 8 + (k)))

/* what a walk over the syntax does at each element */
enum : int { kModeRead = 0, kModeTrace = 1, kModeWrite = 2 };

/* bs.h over a bit position: read half, read half + trace, or write half.  The mode is a template
 * parameter so that the parse kernel carries neither the trace checks nor the writer's code. */
template <int kMode>
struct BitIOT {
    static constexpr bool kTrace = kMode == kModeTrace, kWrite = kMode == kModeWrite;
    const uint8_t* win;            /* first win_bytes of the RBSP (LDS on the device)         */
    const uint8_t* full;           /* the whole RBSP (global memory)                          */
    uint32_t win_bytes;
    uint32_t size;                 /* RBSP bytes                                              */
    uint32_t pos;                  /* bits consumed                                           */
    TraceRec* tr;                  /* trace sink or nullptr                                   */
    uint32_t tr_cap, tr_n;         /* records it holds / records produced (may exceed tr_cap) */
    uint8_t* wbuf;                 /* writing: `size` bytes, zeroed by the caller              */

    HBS_M uint32_t byte_at(uint32_t i) const { return i < win_bytes ? win[i] : full[i]; }
    HBS_M bool eof() const { return (pos >> 3) >= size; }                    /* bs.h:117 */
    HBS_M bool overrun() const { return (pos >> 3) > size; }                 /* bs.h:119 */
    HBS_M bool aligned() const { return (pos & 7u) == 0; }                   /* bs.h:112 */
    HBS_M void log(uint32_t site, uint32_t at, uint32_t value)
    {
        if (kTrace) {
            trace_put(reinterpret_cast<uint32_t*>(tr), tr_cap, tr_n, site, at, value);      /* a CALL (hbs_bitfast.h): inlined at every read site, the three stores and their address arithmetic made the trace kernels spill 3 000 registers */
            ++tr_n;
        }
    }
    /* a cursor position the reference prints without reading anything (record with an empty name) */
    HBS_M void mark(uint32_t site) { if (!kWrite) log(site, pos, 0u); }
    HBS_M uint32_t bit()                                                      /* bs.h:126-140 */
    {
        const uint32_t i = pos >> 3;
        uint32_t r = 0;
        if (i < size) r = (byte_at(i) >> (7u - (pos & 7u))) & 1u;
        ++pos;
        return r;
    }
    /* bs.h:160-169; the shift count wraps at 32 as the reference's `int << n` does on x86 for the
     * 34- and 43-bit reserved fields it reads through the same function */
    HBS_M uint32_t bits(int n)
    {
        uint32_t r = 0;
        if (!fast_bits(*this, n, r)) for (int i = 0; i < n; ++i) r |= bit() << ((uint32_t)(n - i - 1) & 31u);
        return r;
    }
    /* ---- bs.h write half (:224-335), for the syntax writers: the same walk with wr set puts
     * the value it is handed (the struct field the read would fill) instead of reading ------- */
    HBS_M void put(uint32_t v)                                                /* bs.h:224-238 */
    {
        const uint32_t i = pos >> 3;
        if (i < size) {
            const uint32_t sh = 7u - (pos & 7u);
            wbuf[i] = (uint8_t)((wbuf[i] & ~(1u << sh)) | ((v & 1u) << sh));
        }
        ++pos;
    }
    HBS_M void put_bits(int n, uint32_t v)                                    /* bs.h:240-247; shift counts wrap at 32 */
    {
        if (!fast_put_bits(*this, n, v)) for (int i = 0; i < n; ++i) put((v >> ((uint32_t)(n - i - 1) & 31u)) & 1u);
    }
    HBS_M void put_ue(uint32_t v)                                             /* bs.h:264-319 */
    {
        if (v == 0) { put(1u); return; }
        const uint32_t v1 = v + 1u;
        const int len = v1 ? 32 - __builtin_clz(v1) : 1;
        put_bits(2 * len - 1, v1);
    }

    /* every element: (site of the trace, value to put when writing) */
    HBS_M uint32_t u1(uint32_t site, uint32_t wv)
    {
        if (kWrite) { put(wv); return wv; }
        const uint32_t at = pos; const uint32_t r = bit(); log(site, at, r); return r;
    }
    HBS_M uint32_t u(int n, uint32_t site, uint32_t wv)
    {
        if (kWrite) { put_bits(n, wv); return wv; }
        const uint32_t at = pos; const uint32_t r = bits(n); log(site, at, r); return r;
    }
    HBS_M uint32_t u8(uint32_t site, uint32_t wv) { return u(8, site, wv); }  /* bs.h:182-193, :251-262: same either path */
    /* reserved bits the reference reads and only prints; written as the fixed pattern f(n, value) */
    HBS_M void skip(int n, uint32_t site, uint32_t fixed)
    {
        if (kWrite) { put_bits(n, fixed); return; }
        if (kTrace) { const uint32_t at = pos; const uint32_t r = bits(n); log(site, at, r); return; }
        if (n > 0) pos += (uint32_t)n;
    }
    HBS_M uint32_t ue_raw()                                                   /* bs.h:195-207 */
    {
        int i = 0;
        if (!fast_zeros(*this, i)) while ((bit() == 0) && (i < 32) && (!eof())) ++i;
        uint32_t r = bits(i);
        r += (1u << (i & 31)) - 1u;
        return r;
    }
    HBS_M uint32_t ue(uint32_t site, uint32_t wv)
    {
        if (kWrite) { put_ue(wv); return wv; }
        const uint32_t at = pos; const uint32_t r = ue_raw(); log(site, at, r); return r;
    }
    HBS_M int32_t se(uint32_t site, int32_t wv)                               /* bs.h:209-221, :321-331 */
    {
        if (kWrite) { put_ue(wv <= 0 ? (uint32_t)(-wv) * 2u : (uint32_t)wv * 2u - 1u); return wv; }
        const uint32_t at = pos;
        const int32_t r = (int32_t)ue_raw();
        const int32_t v = (r & 1) ? (r + 1) / 2 : -(r / 2);
        log(site, at, (uint32_t)v);
        return v;
    }
    /* hevc_stream.c:630-649: a one, then zeros to the byte boundary (a whole byte if aligned); the
     * reference reads (and prints) them one by one */
    HBS_M void trailing(uint32_t site_first, uint32_t site_rest)
    {
        if (kWrite) { put(1u); while (!aligned()) put(0u); return; }
        if (kTrace) { (void)u1(site_first, 0u); while (!aligned()) (void)u1(site_rest, 0u); return; }
        ++pos;
        while (!aligned()) ++pos;
    }
};

HBS_HD bool in32(int v) { return v >= 0 && v < 32; }

/* (int)ceil(log2(n)) as the reference evaluates it on x86; exact in integers */
HBS_HD int ceil_log2_int(int n)
{
    if (n <= 1) return 0;
    return 32 - __builtin_clz((unsigned)(n - 1));
}

/* hevc_stream.c:115-123 in integers: ceil(w / 2^k) is exact in the reference's
 * float arithmetic for picture dimensions below 2^24 */
HBS_HD int slice_address_bits(const hevc_sps_t* sps)
{
    const int lg = sps->log2_min_luma_coding_block_size_minus3 + 3 + sps->log2_diff_max_min_luma_coding_block_size;
    if (lg < 0 || lg > 30) return 0;
    const int w = sps->pic_width_in_luma_samples, h = sps->pic_height_in_luma_samples;
    if (w < 0 || h < 0) return 0;
    const long long ctb = 1ll << lg;
    const long long n = (((long long)w + ctb - 1) >> lg) * (((long long)h + ctb - 1) >> lg);
    return n > 0x7FFFFFFFll ? 0 : ceil_log2_int((int)n);
}

template <int kMode>
struct ParserT {
    BitIOT<kMode> b;
    /* context of a slice: the parameter sets and tables in force, read-only */
    const hevc_sps_t* sps;
    const hevc_pps_t* pps;
    const RpsTables* sps_rps;
    /* tables being written: an SPS fills `out_rps`; a slice fills `own` (row own_idx) */
    RpsTables* out_rps;
    RpsRow* own;
    int own_idx; int diverged;   /* diverged: this slice reads or clobbers an RPS row whose content depends on NALs in front of its SPS (see hbs_parse.hip, k4_seq) */
    /* the exact re-walk of one slice (hbs_parse_fix.h): up to two rows whose true content -- what the last NAL that wrote them
     * in stream order left -- is handed in; they go before everything else.  (The re-walk also points sps_rps at zeros: the reference's tables always exist.) */
    int ov_idx[2]; RowView ov[2];
    int stop_after_rps;          /* leave slice_segment_header once the slice's own short-term set is derived (the re-walk wants only that row) */
    int rec_own, rec_ref, rec_read;      /* own row written / row it was predicted from / row num_pic_total_curr read, -1: none */
    HBS_M void init_rows()
    {
        sps_rps = nullptr; out_rps = nullptr; own = nullptr; own_idx = -1; diverged = 0;
        ov_idx[0] = ov_idx[1] = -1; stop_after_rps = 0; rec_own = rec_ref = rec_read = -1;
    }
    HBS_M int ovi(int r) const { return r == ov_idx[0] ? 0 : (r == ov_idx[1] ? 1 : -1); }

    /* ---- access to the RPS rows in force ------------------------------------------- */
    HBS_M int numDelta(int r) const { if (!in32(r)) return 0; const int o = ovi(r); if (o >= 0) return ov[o].nd; return (own && r == own_idx) ? own->NumDeltaPocs : (out_rps ? out_rps->NumDeltaPocs[r] : (sps_rps ? sps_rps->NumDeltaPocs[r] : 0)); }
    HBS_M int numNeg(int r) const { if (!in32(r)) return 0; const int o = ovi(r); if (o >= 0) return ov[o].nn; return (own && r == own_idx) ? own->NumNegativePics : (out_rps ? out_rps->NumNegativePics[r] : (sps_rps ? sps_rps->NumNegativePics[r] : 0)); }
    HBS_M int numPos(int r) const { if (!in32(r)) return 0; const int o = ovi(r); if (o >= 0) return ov[o].np; return (own && r == own_idx) ? own->NumPositivePics : (out_rps ? out_rps->NumPositivePics[r] : (sps_rps ? sps_rps->NumPositivePics[r] : 0)); }
    HBS_M const int* rowS0(int r) const { const int o = ovi(r); if (o >= 0) return ov[o].s0; return (own && r == own_idx) ? own->DeltaPocS0 : (out_rps ? out_rps->DeltaPocS0[r] : sps_rps->DeltaPocS0[r]); }
    HBS_M const int* rowU0(int r) const { const int o = ovi(r); if (o >= 0) return ov[o].u0; return (own && r == own_idx) ? own->UsedByCurrPicS0 : (out_rps ? out_rps->UsedByCurrPicS0[r] : sps_rps->UsedByCurrPicS0[r]); }
    HBS_M const int* rowS1(int r) const { const int o = ovi(r); if (o >= 0) return ov[o].s1; return (own && r == own_idx) ? own->DeltaPocS1 : (out_rps ? out_rps->DeltaPocS1[r] : sps_rps->DeltaPocS1[r]); }
    HBS_M const int* rowU1(int r) const { const int o = ovi(r); if (o >= 0) return ov[o].u1; return (own && r == own_idx) ? own->UsedByCurrPicS1 : (out_rps ? out_rps->UsedByCurrPicS1[r] : sps_rps->UsedByCurrPicS1[r]); }
    HBS_M bool have_rows() const { return out_rps != nullptr || sps_rps != nullptr; }
    /* destination row `r` of the set being parsed */
    HBS_M int* wS0(int r) { return (own && r == own_idx) ? own->DeltaPocS0 : out_rps->DeltaPocS0[r]; }
    HBS_M int* wU0(int r) { return (own && r == own_idx) ? own->UsedByCurrPicS0 : out_rps->UsedByCurrPicS0[r]; }
    HBS_M int* wS1(int r) { return (own && r == own_idx) ? own->DeltaPocS1 : out_rps->DeltaPocS1[r]; }
    HBS_M int* wU1(int r) { return (own && r == own_idx) ? own->UsedByCurrPicS1 : out_rps->UsedByCurrPicS1[r]; }
    HBS_M void setCounts(int r, int neg, int pos)
    {
        if (own && r == own_idx) { own->NumNegativePics = neg; own->NumPositivePics = pos; own->NumDeltaPocs = neg + pos; }
        else { out_rps->NumNegativePics[r] = neg; out_rps->NumPositivePics[r] = pos; out_rps->NumDeltaPocs[r] = neg + pos; }
    }
    HBS_M bool can_write(int r) const { return in32(r) && ((own && r == own_idx) || out_rps != nullptr); }

    /* ---- 7.3.3 (hevc_stream.c:652-755) ---------------------------------------------- */
    HBS_M void profile_tier_level(hevc_profile_tier_level_t* ptl, int maxNumSubLayersMinus1)
    {
        ptl->general_profile_space = b.u(2, HBS_SITE(0), ptl->general_profile_space);
        ptl->general_tier_flag = b.u1(HBS_SITE(0), ptl->general_tier_flag);
        const int idc = b.u(5, HBS_SITE(0), ptl->general_profile_idc);
        ptl->general_profile_idc = idc;
        uint32_t compat = 0;
        for (int i = 0; i < 32; ++i) { const uint32_t v = b.u1(HBS_SITE(0), ptl->general_profile_compatibility_flag[i]); ptl->general_profile_compatibility_flag[i] = v; compat |= v << i; }
        ptl->general_progressive_source_flag = b.u1(HBS_SITE(0), ptl->general_progressive_source_flag);
        ptl->general_interlaced_source_flag = b.u1(HBS_SITE(0), ptl->general_interlaced_source_flag);
        ptl->general_non_packed_constraint_flag = b.u1(HBS_SITE(0), ptl->general_non_packed_constraint_flag);
        ptl->general_frame_only_constraint_flag = b.u1(HBS_SITE(0), ptl->general_frame_only_constraint_flag);
        if ((idc >= 4 && idc <= 7) || (compat & 0xF0u)) {
            ptl->general_max_12bit_constraint_flag = b.u1(HBS_SITE(0), ptl->general_max_12bit_constraint_flag);
            ptl->general_max_10bit_constraint_flag = b.u1(HBS_SITE(0), ptl->general_max_10bit_constraint_flag);
            ptl->general_max_8bit_constraint_flag = b.u1(HBS_SITE(0), ptl->general_max_8bit_constraint_flag);
            ptl->general_max_422chroma_constraint_flag = b.u1(HBS_SITE(0), ptl->general_max_422chroma_constraint_flag);
            ptl->general_max_420chroma_constraint_flag = b.u1(HBS_SITE(0), ptl->general_max_420chroma_constraint_flag);
            ptl->general_max_monochrome_constraint_flag = b.u1(HBS_SITE(0), ptl->general_max_monochrome_constraint_flag);
            ptl->general_intra_constraint_flag = b.u1(HBS_SITE(0), ptl->general_intra_constraint_flag);
            ptl->general_one_picture_only_constraint_flag = b.u1(HBS_SITE(0), ptl->general_one_picture_only_constraint_flag);
            ptl->general_lower_bit_rate_constraint_flag = b.u1(HBS_SITE(0), ptl->general_lower_bit_rate_constraint_flag);
            b.skip(34, HBS_SITE(0), 0u);
        } else {
            b.skip(43, HBS_SITE(0), 0u);
        }
        if ((idc >= 1 && idc <= 5) || (compat & 0x3Eu)) ptl->general_inbld_flag = b.u1(HBS_SITE(0), ptl->general_inbld_flag);
        else b.skip(1, HBS_SITE(0), 0u);
        ptl->general_level_idc = b.u8(HBS_SITE(0), ptl->general_level_idc);
        uint32_t prof_present = 0, level_present = 0;
        for (int i = 0; i < maxNumSubLayersMinus1; ++i) {
            const uint32_t p = b.u1(HBS_SITE(0), ptl->sub_layer_profile_present_flag[i]), l = b.u1(HBS_SITE(1), ptl->sub_layer_level_present_flag[i]);
            ptl->sub_layer_profile_present_flag[i] = p;
            ptl->sub_layer_level_present_flag[i] = l;
            prof_present |= p << i; level_present |= l << i;
        }
        if (maxNumSubLayersMinus1 > 0)
            for (int i = maxNumSubLayersMinus1; i < 8; ++i) b.skip(2, HBS_SITE(0), 0u);
        for (int i = 0; i < maxNumSubLayersMinus1; ++i) {
            if ((prof_present >> i) & 1u) {
                ptl->sub_layer_profile_space[i] = b.u(2, HBS_SITE(0), ptl->sub_layer_profile_space[i]);
                ptl->sub_layer_tier_flag[i] = b.u1(HBS_SITE(0), ptl->sub_layer_tier_flag[i]);
                const int sidc = b.u(5, HBS_SITE(0), ptl->sub_layer_profile_idc[i]);
                ptl->sub_layer_profile_idc[i] = sidc;
                uint32_t sc = 0;
                for (int j = 0; j < 32; ++j) { const uint32_t v = b.u(1, HBS_SITE(0), ptl->sub_layer_profile_compatibility_flag[i][j]); ptl->sub_layer_profile_compatibility_flag[i][j] = v; sc |= v << j; }
                ptl->sub_layer_progressive_source_flag[i] = b.u1(HBS_SITE(0), ptl->sub_layer_progressive_source_flag[i]);
                ptl->sub_layer_interlaced_source_flag[i] = b.u1(HBS_SITE(0), ptl->sub_layer_interlaced_source_flag[i]);
                ptl->sub_layer_non_packed_constraint_flag[i] = b.u1(HBS_SITE(0), ptl->sub_layer_non_packed_constraint_flag[i]);
                ptl->sub_layer_frame_only_constraint_flag[i] = b.u1(HBS_SITE(0), ptl->sub_layer_frame_only_constraint_flag[i]);
                if ((sidc >= 4 && sidc <= 7) || (sc & 0xF0u)) {
                    ptl->sub_layer_max_12bit_constraint_flag[i] = b.u1(HBS_SITE(0), ptl->sub_layer_max_12bit_constraint_flag[i]);
                    ptl->sub_layer_max_10bit_constraint_flag[i] = b.u1(HBS_SITE(0), ptl->sub_layer_max_10bit_constraint_flag[i]);
                    ptl->sub_layer_max_8bit_constraint_flag[i] = b.u1(HBS_SITE(0), ptl->sub_layer_max_8bit_constraint_flag[i]);
                    ptl->sub_layer_max_422chroma_constraint_flag[i] = b.u1(HBS_SITE(0), ptl->sub_layer_max_422chroma_constraint_flag[i]);
                    ptl->sub_layer_max_420chroma_constraint_flag[i] = b.u1(HBS_SITE(0), ptl->sub_layer_max_420chroma_constraint_flag[i]);
                    ptl->sub_layer_max_monochrome_constraint_flag[i] = b.u1(HBS_SITE(0), ptl->sub_layer_max_monochrome_constraint_flag[i]);
                    ptl->sub_layer_intra_constraint_flag[i] = b.u1(HBS_SITE(0), ptl->sub_layer_intra_constraint_flag[i]);
                    ptl->sub_layer_one_picture_only_constraint_flag[i] = b.u1(HBS_SITE(0), ptl->sub_layer_one_picture_only_constraint_flag[i]);
                    ptl->sub_layer_lower_bit_rate_constraint_flag[i] = b.u1(HBS_SITE(0), ptl->sub_layer_lower_bit_rate_constraint_flag[i]);
                    b.skip(34, HBS_SITE(0), 0u);
                } else {
                    b.skip(43, HBS_SITE(0), 0u);
                }
                ptl->sub_layer_inbld_flag[i] = b.u1(HBS_SITE(0), ptl->sub_layer_inbld_flag[i]);          /* :739-745: the test is always true */
            }
            /* the reference's plain reader takes 8 bits here (hevc_stream.c:751), its read_debug_* and
             * write_* variants one (:2939, :1845: the committed file differs from its template) */
            if ((level_present >> i) & 1u) ptl->sub_layer_level_idc[i] = (kMode != kModeRead) ? b.u1(HBS_SITE(0), ptl->sub_layer_level_idc[i]) : b.u8(HBS_SITE(1), 0);
        }
    }

    /* ---- 7.3.4 (hevc_stream.c:758-779) ----------------------------------------------- */
    HBS_M void scaling_list_data(hevc_scaling_list_data_t* sld)
    {
        for (int sizeId = 0; sizeId < 4; ++sizeId)
            for (int matrixId = 0; matrixId < 6; matrixId += (sizeId == 3) ? 3 : 1) {
                const uint32_t mode = b.u1(HBS_SITE(0), sld->scaling_list_pred_mode_flag[sizeId][matrixId]);
                sld->scaling_list_pred_mode_flag[sizeId][matrixId] = mode;
                if (!mode) {
                    sld->scaling_list_pred_matrix_id_delta[sizeId][matrixId] = b.ue(HBS_SITE(0), sld->scaling_list_pred_matrix_id_delta[sizeId][matrixId]);
                } else {
                    const int coefNum = (sizeId == 0) ? 16 : 64;
                    if (sizeId > 1) sld->scaling_list_dc_coef_minus8[sizeId - 2][matrixId] = b.se(HBS_SITE(0), sld->scaling_list_dc_coef_minus8[sizeId - 2][matrixId]);
                    int last = 0;
                    for (int i = 0; i < coefNum; ++i) last = b.se(HBS_SITE(0), sld->scaling_list_delta_coef[sizeId][matrixId]);
                    sld->scaling_list_delta_coef[sizeId][matrixId] = last;      /* :774: one element keeps the last value */
                }
            }
    }

    /* ---- E.2.3 (hevc_stream.c:1207-1218) ------------------------------------------------ */
    HBS_M void sub_layer_hrd(hevc_sub_layer_hrd_t* s, int CpbCnt, int sub_pic)
    {
        for (int i = 0; i <= CpbCnt; ++i) {
            const int k = i < MAX_CPB_CNT ? i : MAX_CPB_CNT - 1;
            s->bit_rate_value_minus1[k] = b.ue(HBS_SITE(0), s->bit_rate_value_minus1[k]);
            s->cpb_size_value_minus1[k] = b.ue(HBS_SITE(0), s->cpb_size_value_minus1[k]);
            if (sub_pic) {
                s->cpb_size_du_value_minus1[k] = b.ue(HBS_SITE(0), s->cpb_size_du_value_minus1[k]);
                s->bit_rate_du_value_minus1[k] = b.ue(HBS_SITE(0), s->bit_rate_du_value_minus1[k]);
            }
            s->cbr_flag[k] = b.u1(HBS_SITE(0), s->cbr_flag[k]);
        }
    }

    /* ---- E.2.2 (hevc_stream.c:1160-1204) -------------------------------------------------- */
    HBS_M void hrd_parameters(hevc_hrd_t* hrd, int commonInfPresentFlag, int maxNumSubLayersMinus1)
    {
        if (commonInfPresentFlag) {
            const uint32_t nal_p = b.u1(HBS_SITE(0), hrd->nal_hrd_parameters_present_flag), vcl_p = b.u1(HBS_SITE(1), hrd->vcl_hrd_parameters_present_flag);
            hrd->nal_hrd_parameters_present_flag = nal_p;
            hrd->vcl_hrd_parameters_present_flag = vcl_p;
            if (nal_p || vcl_p) {
                const uint32_t sub_pic = b.u1(HBS_SITE(0), hrd->sub_pic_hrd_params_present_flag);
                hrd->sub_pic_hrd_params_present_flag = sub_pic;
                if (sub_pic) {
                    hrd->tick_divisor_minus2 = b.u8(HBS_SITE(0), hrd->tick_divisor_minus2);
                    hrd->du_cpb_removal_delay_increment_length_minus1 = b.u(5, HBS_SITE(0), hrd->du_cpb_removal_delay_increment_length_minus1);
                    hrd->sub_pic_cpb_params_in_pic_timing_sei_flag = b.u1(HBS_SITE(0), hrd->sub_pic_cpb_params_in_pic_timing_sei_flag);
                    hrd->dpb_output_delay_du_length_minus1 = b.u(5, HBS_SITE(0), hrd->dpb_output_delay_du_length_minus1);
                }
                hrd->bit_rate_scale = b.u(4, HBS_SITE(0), hrd->bit_rate_scale);
                hrd->cpb_size_scale = b.u(4, HBS_SITE(0), hrd->cpb_size_scale);
                if (sub_pic) hrd->cpb_size_du_scale = b.u(4, HBS_SITE(0), hrd->cpb_size_du_scale);
                hrd->initial_cpb_removal_delay_length_minus1 = b.u(5, HBS_SITE(0), hrd->initial_cpb_removal_delay_length_minus1);
                hrd->au_cpb_removal_delay_length_minus1 = b.u(5, HBS_SITE(0), hrd->au_cpb_removal_delay_length_minus1);
                hrd->dpb_output_delay_length_minus1 = b.u(5, HBS_SITE(0), hrd->dpb_output_delay_length_minus1);
            }
        }
        /* the tests below read the struct, as the reference does: without common
         * info they see what an earlier parse (or the memset) left there */
        const int nal_p = hrd->nal_hrd_parameters_present_flag, vcl_p = hrd->vcl_hrd_parameters_present_flag;
        const int sub_pic = hrd->sub_pic_hrd_params_present_flag;
        for (int i = 0; i <= maxNumSubLayersMinus1; ++i) {
            const uint32_t general = b.u1(HBS_SITE(0), hrd->fixed_pic_rate_general_flag[i]);
            hrd->fixed_pic_rate_general_flag[i] = general;
            if (!general) hrd->fixed_pic_rate_within_cvs_flag[i] = b.u1(HBS_SITE(0), hrd->fixed_pic_rate_within_cvs_flag[i]);
            if (hrd->fixed_pic_rate_within_cvs_flag[i]) hrd->elemental_duration_in_tc_minus1[i] = b.ue(HBS_SITE(0), hrd->elemental_duration_in_tc_minus1[i]);
            else hrd->low_delay_hrd_flag[i] = b.u1(HBS_SITE(0), hrd->low_delay_hrd_flag[i]);
            if (hrd->low_delay_hrd_flag[i]) hrd->cpb_cnt_minus1[i] = b.ue(HBS_SITE(0), hrd->cpb_cnt_minus1[i]);                        /* :1194 */
            const int cpb_cnt = hrd->cpb_cnt_minus1[i] + 1;
            if (nal_p) sub_layer_hrd(&hrd->sub_layer_hrd_nal[i], cpb_cnt, sub_pic);
            if (vcl_p) sub_layer_hrd(&hrd->sub_layer_hrd_vcl[i], cpb_cnt, sub_pic);
        }
    }

    /* ---- 7.3.7 + derivation (hevc_stream.c:1032-1085, :61-113) ------------------------------ */
    template <class RPS> HBS_M void st_ref_pic_set(RPS* rps, int stRpsIdx, int num_sets)
    {
        if (stRpsIdx == num_sets && can_write(stRpsIdx)) rec_own = stRpsIdx;
        int inter = 0;
        if (stRpsIdx != 0) { inter = b.u1(HBS_SITE(0), rps->inter_ref_pic_set_prediction_flag); rps->inter_ref_pic_set_prediction_flag = inter; }
        if (inter) {
            int delta_idx_minus1 = 0;
            if (stRpsIdx == num_sets) { delta_idx_minus1 = (int)b.ue(HBS_SITE(0), rps->delta_idx_minus1); rps->delta_idx_minus1 = delta_idx_minus1; }
            const int sign = b.u1(HBS_SITE(0), rps->delta_rps_sign);
            rps->delta_rps_sign = sign;
            const int absd = (int)b.ue(HBS_SITE(0), rps->abs_delta_rps_minus1);
            rps->abs_delta_rps_minus1 = absd;
            const int RefRpsIdx = stRpsIdx - (delta_idx_minus1 + 1);
            if (stRpsIdx == num_sets && in32(RefRpsIdx)) rec_ref = RefRpsIdx;      /* a slice's own set, predicted from a row it did not write */
            const bool ref_ok = in32(RefRpsIdx) && have_rows();
            const int lim = ref_ok ? numDelta(RefRpsIdx) : 0;
            uint64_t used = 0, use_delta = 0;                     /* bit k: flag k (k < 32) */
            for (int j = 0; j <= lim; ++j) {
                const int k = in32(j) ? j : 31;
                const uint64_t u = b.u1(HBS_SITE(0), rps->used_by_curr_pic_flag[k]);
                rps->used_by_curr_pic_flag[k] = (int)u;
                used = (used & ~(1ull << k)) | (u << k);
                if (!u) {
                    const uint64_t d = b.u1(HBS_SITE(0), rps->use_delta_flag[k]);
                    rps->use_delta_flag[k] = (int)d;
                    use_delta = (use_delta & ~(1ull << k)) | (d << k);
                }
            }
            if (!can_write(stRpsIdx)) return;
#define HBS_USED(k) (in32(k) ? (int)((used >> (k)) & 1ull) : 0)
#define HBS_USE_DELTA(k) (in32(k) ? (int)((use_delta >> (k)) & 1ull) : 0)
            const int deltaRps = (1 - 2 * sign) * (absd + 1);
            const int refNeg = ref_ok ? numNeg(RefRpsIdx) : 0, refPos = ref_ok ? numPos(RefRpsIdx) : 0, refNum = lim;
            int* dS0 = wS0(stRpsIdx); int* dU0 = wU0(stRpsIdx); int* dS1 = wS1(stRpsIdx); int* dU1 = wU1(stRpsIdx);
            int i = 0;
            for (int j = refPos - 1; j >= 0; --j) {
                if (!in32(j)) continue;
                const int dPoc = rowS1(RefRpsIdx)[j] + deltaRps;
                if (dPoc < 0 && HBS_USE_DELTA(refNeg + j)) { if (in32(i)) { dS0[i] = dPoc; dU0[i] = HBS_USED(refNeg + j); } ++i; }
            }
            if (deltaRps < 0 && HBS_USE_DELTA(refNum)) { if (in32(i)) { dS0[i] = deltaRps; dU0[i] = HBS_USED(refNum); } ++i; }
            for (int j = 0; j < refNeg; ++j) {
                if (!in32(j)) continue;
                const int dPoc = rowS0(RefRpsIdx)[j] + deltaRps;
                if (dPoc < 0 && HBS_USE_DELTA(j)) { if (in32(i)) { dS0[i] = dPoc; dU0[i] = HBS_USED(j); } ++i; }
            }
            const int neg = i;
            i = 0;
            for (int j = refNeg - 1; j >= 0; --j) {
                if (!in32(j)) continue;
                const int dPoc = rowS0(RefRpsIdx)[j] + deltaRps;
                if (dPoc > 0 && HBS_USE_DELTA(j)) { if (in32(i)) { dS1[i] = dPoc; dU1[i] = HBS_USED(j); } ++i; }
            }
            if (deltaRps > 0 && HBS_USE_DELTA(refNum)) { if (in32(i)) { dS1[i] = deltaRps; dU1[i] = HBS_USED(refNum); } ++i; }
            for (int j = 0; j < refPos; ++j) {
                if (!in32(j)) continue;
                const int dPoc = rowS1(RefRpsIdx)[j] + deltaRps;
                if (dPoc > 0 && HBS_USE_DELTA(refNeg + j)) { if (in32(i)) { dS1[i] = dPoc; dU1[i] = HBS_USED(refNeg + j); } ++i; }
            }
            setCounts(stRpsIdx, neg, i);
#undef HBS_USED
#undef HBS_USE_DELTA
        } else {
            const int neg = (int)b.ue(HBS_SITE(0), rps->num_negative_pics), pos = (int)b.ue(HBS_SITE(1), rps->num_positive_pics);
            rps->num_negative_pics = neg;
            rps->num_positive_pics = pos;
            const bool wr = can_write(stRpsIdx);
            int acc = 0;
            for (int i = 0; i < neg; ++i) {
                const int k = in32(i) ? i : 31;
                const int d = (int)b.ue(HBS_SITE(0), rps->delta_poc_s0_minus1[k]), u = (int)b.u1(HBS_SITE(1), rps->used_by_curr_pic_s0_flag[k]);
                rps->delta_poc_s0_minus1[k] = d;
                rps->used_by_curr_pic_s0_flag[k] = u;
                if (wr) {
                    int* dS0 = wS0(stRpsIdx);
                    wU0(stRpsIdx)[k] = u;
                    acc = (i == 0) ? -(d + 1) : dS0[k - 1 >= 0 ? k - 1 : 0] - (d + 1);
                    dS0[k] = acc;
                }
            }
            for (int i = 0; i < pos; ++i) {
                const int k = in32(i) ? i : 31;
                const int d = (int)b.ue(HBS_SITE(0), rps->delta_poc_s1_minus1[k]), u = (int)b.u1(HBS_SITE(1), rps->used_by_curr_pic_s1_flag[k]);
                rps->delta_poc_s1_minus1[k] = d;
                rps->used_by_curr_pic_s1_flag[k] = u;
                if (wr) {
                    int* dS1 = wS1(stRpsIdx);
                    wU1(stRpsIdx)[k] = u;
                    acc = (i == 0) ? (d + 1) : dS1[k - 1 >= 0 ? k - 1 : 0] + (d + 1);
                    dS1[k] = acc;
                }
            }
            if (wr) setCounts(stRpsIdx, neg, pos);
        }
    }

    /* ---- E.2.1 (hevc_stream.c:1088-1157) -------------------------------------------------------- */
    HBS_M void vui_parameters(hevc_vui_t* vui, int sps_max_sub_layers_minus1)
    {
        uint32_t f = b.u1(HBS_SITE(0), vui->aspect_ratio_info_present_flag);
        vui->aspect_ratio_info_present_flag = f;
        if (f) {
            const uint32_t idc = b.u8(HBS_SITE(0), vui->aspect_ratio_idc);
            vui->aspect_ratio_idc = idc;
            if (idc == 255) { vui->sar_width = b.u(16, HBS_SITE(0), vui->sar_width); vui->sar_height = b.u(16, HBS_SITE(1), vui->sar_height); }
        }
        f = b.u1(HBS_SITE(0), vui->overscan_info_present_flag); vui->overscan_info_present_flag = f;
        if (f) vui->overscan_appropriate_flag = b.u1(HBS_SITE(0), vui->overscan_appropriate_flag);
        f = b.u1(HBS_SITE(0), vui->video_signal_type_present_flag); vui->video_signal_type_present_flag = f;
        if (f) {
            vui->video_format = b.u(3, HBS_SITE(0), vui->video_format);
            vui->video_full_range_flag = b.u1(HBS_SITE(0), vui->video_full_range_flag);
            const uint32_t cd = b.u1(HBS_SITE(0), vui->colour_description_present_flag);
            vui->colour_description_present_flag = cd;
            if (cd) { vui->colour_primaries = b.u8(HBS_SITE(0), vui->colour_primaries); vui->transfer_characteristics = b.u8(HBS_SITE(1), vui->transfer_characteristics); vui->matrix_coefficients = b.u8(HBS_SITE(2), vui->matrix_coefficients); }
        }
        f = b.u1(HBS_SITE(0), vui->chroma_loc_info_present_flag); vui->chroma_loc_info_present_flag = f;
        if (f) { vui->chroma_sample_loc_type_top_field = b.ue(HBS_SITE(0), vui->chroma_sample_loc_type_top_field); vui->chroma_sample_loc_type_bottom_field = b.ue(HBS_SITE(1), vui->chroma_sample_loc_type_bottom_field); }
        vui->neutral_chroma_indication_flag = b.u1(HBS_SITE(0), vui->neutral_chroma_indication_flag);
        vui->field_seq_flag = b.u1(HBS_SITE(0), vui->field_seq_flag);
        vui->frame_field_info_present_flag = b.u1(HBS_SITE(0), vui->frame_field_info_present_flag);
        f = b.u1(HBS_SITE(0), vui->default_display_window_flag); vui->default_display_window_flag = f;
        if (f) {
            vui->def_disp_win_left_offset = b.ue(HBS_SITE(0), vui->def_disp_win_left_offset); vui->def_disp_win_right_offset = b.ue(HBS_SITE(1), vui->def_disp_win_right_offset);
            vui->def_disp_win_top_offset = b.ue(HBS_SITE(0), vui->def_disp_win_top_offset); vui->def_disp_win_bottom_offset = b.ue(HBS_SITE(1), vui->def_disp_win_bottom_offset);
        }
        f = b.u1(HBS_SITE(0), vui->vui_timing_info_present_flag); vui->vui_timing_info_present_flag = f;
        if (f) {
            vui->vui_num_units_in_tick = b.u(32, HBS_SITE(0), vui->vui_num_units_in_tick);
            vui->vui_time_scale = b.u(32, HBS_SITE(0), vui->vui_time_scale);
            const uint32_t poc = b.u1(HBS_SITE(0), vui->vui_poc_proportional_to_timing_flag);
            vui->vui_poc_proportional_to_timing_flag = poc;
            if (poc) vui->vui_num_ticks_poc_diff_one_minus1 = b.ue(HBS_SITE(0), vui->vui_num_ticks_poc_diff_one_minus1);
            const uint32_t hp = b.u1(HBS_SITE(0), vui->vui_hrd_parameters_present_flag);
            vui->vui_hrd_parameters_present_flag = hp;
            if (hp) hrd_parameters(&vui->hrd, 1, sps_max_sub_layers_minus1);
        }
        f = b.u1(HBS_SITE(0), vui->bitstream_restriction_flag); vui->bitstream_restriction_flag = f;
        if (f) {
            vui->tiles_fixed_structure_flag = b.u1(HBS_SITE(0), vui->tiles_fixed_structure_flag);
            vui->motion_vectors_over_pic_boundaries_flag = b.u1(HBS_SITE(0), vui->motion_vectors_over_pic_boundaries_flag);
            vui->restricted_ref_pic_lists_flag = b.u1(HBS_SITE(0), vui->restricted_ref_pic_lists_flag);
            vui->min_spatial_segmentation_idc = b.ue(HBS_SITE(0), vui->min_spatial_segmentation_idc);
            vui->max_bytes_per_pic_denom = b.ue(HBS_SITE(0), vui->max_bytes_per_pic_denom);
            vui->max_bits_per_min_cu_denom = b.ue(HBS_SITE(0), vui->max_bits_per_min_cu_denom);
            vui->log2_max_mv_length_horizontal = b.ue(HBS_SITE(0), vui->log2_max_mv_length_horizontal);
            vui->log2_max_mv_length_vertical = b.ue(HBS_SITE(0), vui->log2_max_mv_length_vertical);
        }
    }

    /* ---- 7.3.2.1 (hevc_stream.c:243-300); *vps is already zero ---------------------------------- */
    HBS_M void video_parameter_set(hevc_vps_t* vps)
    {
        vps->vps_video_parameter_set_id = b.u(4, HBS_SITE(0), vps->vps_video_parameter_set_id);
        vps->vps_base_layer_internal_flag = b.u1(HBS_SITE(0), vps->vps_base_layer_internal_flag);
        vps->vps_base_layer_available_flag = b.u1(HBS_SITE(0), vps->vps_base_layer_available_flag);
        vps->vps_max_layers_minus1 = b.u(6, HBS_SITE(0), vps->vps_max_layers_minus1);
        const int msl = b.u(3, HBS_SITE(0), vps->vps_max_sub_layers_minus1);
        vps->vps_max_sub_layers_minus1 = msl;
        vps->vps_temporal_id_nesting_flag = b.u1(HBS_SITE(0), vps->vps_temporal_id_nesting_flag);
        b.skip(16, HBS_SITE(0), 0xFFFFu);
        profile_tier_level(&vps->ptl, msl);
        const uint32_t info = b.u1(HBS_SITE(0), vps->vps_sub_layer_ordering_info_present_flag);
        vps->vps_sub_layer_ordering_info_present_flag = info;
        for (int i = (info ? 0 : msl); i <= msl; ++i) {
            vps->vps_max_dec_pic_buffering_minus1[i] = b.ue(HBS_SITE(0), vps->vps_max_dec_pic_buffering_minus1[i]);
            vps->vps_max_num_reorder_pics[i] = b.ue(HBS_SITE(0), vps->vps_max_num_reorder_pics[i]);
            vps->vps_max_latency_increase_plus1[i] = b.ue(HBS_SITE(0), vps->vps_max_latency_increase_plus1[i]);
        }
        const int max_layer_id = b.u(6, HBS_SITE(0), vps->vps_max_layer_id);
        vps->vps_max_layer_id = max_layer_id;
        const int sets = (int)b.ue(HBS_SITE(0), vps->vps_num_layer_sets_minus1);
        vps->vps_num_layer_sets_minus1 = sets;
        for (int i = 1; i <= sets; ++i)
            for (int j = 0; j <= max_layer_id; ++j) {
                const int v = b.u1(HBS_SITE(0), (i < MAX_NUM_SUBLAYERS && j < MAX_NUM_SUBLAYERS) ? vps->layer_id_included_flag[i][j] : 0);
                if (i < MAX_NUM_SUBLAYERS && j < MAX_NUM_SUBLAYERS) vps->layer_id_included_flag[i][j] = v;
            }
        const uint32_t timing = b.u1(HBS_SITE(0), vps->vps_timing_info_present_flag);
        vps->vps_timing_info_present_flag = timing;
        if (timing) {
            vps->vps_num_units_in_tick = b.u(32, HBS_SITE(0), vps->vps_num_units_in_tick);
            vps->vps_time_scale = b.u(32, HBS_SITE(0), vps->vps_time_scale);
            const uint32_t poc = b.u1(HBS_SITE(0), vps->vps_poc_proportional_to_timing_flag);
            vps->vps_poc_proportional_to_timing_flag = poc;
            if (poc) vps->vps_num_ticks_poc_diff_one_minus1 = b.ue(HBS_SITE(0), vps->vps_num_ticks_poc_diff_one_minus1);
            const int nhrd = (int)b.ue(HBS_SITE(0), vps->vps_num_hrd_parameters);
            vps->vps_num_hrd_parameters = nhrd;
            for (int i = 0; i < nhrd; ++i) {
                const int k = i < MAX_NUM_HRD_PARAM ? i : MAX_NUM_HRD_PARAM - 1;
                vps->hrd_layer_set_idx[k] = b.ue(HBS_SITE(0), vps->hrd_layer_set_idx[k]);
                if (i > 0) vps->cprms_present_flag[k] = b.u1(HBS_SITE(0), vps->cprms_present_flag[k]);
                hrd_parameters(&vps->hrd[k], vps->cprms_present_flag[k], msl);
            }
        }
        vps->vps_extension_flag = b.u1(HBS_SITE(0), vps->vps_extension_flag);
        b.trailing(HBS_SITE(0), HBS_SITE(1));
    }

    /* ---- 7.3.2.2 (hevc_stream.c:303-415); *sps is already zero; no trailing bits ------------------ */
    HBS_M void seq_parameter_set(hevc_sps_t* sps_out)
    {
        hevc_sps_t* s = sps_out;
        s->sps_video_parameter_set_id = b.u(4, HBS_SITE(0), s->sps_video_parameter_set_id);
        const int msl = b.u(3, HBS_SITE(0), s->sps_max_sub_layers_minus1);
        s->sps_max_sub_layers_minus1 = msl;
        s->sps_temporal_id_nesting_flag = b.u1(HBS_SITE(0), s->sps_temporal_id_nesting_flag);
        profile_tier_level(&s->ptl, msl);
        s->sps_seq_parameter_set_id = b.ue(HBS_SITE(0), s->sps_seq_parameter_set_id);
        const int chroma = (int)b.ue(HBS_SITE(0), s->chroma_format_idc);
        s->chroma_format_idc = chroma;
        if (chroma == 3) s->separate_colour_plane_flag = b.u1(HBS_SITE(0), s->separate_colour_plane_flag);
        s->pic_width_in_luma_samples = b.ue(HBS_SITE(0), s->pic_width_in_luma_samples);
        s->pic_height_in_luma_samples = b.ue(HBS_SITE(0), s->pic_height_in_luma_samples);
        const uint32_t cw = b.u1(HBS_SITE(0), s->conformance_window_flag);
        s->conformance_window_flag = cw;
        if (cw) {
            s->conf_win_left_offset = b.ue(HBS_SITE(0), s->conf_win_left_offset); s->conf_win_right_offset = b.ue(HBS_SITE(1), s->conf_win_right_offset);
            s->conf_win_top_offset = b.ue(HBS_SITE(0), s->conf_win_top_offset); s->conf_win_bottom_offset = b.ue(HBS_SITE(1), s->conf_win_bottom_offset);
        }
        s->bit_depth_luma_minus8 = b.ue(HBS_SITE(0), s->bit_depth_luma_minus8);
        s->bit_depth_chroma_minus8 = b.ue(HBS_SITE(0), s->bit_depth_chroma_minus8);
        const int poc_minus4 = (int)b.ue(HBS_SITE(0), s->log2_max_pic_order_cnt_lsb_minus4);
        s->log2_max_pic_order_cnt_lsb_minus4 = poc_minus4;
        const uint32_t info = b.u1(HBS_SITE(0), s->sps_sub_layer_ordering_info_present_flag);
        s->sps_sub_layer_ordering_info_present_flag = info;
        for (int i = (info ? 0 : msl); i <= msl; ++i) {
            s->sps_max_dec_pic_buffering_minus1[i] = b.ue(HBS_SITE(0), s->sps_max_dec_pic_buffering_minus1[i]);
            s->sps_max_num_reorder_pics[i] = b.ue(HBS_SITE(0), s->sps_max_num_reorder_pics[i]);
            s->sps_max_latency_increase_plus1[i] = b.ue(HBS_SITE(0), s->sps_max_latency_increase_plus1[i]);
        }
        s->log2_min_luma_coding_block_size_minus3 = b.ue(HBS_SITE(0), s->log2_min_luma_coding_block_size_minus3);
        s->log2_diff_max_min_luma_coding_block_size = b.ue(HBS_SITE(0), s->log2_diff_max_min_luma_coding_block_size);
        s->log2_min_luma_transform_block_size_minus2 = b.ue(HBS_SITE(0), s->log2_min_luma_transform_block_size_minus2);
        s->log2_diff_max_min_luma_transform_block_size = b.ue(HBS_SITE(0), s->log2_diff_max_min_luma_transform_block_size);
        s->max_transform_hierarchy_depth_inter = b.ue(HBS_SITE(0), s->max_transform_hierarchy_depth_inter);
        s->max_transform_hierarchy_depth_intra = b.ue(HBS_SITE(0), s->max_transform_hierarchy_depth_intra);
        const uint32_t sl = b.u1(HBS_SITE(0), s->scaling_list_enabled_flag);
        s->scaling_list_enabled_flag = sl;
        if (sl) {
            const uint32_t present = b.u1(HBS_SITE(0), s->sps_scaling_list_data_present_flag);
            s->sps_scaling_list_data_present_flag = present;
            if (present) scaling_list_data(&s->scaling_list_data);
        }
        s->amp_enabled_flag = b.u1(HBS_SITE(0), s->amp_enabled_flag);
        s->sample_adaptive_offset_enabled_flag = b.u1(HBS_SITE(0), s->sample_adaptive_offset_enabled_flag);
        const uint32_t pcm = b.u1(HBS_SITE(0), s->pcm_enabled_flag);
        s->pcm_enabled_flag = pcm;
        if (pcm) {
            s->pcm_sample_bit_depth_luma_minus1 = b.u(4, HBS_SITE(0), s->pcm_sample_bit_depth_luma_minus1);
            s->pcm_sample_bit_depth_chroma_minus1 = b.u(4, HBS_SITE(0), s->pcm_sample_bit_depth_chroma_minus1);
            s->log2_min_pcm_luma_coding_block_size_minus3 = b.ue(HBS_SITE(0), s->log2_min_pcm_luma_coding_block_size_minus3);
            s->log2_diff_max_min_pcm_luma_coding_block_size = b.ue(HBS_SITE(0), s->log2_diff_max_min_pcm_luma_coding_block_size);
            s->pcm_loop_filter_disabled_flag = b.u1(HBS_SITE(0), s->pcm_loop_filter_disabled_flag);
        }
        const int nsets = (int)b.ue(HBS_SITE(0), s->num_short_term_ref_pic_sets);
        s->num_short_term_ref_pic_sets = nsets;
        for (int i = 0; i < nsets; ++i) {
            const int k = i < MAX_NUM_SHORT_TERM_REF_PICS ? i : MAX_NUM_SHORT_TERM_REF_PICS - 1;
            st_ref_pic_set(&s->st_ref_pic_set[k], i, nsets);
        }
        const uint32_t lt = b.u1(HBS_SITE(0), s->long_term_ref_pics_present_flag);
        s->long_term_ref_pics_present_flag = lt;
        if (lt) {
            const int nlt = (int)b.ue(HBS_SITE(0), s->num_long_term_ref_pics_sps);
            s->num_long_term_ref_pics_sps = nlt;
            for (int i = 0; i < nlt; ++i) {
                const int k = in32(i) ? i : 31;
                s->lt_ref_pic_poc_lsb_sps[k] = b.u(poc_minus4 + 4, HBS_SITE(0), s->lt_ref_pic_poc_lsb_sps[k]);
                s->used_by_curr_pic_lt_sps_flag[k] = b.u1(HBS_SITE(0), s->used_by_curr_pic_lt_sps_flag[k]);
            }
        }
        s->sps_temporal_mvp_enabled_flag = b.u1(HBS_SITE(0), s->sps_temporal_mvp_enabled_flag);
        s->strong_intra_smoothing_enabled_flag = b.u1(HBS_SITE(0), s->strong_intra_smoothing_enabled_flag);
        const uint32_t vui = b.u1(HBS_SITE(0), s->vui_parameters_present_flag);
        s->vui_parameters_present_flag = vui;
        if (vui) vui_parameters(&s->vui, msl);
        const uint32_t ext = b.u1(HBS_SITE(0), s->sps_extension_present_flag);
        s->sps_extension_present_flag = ext;
        uint32_t range_ext = 0;
        if (ext) {
            range_ext = b.u1(HBS_SITE(0), s->sps_range_extension_flag);
            s->sps_range_extension_flag = range_ext;
            s->sps_multilayer_extension_flag = b.u1(HBS_SITE(0), s->sps_multilayer_extension_flag);
            s->sps_3d_extension_flag = b.u1(HBS_SITE(0), s->sps_3d_extension_flag);
            s->sps_extension_5bits = b.u(5, HBS_SITE(0), s->sps_extension_5bits);
        }
        if (range_ext) {
            hevc_sps_range_ext_t* e = &s->sps_range_ext;
            e->transform_skip_rotation_enabled_flag = b.u1(HBS_SITE(0), e->transform_skip_rotation_enabled_flag);
            e->transform_skip_context_enabled_flag = b.u1(HBS_SITE(0), e->transform_skip_context_enabled_flag);
            e->implicit_rdpcm_enabled_flag = b.u1(HBS_SITE(0), e->implicit_rdpcm_enabled_flag);
            e->explicit_rdpcm_enabled_flag = b.u1(HBS_SITE(0), e->explicit_rdpcm_enabled_flag);
            e->extended_precision_processing_flag = b.u1(HBS_SITE(0), e->extended_precision_processing_flag);
            e->intra_smoothing_disabled_flag = b.u1(HBS_SITE(0), e->intra_smoothing_disabled_flag);
            e->high_precision_offsets_enabled_flag = b.u1(HBS_SITE(0), e->high_precision_offsets_enabled_flag);
            e->persistent_rice_adaptation_enabled_flag = b.u1(HBS_SITE(0), e->persistent_rice_adaptation_enabled_flag);
            e->cabac_bypass_alignment_enabled_flag = b.u1(HBS_SITE(0), e->cabac_bypass_alignment_enabled_flag);
        }
    }

    /* ---- 7.3.2.3 (hevc_stream.c:419-521); *pps is already zero -------------------------------------- */
    HBS_M void pic_parameter_set(hevc_pps_t* p)
    {
        p->pic_parameter_set_id = b.ue(HBS_SITE(0), p->pic_parameter_set_id);
        p->seq_parameter_set_id = b.ue(HBS_SITE(0), p->seq_parameter_set_id);
        p->dependent_slice_segments_enabled_flag = b.u1(HBS_SITE(0), p->dependent_slice_segments_enabled_flag);
        p->output_flag_present_flag = b.u1(HBS_SITE(0), p->output_flag_present_flag);
        p->num_extra_slice_header_bits = b.u(3, HBS_SITE(0), p->num_extra_slice_header_bits);
        p->sign_data_hiding_enabled_flag = b.u1(HBS_SITE(0), p->sign_data_hiding_enabled_flag);
        p->cabac_init_present_flag = b.u1(HBS_SITE(0), p->cabac_init_present_flag);
        p->num_ref_idx_l0_default_active_minus1 = b.ue(HBS_SITE(0), p->num_ref_idx_l0_default_active_minus1);
        p->num_ref_idx_l1_default_active_minus1 = b.ue(HBS_SITE(0), p->num_ref_idx_l1_default_active_minus1);
        p->init_qp_minus26 = b.se(HBS_SITE(0), p->init_qp_minus26);
        p->constrained_intra_pred_flag = b.u1(HBS_SITE(0), p->constrained_intra_pred_flag);
        const uint32_t ts = b.u1(HBS_SITE(0), p->transform_skip_enabled_flag);
        p->transform_skip_enabled_flag = ts;
        const uint32_t cuqp = b.u1(HBS_SITE(0), p->cu_qp_delta_enabled_flag);
        p->cu_qp_delta_enabled_flag = cuqp;
        if (cuqp) p->diff_cu_qp_delta_depth = b.ue(HBS_SITE(0), p->diff_cu_qp_delta_depth);
        p->pps_cb_qp_offset = b.se(HBS_SITE(0), p->pps_cb_qp_offset);
        p->pps_cr_qp_offset = b.se(HBS_SITE(0), p->pps_cr_qp_offset);
        p->pps_slice_chroma_qp_offsets_present_flag = b.u1(HBS_SITE(0), p->pps_slice_chroma_qp_offsets_present_flag);
        p->weighted_pred_flag = b.u1(HBS_SITE(0), p->weighted_pred_flag);
        p->weighted_bipred_flag = b.u1(HBS_SITE(0), p->weighted_bipred_flag);
        p->transquant_bypass_enabled_flag = b.u1(HBS_SITE(0), p->transquant_bypass_enabled_flag);
        const uint32_t tiles = b.u1(HBS_SITE(0), p->tiles_enabled_flag);
        p->tiles_enabled_flag = tiles;
        p->entropy_coding_sync_enabled_flag = b.u1(HBS_SITE(0), p->entropy_coding_sync_enabled_flag);
        if (tiles) {
            const int cols = (int)b.ue(HBS_SITE(0), p->num_tile_columns_minus1), rows = (int)b.ue(HBS_SITE(1), p->num_tile_rows_minus1);
            p->num_tile_columns_minus1 = cols;
            p->num_tile_rows_minus1 = rows;
            const uint32_t uni = b.u1(HBS_SITE(0), p->uniform_spacing_flag);
            p->uniform_spacing_flag = uni;
            if (!uni) {
                for (int i = 0; i < cols; ++i) p->column_width_minus1[in32(i) ? i : 31] = b.ue(HBS_SITE(0), p->column_width_minus1[in32(i) ? i : 31]);
                for (int i = 0; i < rows; ++i) p->row_height_minus1[in32(i) ? i : 31] = b.ue(HBS_SITE(0), p->row_height_minus1[in32(i) ? i : 31]);
            }
            p->loop_filter_across_tiles_enabled_flag = b.u1(HBS_SITE(0), p->loop_filter_across_tiles_enabled_flag);
        }
        p->pps_loop_filter_across_slices_enabled_flag = b.u1(HBS_SITE(0), p->pps_loop_filter_across_slices_enabled_flag);
        const uint32_t dbc = b.u1(HBS_SITE(0), p->deblocking_filter_control_present_flag);
        p->deblocking_filter_control_present_flag = dbc;
        if (dbc) {
            p->deblocking_filter_override_enabled_flag = b.u1(HBS_SITE(0), p->deblocking_filter_override_enabled_flag);
            const uint32_t dis = b.u1(HBS_SITE(0), p->pps_deblocking_filter_disabled_flag);
            p->pps_deblocking_filter_disabled_flag = dis;
            if (dis) { p->pps_beta_offset_div2 = b.se(HBS_SITE(0), p->pps_beta_offset_div2); p->pps_tc_offset_div2 = b.se(HBS_SITE(1), p->pps_tc_offset_div2); }       /* :471 */
        }
        const uint32_t sl = b.u1(HBS_SITE(0), p->pps_scaling_list_data_present_flag);
        p->pps_scaling_list_data_present_flag = sl;
        if (sl) scaling_list_data(&p->scaling_list_data);
        p->lists_modification_present_flag = b.u1(HBS_SITE(0), p->lists_modification_present_flag);
        p->log2_parallel_merge_level_minus2 = b.ue(HBS_SITE(0), p->log2_parallel_merge_level_minus2);
        p->slice_segment_header_extension_present_flag = b.u1(HBS_SITE(0), p->slice_segment_header_extension_present_flag);
        const uint32_t ext = b.u1(HBS_SITE(0), p->pps_extension_present_flag);
        p->pps_extension_present_flag = ext;
        uint32_t range_ext = 0;
        if (ext) {
            range_ext = b.u1(HBS_SITE(0), p->pps_range_extension_flag);
            p->pps_range_extension_flag = range_ext;
            p->pps_multilayer_extension_flag = b.u1(HBS_SITE(0), p->pps_multilayer_extension_flag);
            p->pps_3d_extension_flag = b.u1(HBS_SITE(0), p->pps_3d_extension_flag);
            p->pps_extension_5bits = b.u1(HBS_SITE(0), p->pps_extension_5bits);                                                       /* :488: one bit */
        }
        if (range_ext) {
            hevc_pps_range_ext_t* e = &p->pps_range_ext;
            if (ts) e->log2_max_transform_skip_block_size_minus2 = b.ue(HBS_SITE(0), e->log2_max_transform_skip_block_size_minus2);
            e->cross_component_prediction_enabled_flag = b.u1(HBS_SITE(0), e->cross_component_prediction_enabled_flag);
            const uint32_t l = b.u1(HBS_SITE(0), e->chroma_qp_offset_list_enabled_flag);
            e->chroma_qp_offset_list_enabled_flag = l;
            if (l) {
                e->diff_cu_chroma_qp_offset_depth = b.ue(HBS_SITE(0), e->diff_cu_chroma_qp_offset_depth);
                const int n = (int)b.ue(HBS_SITE(0), e->chroma_qp_offset_list_len_minus1);
                e->chroma_qp_offset_list_len_minus1 = n;
                for (int i = 0; i <= n; ++i) {
                    const int k = in32(i) ? i : 31;
                    e->cb_qp_offset_list[k] = b.se(HBS_SITE(0), e->cb_qp_offset_list[k]);
                    e->cr_qp_offset_list[k] = b.se(HBS_SITE(0), e->cr_qp_offset_list[k]);
                }
            }
            e->log2_sao_offset_scale_luma = b.ue(HBS_SITE(0), e->log2_sao_offset_scale_luma);
            e->log2_sao_offset_scale_chroma = b.ue(HBS_SITE(0), e->log2_sao_offset_scale_chroma);
        }
        b.trailing(HBS_SITE(0), HBS_SITE(1));
    }

    /* hevc_stream.c:35-59 */
    template <class SH> HBS_M int num_pic_total_curr(const SH* sh, int sps_flag, int rps_idx, int nlt_sps, int nlt) const
    {
        int n = 0;
        const int cur = sps_flag ? rps_idx : sps->num_short_term_ref_pic_sets;
        if (in32(cur)) const_cast<ParserT*>(this)->rec_read = cur;
        if (in32(cur) && (have_rows() || (own && cur == own_idx))) {
            const int nn = numNeg(cur), np = numPos(cur);
            if (nn > 0) { const int* u = rowU0(cur); for (int i = 0; i < nn && i < 32; ++i) if (u[i]) ++n; }
            if (np > 0) { const int* u = rowU1(cur); for (int i = 0; i < np && i < 32; ++i) if (u[i]) ++n; }
        }
        for (int i = 0; i < nlt_sps + nlt && i < 32; ++i) {
            int used;
            if (i < nlt_sps) { const int k = sh->lt_idx_sps[i]; used = in32(k) ? sps->used_by_curr_pic_lt_sps_flag[k] : 0; }
            else used = sh->used_by_curr_pic_lt_flag[i];
            if (used) ++n;
        }
        return n;
    }

    /* ---- 7.3.6.3 (hevc_stream.c:969-1029) -------------------------------------------------------------- */
    template <class PWT> HBS_M void pred_weight_table(PWT* pwt, int slice_type, int l0, int l1)
    {
        pwt->luma_log2_weight_denom = b.ue(HBS_SITE(0), pwt->luma_log2_weight_denom);
        const int cat = (sps->separate_colour_plane_flag == 0) ? sps->chroma_format_idc : 0;
        if (cat != 0) pwt->delta_chroma_log2_weight_denom = b.se(HBS_SITE(0), pwt->delta_chroma_log2_weight_denom);
#define HBS_K(i) (in32(i) ? (i) : 31)
        for (int i = 0; i <= l0; ++i) pwt->luma_weight_l0_flag[HBS_K(i)] = b.u1(HBS_SITE(0), pwt->luma_weight_l0_flag[HBS_K(i)]);
        if (cat != 0) for (int i = 0; i <= l0; ++i) pwt->chroma_weight_l0_flag[HBS_K(i)] = b.u1(HBS_SITE(0), pwt->chroma_weight_l0_flag[HBS_K(i)]);
        for (int i = 0; i <= l0; ++i) {
            if (pwt->luma_weight_l0_flag[HBS_K(i)]) { pwt->delta_luma_weight_l0[HBS_K(i)] = b.se(HBS_SITE(0), pwt->delta_luma_weight_l0[HBS_K(i)]); pwt->luma_offset_l0[HBS_K(i)] = b.se(HBS_SITE(1), pwt->luma_offset_l0[HBS_K(i)]); }
            if (pwt->chroma_weight_l0_flag[HBS_K(i)])
                for (int j = 0; j < 2; ++j) { pwt->delta_chroma_weight_l0[HBS_K(i)][j] = b.se(HBS_SITE(0), pwt->delta_chroma_weight_l0[HBS_K(i)][j]); pwt->delta_chroma_offset_l0[HBS_K(i)][j] = b.se(HBS_SITE(1), pwt->delta_chroma_offset_l0[HBS_K(i)][j]); }
        }
        if (slice_type == HEVC_SLICE_TYPE_B) {
            for (int i = 0; i <= l1; ++i) pwt->luma_weight_l1_flag[HBS_K(i)] = b.u1(HBS_SITE(0), pwt->luma_weight_l1_flag[HBS_K(i)]);
            if (cat != 0) for (int i = 0; i <= l1; ++i) pwt->chroma_weight_l1_flag[HBS_K(i)] = b.u1(HBS_SITE(0), pwt->chroma_weight_l1_flag[HBS_K(i)]);
            for (int i = 0; i <= l1; ++i) {
                if (pwt->luma_weight_l1_flag[HBS_K(i)]) { pwt->delta_luma_weight_l1[HBS_K(i)] = b.se(HBS_SITE(0), pwt->delta_luma_weight_l1[HBS_K(i)]); pwt->luma_offset_l1[HBS_K(i)] = b.se(HBS_SITE(1), pwt->luma_offset_l1[HBS_K(i)]); }
                if (pwt->chroma_weight_l1_flag[HBS_K(i)])
                    for (int j = 0; j < 2; ++j) { pwt->delta_chroma_weight_l1[HBS_K(i)][j] = b.se(HBS_SITE(0), pwt->delta_chroma_weight_l1[HBS_K(i)][j]); pwt->delta_chroma_offset_l1[HBS_K(i)][j] = b.se(HBS_SITE(1), pwt->delta_chroma_offset_l1[HBS_K(i)][j]); }
            }
        }
#undef HBS_K
    }

    /* ---- 7.3.6 (hevc_stream.c:782-966); *sh is zero except collocated_from_l0_flag = 1 (:19-24) ----------- */
    template <class SH> HBS_M void slice_segment_header(SH* sh, int nal_unit_type, const hevc_pps_t* last_pps,
                                    const hevc_sps_t* last_sps, const hevc_pps_t* zero_pps, const hevc_sps_t* zero_sps)
    {
        const uint32_t first = b.u1(HBS_SITE(0), sh->first_slice_segment_in_pic_flag);
        sh->first_slice_segment_in_pic_flag = first;
        if (nal_unit_type >= HEVC_NAL_UNIT_TYPE_BLA_W_LP && nal_unit_type <= HEVC_NAL_UNIT_TYPE_RSV_IRAP_VCL23)
            sh->no_output_of_prior_pics_flag = b.u1(HBS_SITE(0), sh->no_output_of_prior_pics_flag);
        const int pps_id = (int)b.ue(HBS_SITE(0), sh->pic_parameter_set_id);
        sh->pic_parameter_set_id = pps_id;
        pps = (pps_id == 0) ? last_pps : zero_pps;                                                     /* :800 */
        sps = (pps->seq_parameter_set_id == 0) ? last_sps : zero_sps;                                   /* :801 */
        own_idx = sps->num_short_term_ref_pic_sets;

        int l0 = pps->num_ref_idx_l0_default_active_minus1, l1 = pps->num_ref_idx_l1_default_active_minus1;
        sh->num_ref_idx_l0_active_minus1 = l0;
        sh->num_ref_idx_l1_active_minus1 = l1;
        uint32_t dependent = 0;
        if (!first) {
            if (pps->dependent_slice_segments_enabled_flag) { dependent = b.u1(HBS_SITE(0), sh->dependent_slice_segment_flag); sh->dependent_slice_segment_flag = dependent; }
            sh->slice_segment_address = b.u(slice_address_bits(sps), HBS_SITE(0), sh->slice_segment_address);
        }
        if (!dependent) {
            for (int i = 0; i < pps->num_extra_slice_header_bits; ++i) b.skip(1, HBS_SITE(0), 1u);
            const int slice_type = (int)b.ue(HBS_SITE(0), sh->slice_type);
            sh->slice_type = slice_type;
            if (pps->output_flag_present_flag) sh->pic_output_flag = b.u1(HBS_SITE(0), sh->pic_output_flag);
            if (sps->separate_colour_plane_flag == 1) sh->colour_plane_id = b.u(2, HBS_SITE(0), sh->colour_plane_id);
            int sps_flag = 0, rps_idx = 0, nlt_sps = 0, nlt = 0;
            uint32_t tmvp = 0;
            if (nal_unit_type != HEVC_NAL_UNIT_TYPE_IDR_W_RADL && nal_unit_type != HEVC_NAL_UNIT_TYPE_IDR_N_LP) {
                const int poc_bits = sps->log2_max_pic_order_cnt_lsb_minus4 + 4;
                sh->slice_pic_order_cnt_lsb = b.u(poc_bits, HBS_SITE(0), sh->slice_pic_order_cnt_lsb);
                sps_flag = b.u1(HBS_SITE(0), sh->short_term_ref_pic_set_sps_flag);
                sh->short_term_ref_pic_set_sps_flag = sps_flag;
                if (!sps_flag) {
                    st_ref_pic_set(&sh->st_ref_pic_set, sps->num_short_term_ref_pic_sets, sps->num_short_term_ref_pic_sets); if (own && own_idx == 0 && last_sps->num_short_term_ref_pic_sets > 0) diverged = 1;
                    if (stop_after_rps) return;
                } else if (sps->num_short_term_ref_pic_sets > 1) {
                    rps_idx = b.u(ceil_log2_int(sps->num_short_term_ref_pic_sets), HBS_SITE(0), sh->short_term_ref_pic_set_idx);
                    sh->short_term_ref_pic_set_idx = rps_idx; } if (own && sps_flag && in32(rps_idx) && rps_idx >= last_sps->num_short_term_ref_pic_sets) { diverged = 1;
                }
                if (sps->long_term_ref_pics_present_flag) {
                    if (sps->num_long_term_ref_pics_sps > 0) { nlt_sps = (int)b.ue(HBS_SITE(0), sh->num_long_term_sps); sh->num_long_term_sps = nlt_sps; }
                    nlt = (int)b.ue(HBS_SITE(0), sh->num_long_term_pics);
                    sh->num_long_term_pics = nlt;
                    for (int i = 0; i < nlt_sps + nlt; ++i) {
                        const int k = in32(i) ? i : 31;
                        if (i < nlt_sps) {
                            if (sps->num_long_term_ref_pics_sps > 1) sh->lt_idx_sps[k] = b.u(ceil_log2_int(sps->num_long_term_ref_pics_sps), HBS_SITE(0), sh->lt_idx_sps[k]);
                        } else {
                            sh->poc_lsb_lt[k] = b.u(poc_bits, HBS_SITE(0), sh->poc_lsb_lt[k]);
                            sh->used_by_curr_pic_lt_flag[k] = b.u1(HBS_SITE(0), sh->used_by_curr_pic_lt_flag[k]);
                        }
                        const uint32_t msb = b.u1(HBS_SITE(0), sh->delta_poc_msb_present_flag[k]);
                        sh->delta_poc_msb_present_flag[k] = msb;
                        if (msb) sh->delta_poc_msb_cycle_lt[k] = b.ue(HBS_SITE(0), sh->delta_poc_msb_cycle_lt[k]);
                    }
                }
                if (sps->sps_temporal_mvp_enabled_flag) { tmvp = b.u1(HBS_SITE(0), sh->slice_temporal_mvp_enabled_flag); sh->slice_temporal_mvp_enabled_flag = tmvp; }
            }
            uint32_t sao_l = 0, sao_c = 0;
            if (sps->sample_adaptive_offset_enabled_flag) {
                sao_l = b.u1(HBS_SITE(0), sh->slice_sao_luma_flag);
                sh->slice_sao_luma_flag = sao_l;
                const int cat = (sps->separate_colour_plane_flag == 0) ? sps->chroma_format_idc : 0;
                if (cat != 0) { sao_c = b.u1(HBS_SITE(0), sh->slice_sao_chroma_flag); sh->slice_sao_chroma_flag = sao_c; }
            }
            if (slice_type == HEVC_SLICE_TYPE_P || slice_type == HEVC_SLICE_TYPE_B) {
                const uint32_t ov = b.u1(HBS_SITE(0), sh->num_ref_idx_active_override_flag);
                sh->num_ref_idx_active_override_flag = ov;
                if (ov) {
                    l0 = (int)b.ue(HBS_SITE(0), sh->num_ref_idx_l0_active_minus1); sh->num_ref_idx_l0_active_minus1 = l0;
                    if (slice_type == HEVC_SLICE_TYPE_B) { l1 = (int)b.ue(HBS_SITE(0), sh->num_ref_idx_l1_active_minus1); sh->num_ref_idx_l1_active_minus1 = l1; }
                }
                if (pps->lists_modification_present_flag) {
                    const int npc = num_pic_total_curr(sh, sps_flag, rps_idx, nlt_sps, nlt); if (own && !sps_flag && (nal_unit_type == HEVC_NAL_UNIT_TYPE_IDR_W_RADL || nal_unit_type == HEVC_NAL_UNIT_TYPE_IDR_N_LP)) diverged = 1;   /* an IDR with a P/B type reads the row an EARLIER slice's own set left */
                    if (npc > 1) {                                                   /* :944-966; list1's flag is never read */
                        const uint32_t m0 = b.u1(HBS_SITE(0), sh->rpld.ref_pic_list_modification_flag_l0);
                        sh->rpld.ref_pic_list_modification_flag_l0 = m0;
                        if (m0) for (int i = 0; i <= l0; ++i) sh->rpld.list_entry_l0[in32(i) ? i : 31] = b.u(ceil_log2_int(npc), HBS_SITE(0), sh->rpld.list_entry_l0[in32(i) ? i : 31]);
                        /* where list1's flag would be read the reference's debug reader prints a cursor and nothing else (:3147) */
                        if (slice_type == HEVC_SLICE_TYPE_B) b.mark(HBS_SITE(0));
                    }
                }
                if (slice_type == HEVC_SLICE_TYPE_B) sh->mvd_l1_zero_flag = b.u1(HBS_SITE(0), sh->mvd_l1_zero_flag);
                if (pps->cabac_init_present_flag) sh->cabac_init_flag = b.u1(HBS_SITE(0), sh->cabac_init_flag);
                if (tmvp) {
                    uint32_t col_l0 = 1;
                    if (slice_type == HEVC_SLICE_TYPE_B) { col_l0 = b.u1(HBS_SITE(0), sh->collocated_from_l0_flag); sh->collocated_from_l0_flag = col_l0; }
                    if ((col_l0 && l0 > 0) || (!col_l0 && l1 > 0)) sh->collocated_ref_idx = b.ue(HBS_SITE(0), sh->collocated_ref_idx);
                }
                if ((pps->weighted_pred_flag && slice_type == HEVC_SLICE_TYPE_P) ||
                    (pps->weighted_bipred_flag && slice_type == HEVC_SLICE_TYPE_B))
                    pred_weight_table(&sh->pwt, slice_type, l0, l1);
                sh->five_minus_max_num_merge_cand = b.ue(HBS_SITE(0), sh->five_minus_max_num_merge_cand);
            }
            sh->slice_qp_delta = b.se(HBS_SITE(0), sh->slice_qp_delta);
            if (pps->pps_slice_chroma_qp_offsets_present_flag) { sh->slice_cb_qp_offset = b.se(HBS_SITE(0), sh->slice_cb_qp_offset); sh->slice_cr_qp_offset = b.se(HBS_SITE(1), sh->slice_cr_qp_offset); }
            if (pps->pps_range_ext.chroma_qp_offset_list_enabled_flag) sh->cu_chroma_qp_offset_enabled_flag = b.u1(HBS_SITE(0), sh->cu_chroma_qp_offset_enabled_flag);
            uint32_t dbo = 0, dis = 0;
            if (pps->deblocking_filter_override_enabled_flag) { dbo = b.u1(HBS_SITE(0), sh->deblocking_filter_override_flag); sh->deblocking_filter_override_flag = dbo; }
            if (dbo) {
                dis = b.u1(HBS_SITE(0), sh->slice_deblocking_filter_disabled_flag);
                sh->slice_deblocking_filter_disabled_flag = dis;
                if (!dis) { sh->slice_beta_offset_div2 = b.se(HBS_SITE(0), sh->slice_beta_offset_div2); sh->slice_tc_offset_div2 = b.se(HBS_SITE(1), sh->slice_tc_offset_div2); }
            }
            if (pps->pps_loop_filter_across_slices_enabled_flag && (sao_l || sao_c || !dis))
                sh->slice_loop_filter_across_slices_enabled_flag = b.u1(HBS_SITE(0), sh->slice_loop_filter_across_slices_enabled_flag);
        }
        if (pps->tiles_enabled_flag || pps->entropy_coding_sync_enabled_flag) {
            const int n = (int)b.ue(HBS_SITE(0), sh->num_entry_point_offsets);
            sh->num_entry_point_offsets = n;
            if (n > 0) {
                const int ol = (int)b.ue(HBS_SITE(0), sh->offset_len_minus1);
                sh->offset_len_minus1 = ol;
                for (int i = 0; i < n; ++i) {
                    const int v = b.u(ol + 1, HBS_SITE(0), i < MAX_NUM_ENTRY_POINT_OFFSET ? sh->entry_point_offset_minus1[i] : 0);
                    if (i < MAX_NUM_ENTRY_POINT_OFFSET) sh->entry_point_offset_minus1[i] = v;
                }
            }
        }
        if (pps->slice_segment_header_extension_present_flag) {
            const int n = (int)b.ue(HBS_SITE(0), sh->slice_segment_header_extension_length);
            sh->slice_segment_header_extension_length = n;
            for (int i = 0; i < n; ++i) b.skip(8, HBS_SITE(0), 0u);
        }
        b.trailing(HBS_SITE(0), HBS_SITE(1));                                                                    /* byte_alignment() */
    }
};

HBS_HD bool is_slice_type_nal(int t) { return (t >= 0 && t <= 9) || (t >= 16 && t <= 21); }

/* bytes of the struct a NAL of this type parses into (0: none) */
HBS_HD uint32_t struct_bytes_of(int nal_unit_type)
{
    if (is_slice_type_nal(nal_unit_type)) return (uint32_t)sizeof(hevc_slice_header_t);
    if (nal_unit_type == HEVC_NAL_UNIT_TYPE_VPS_NUT) return (uint32_t)sizeof(hevc_vps_t);
    if (nal_unit_type == HEVC_NAL_UNIT_TYPE_SPS_NUT) return (uint32_t)sizeof(hevc_sps_t);
    if (nal_unit_type == HEVC_NAL_UNIT_TYPE_PPS_NUT) return (uint32_t)sizeof(hevc_pps_t);
    return 0u;
}

HBS_HD uint64_t round16(uint64_t v) { return (v + 15ull) & ~15ull; }

/* arena slot of a NAL: its struct; an SPS is followed by its derived RPS tables */
HBS_HD uint64_t slot_bytes_of(int type)
{
    const uint64_t s = struct_bytes_of(type);
    if (type == HEVC_NAL_UNIT_TYPE_SPS_NUT) return round16(s) + round16(sizeof(RpsTables));
    return round16(s);
}

/* NAL header from RBSP bytes 0-1 (hevc_stream.c:176-179; bits past the end read as 0) */
HBS_HD void nal_header_of(const uint8_t* rbsp, uint32_t len, ParsedNal& p)
{
    const uint32_t b0 = len > 0 ? rbsp[0] : 0u, b1 = len > 1 ? rbsp[1] : 0u;
    p.nal_unit_type = (int32_t)((b0 >> 1) & 0x3Fu);
    p.nal_layer_id = (int32_t)(((b0 & 1u) << 5) | (b1 >> 3));
    p.nal_temporal_id_plus1 = (int32_t)(b1 & 7u);
}

/*
 * Lane-0 part of reading one NAL whose struct slot has been prepared (zeroed;
 * slices: collocated_from_l0_flag = 1).  Mirrors read_hevc_nal_unit
 * (hevc_stream.c:175-239) after nal_to_rbsp.  `consumed` = what nal_to_rbsp
 * reported (NAL length, or length-1 with a dropped trailing 03).
 */
template <int kMode>
HBS_D void parse_one_nal(ParserT<kMode>& ps, int nal_unit_type, void* slot, int consumed, ParsedNal* out,
                         const hevc_pps_t* last_pps, const hevc_sps_t* last_sps,
                         const hevc_pps_t* zero_pps, const hevc_sps_t* zero_sps)
{
    out->slice_data_size = 0;
    out->slice_data_off = 0;
    if (is_slice_type_nal(nal_unit_type)) {
        ps.slice_segment_header(static_cast<hevc_slice_header_t*>(slot), nal_unit_type, last_pps, last_sps, zero_pps, zero_sps);
        /* hevc_stream.c:608-616: payload starts one byte after the cursor; then one more "trailing" byte is consumed */
        out->slice_data_off = (ps.b.pos >> 3) + 1u;
        out->slice_data_size = (int32_t)ps.b.size - (int32_t)(ps.b.pos >> 3) - 1;
        ps.b.trailing(HBS_SITE(0), HBS_SITE(1));
    } else if (nal_unit_type == HEVC_NAL_UNIT_TYPE_VPS_NUT) {
        ps.video_parameter_set(static_cast<hevc_vps_t*>(slot));
    } else if (nal_unit_type == HEVC_NAL_UNIT_TYPE_SPS_NUT) {
        ps.seq_parameter_set(static_cast<hevc_sps_t*>(slot));
    } else if (nal_unit_type == HEVC_NAL_UNIT_TYPE_PPS_NUT) {
        ps.pic_parameter_set(static_cast<hevc_pps_t*>(slot));
    } else {
        out->rc = -1;                                                   /* :221-222 */
        return;
    }
    out->rc = ps.b.overrun() ? -1 : consumed;                            /* :225, :239 */
}

/* per-NAL result of the syntax writers */
struct WrittenNal {
    int32_t rc;                    /* 0, or -1: unsupported type / wrote past the buffer (hevc_stream.c:1309-1312) */
    uint32_t rbsp_size;            /* bs_pos(): whole bytes written, at most the buffer (bs.h:121)                */
    int32_t slice_data_size;       /* what the writer leaves in h->slice_data->rbsp_size (hevc_stream.c:1699-1706) */
    uint32_t pad;
};

/*
 * Lane-0 part of writing one NAL from its struct: write_hevc_nal_unit (hevc_stream.c:1249-1327) up
 * to, not including, rbsp_to_nal.  ps.b is set up for writing into a zeroed buffer of `size * 3 / 4`
 * bytes (:1265).  The walk is the readers' (same conditions, same derived tables), each element
 * putting the struct field the read would fill; so the reference's writer quirks come for free: an
 * SPS gets no trailing bits and loses the bits of its last, unfinished byte; a slice is its header
 * plus trailing bits; the active reference counts are written as the PPS defaults (:780 runs in
 * every mode); one bit for sub_layer_level_idc (:1845).
 */
HBS_D void write_one_nal(ParserT<kModeWrite>& ps, int nal_unit_type, int nal_layer_id, int nal_temporal_id_plus1, void* slot,
                         const hevc_pps_t* last_pps, const hevc_sps_t* last_sps,
                         const hevc_pps_t* zero_pps, const hevc_sps_t* zero_sps, WrittenNal* out)
{
    out->slice_data_size = 0;
    ps.b.put_bits(1, 0u);
    ps.b.put_bits(6, (uint32_t)nal_unit_type);
    ps.b.put_bits(6, (uint32_t)nal_layer_id);
    ps.b.put_bits(3, (uint32_t)nal_temporal_id_plus1);
    if (is_slice_type_nal(nal_unit_type)) {
        ps.slice_segment_header(static_cast<hevc_slice_header_t*>(slot), nal_unit_type, last_pps, last_sps, zero_pps, zero_sps);
        out->slice_data_size = (int32_t)ps.b.size - (int32_t)(ps.b.pos >> 3) - 1;      /* b->end - (b->p + 1) */
        ps.b.trailing(0u, 0u);
    } else if (nal_unit_type == HEVC_NAL_UNIT_TYPE_VPS_NUT) {
        ps.video_parameter_set(static_cast<hevc_vps_t*>(slot));
    } else if (nal_unit_type == HEVC_NAL_UNIT_TYPE_SPS_NUT) {
        ps.seq_parameter_set(static_cast<hevc_sps_t*>(slot));
    } else if (nal_unit_type == HEVC_NAL_UNIT_TYPE_PPS_NUT) {
        ps.pic_parameter_set(static_cast<hevc_pps_t*>(slot));
    } else {
        out->rc = -1; out->rbsp_size = 0;
        return;
    }
    out->rc = ps.b.overrun() ? -1 : 0;
    const uint32_t whole = ps.b.pos >> 3;
    out->rbsp_size = whole > ps.b.size ? ps.b.size : whole;
}

} // namespace hbs
#endif
