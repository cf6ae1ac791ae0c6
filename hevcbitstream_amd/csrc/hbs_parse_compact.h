/*
 * hbs_parse_compact.h -- the slice-header walk of hbs_parse.h into a SINK instead of a hevc_slice_header_t
 * (hbs_parse_headers_compact, round 5).
 *
 * K4 writes one hevc_slice_header_t per slice: 4 024 bytes, cleared first (the reference's memset, hevc_stream.c:19-24)
 * and then a few dozen members -- 405 MB for the 100 k NALs of a 4K30 stream, more than half of BASELINE config 3's time,
 * for a struct of which a caller that indexes a stream looks at a handful of members (reference struct hevc_stream.h:465-515,
 * reader hevc_stream.c:782-941).  The walk itself needs almost nothing of it back: every condition is taken from a local
 * (hbs_parse.h keeps them), and only two places read members again -- NumPicTotalCurr (:35-59) looks at lt_idx_sps[] and
 * used_by_curr_pic_lt_flag[], pred_weight_table (:969-1029) at its four flag arrays.  So the same walk, instantiated on
 * SliceSink, runs without a struct: members nobody reads are Discard (assignments vanish), the flag arrays are 32-bit
 * masks, lt_idx_sps stays an array, and the members of the compact record are plain ints.  Same bits read in the same
 * order, same cursor, same rc / slice_data_off / slice_data_size; hbs_parse_materialize walks a NAL into the full struct
 * when a caller wants it.
 *
 * Included behind hbs_parse.h (whose line numbers are the trace's site ids: nothing is added there).
 */
#ifndef HBS_PARSE_COMPACT_H
#define HBS_PARSE_COMPACT_H

#include "hbs_parse.h"

namespace hbs {

/* a member nobody reads back: assignments vanish, a read (the writer's "current value" argument, unused when reading) is 0 */
struct Discard {
    template <class T> HBS_HD const Discard& operator=(const T&) const { return *this; }
    HBS_HD operator int() const { return 0; }
};
struct DiscardArr {
    Discard d;
    HBS_HD Discard& operator[](int) { return d; }
    HBS_HD const Discard& operator[](int) const { return d; }
};
struct DiscardArr2 {
    DiscardArr a;
    HBS_HD DiscardArr& operator[](int) { return a; }
    HBS_HD const DiscardArr& operator[](int) const { return a; }
};
/* 32 one-bit members that ARE read back */
struct Bits32 {
    uint32_t bits;
    struct Ref {
        uint32_t* w; int k;
        template <class T> HBS_HD const Ref& operator=(const T& v) const { *w = (*w & ~(1u << k)) | (((uint32_t)v & 1u) << k); return *this; }
        HBS_HD operator int() const { return (int)((*w >> k) & 1u); }
    };
    HBS_HD Ref operator[](int k) { Ref r; r.w = &bits; r.k = k & 31; return r; }
    HBS_HD int operator[](int k) const { return (int)((bits >> (k & 31)) & 1u); }
};

/* 32 small integers that ARE read back, a byte each in eight registers (an int[32] indexed by a variable lives in scratch memory):
 * what reads them back (NumPicTotalCurr, hevc_stream.c:35-59) only asks "is it 0..31, and which" -- anything else, negative
 * values included, is stored as 255 */
struct Bytes32 {
    uint32_t w0, w1, w2, w3, w4, w5, w6, w7;
    HBS_HD void clear() { w0 = w1 = w2 = w3 = w4 = w5 = w6 = w7 = 0u; }
    HBS_HD uint32_t word(int j) const { return j == 0 ? w0 : j == 1 ? w1 : j == 2 ? w2 : j == 3 ? w3 : j == 4 ? w4 : j == 5 ? w5 : j == 6 ? w6 : w7; }
    HBS_HD void put(int k, int v)
    {
        const uint32_t b = (v >= 0 && v < 255) ? (uint32_t)v : 255u, sh = 8u * ((uint32_t)k & 3u), m = ~(0xFFu << sh), j = ((uint32_t)k >> 2) & 7u;
        if (j == 0) w0 = (w0 & m) | (b << sh); else if (j == 1) w1 = (w1 & m) | (b << sh); else if (j == 2) w2 = (w2 & m) | (b << sh);
        else if (j == 3) w3 = (w3 & m) | (b << sh); else if (j == 4) w4 = (w4 & m) | (b << sh); else if (j == 5) w5 = (w5 & m) | (b << sh);
        else if (j == 6) w6 = (w6 & m) | (b << sh); else w7 = (w7 & m) | (b << sh);
    }
    struct Ref {
        Bytes32* p; int k;
        template <class T> HBS_HD const Ref& operator=(const T& v) const { p->put(k, (int)v); return *this; }
        HBS_HD operator int() const { return (int)((p->word((k >> 2) & 7) >> (8u * ((uint32_t)k & 3u))) & 0xFFu); }
    };
    HBS_HD Ref operator[](int k) { Ref r; r.p = this; r.k = k & 31; return r; }
    HBS_HD int operator[](int k) const { return (int)((word((k >> 2) & 7) >> (8u * ((uint32_t)k & 3u))) & 0xFFu); }
};

/* hevc_st_ref_pic_set_t, hevc_ref_pics_lists_mod_t as sinks: the walk reads nothing back from them */
#define HBS_SINK_F(name)            Discard name;
#define HBS_SINK_A(name, n)         DiscardArr name;
#define HBS_SINK_A2(name, n, m)     DiscardArr2 name;
#define HBS_SINK_S(type, name)      static_assert(sizeof(type) == 0, "no nested struct expected here");
#define HBS_SINK_SA(type, name, n)  static_assert(sizeof(type) == 0, "no nested struct expected here");
struct StRpsSink { HBS_ST_RPS_FIELDS(HBS_SINK_F, HBS_SINK_A, HBS_SINK_A2, HBS_SINK_S, HBS_SINK_SA) };
struct RplmSink { HBS_RPLM_FIELDS(HBS_SINK_F, HBS_SINK_A, HBS_SINK_A2, HBS_SINK_S, HBS_SINK_SA) };
#undef HBS_SINK_F
#undef HBS_SINK_A
#undef HBS_SINK_A2
#undef HBS_SINK_S
#undef HBS_SINK_SA

/* hevc_pred_weight_table_t (HBS_PWT_FIELDS): the four flag arrays are read back by the walk (hevc_stream.c:987-1027) */
struct PwtSink {
    Discard luma_log2_weight_denom, delta_chroma_log2_weight_denom;
    Bits32 luma_weight_l0_flag, chroma_weight_l0_flag;
    DiscardArr delta_luma_weight_l0, luma_offset_l0;
    DiscardArr2 delta_chroma_weight_l0, delta_chroma_offset_l0;
    Bits32 luma_weight_l1_flag, chroma_weight_l1_flag;
    DiscardArr delta_luma_weight_l1, luma_offset_l1;
    DiscardArr2 delta_chroma_weight_l1, delta_chroma_offset_l1;
};

/* hevc_slice_header_t (HBS_SLICE_HEADER_FIELDS, same member names): ints where the compact record wants the value or the
 * walk reads it back, sinks elsewhere */
struct SliceSink {
    int first_slice_segment_in_pic_flag, no_output_of_prior_pics_flag, pic_parameter_set_id, dependent_slice_segment_flag;
    int slice_segment_address, slice_type, pic_output_flag;
    Discard colour_plane_id;
    int slice_pic_order_cnt_lsb, short_term_ref_pic_set_sps_flag;
    StRpsSink st_ref_pic_set;
    int short_term_ref_pic_set_idx;
    Discard num_long_term_sps;
    int num_long_term_pics;
    Bytes32 lt_idx_sps;                                         /* read back by NumPicTotalCurr */
    DiscardArr poc_lsb_lt;
    Bits32 used_by_curr_pic_lt_flag;                            /* read back by NumPicTotalCurr */
    DiscardArr delta_poc_msb_present_flag, delta_poc_msb_cycle_lt;
    int slice_temporal_mvp_enabled_flag;
    Discard slice_sao_luma_flag, slice_sao_chroma_flag, num_ref_idx_active_override_flag;
    int num_ref_idx_l0_active_minus1, num_ref_idx_l1_active_minus1;
    RplmSink rpld;
    Discard mvd_l1_zero_flag, cabac_init_flag, collocated_from_l0_flag, collocated_ref_idx;
    PwtSink pwt;
    Discard five_minus_max_num_merge_cand;
    int slice_qp_delta;
    Discard slice_cb_qp_offset, slice_cr_qp_offset, cu_chroma_qp_offset_enabled_flag, deblocking_filter_override_flag;
    Discard slice_deblocking_filter_disabled_flag, slice_beta_offset_div2, slice_tc_offset_div2, slice_loop_filter_across_slices_enabled_flag;
    int num_entry_point_offsets;
    Discard offset_len_minus1;
    DiscardArr entry_point_offset_minus1;
    Discard slice_segment_header_extension_length;
};

/* the compact record of a slice (layout = hbs_slice_compact in the public header): sixteen members of hevc_slice_header_t,
 * each what the full struct holds under the same name (0 where the header does not code it) */
struct SliceCompact {
    int32_t first_slice_segment_in_pic_flag, no_output_of_prior_pics_flag, pic_parameter_set_id, dependent_slice_segment_flag;
    int32_t slice_segment_address, slice_type, pic_output_flag, slice_pic_order_cnt_lsb;
    int32_t short_term_ref_pic_set_sps_flag, short_term_ref_pic_set_idx, num_long_term_pics, slice_temporal_mvp_enabled_flag;
    int32_t num_ref_idx_l0_active_minus1, num_ref_idx_l1_active_minus1, slice_qp_delta, num_entry_point_offsets;
};
static_assert(sizeof(SliceCompact) == 64, "one 64-byte record per NAL");

template <class SH>
HBS_HD SliceCompact compact_of(const SH& s)
{
    SliceCompact c;
    c.first_slice_segment_in_pic_flag = s.first_slice_segment_in_pic_flag; c.no_output_of_prior_pics_flag = s.no_output_of_prior_pics_flag;
    c.pic_parameter_set_id = s.pic_parameter_set_id; c.dependent_slice_segment_flag = s.dependent_slice_segment_flag;
    c.slice_segment_address = s.slice_segment_address; c.slice_type = s.slice_type; c.pic_output_flag = s.pic_output_flag;
    c.slice_pic_order_cnt_lsb = s.slice_pic_order_cnt_lsb; c.short_term_ref_pic_set_sps_flag = s.short_term_ref_pic_set_sps_flag;
    c.short_term_ref_pic_set_idx = s.short_term_ref_pic_set_idx; c.num_long_term_pics = s.num_long_term_pics;
    c.slice_temporal_mvp_enabled_flag = s.slice_temporal_mvp_enabled_flag;
    c.num_ref_idx_l0_active_minus1 = s.num_ref_idx_l0_active_minus1; c.num_ref_idx_l1_active_minus1 = s.num_ref_idx_l1_active_minus1;
    c.slice_qp_delta = s.slice_qp_delta; c.num_entry_point_offsets = s.num_entry_point_offsets;
    return c;
}
HBS_HD SliceCompact compact_zero()
{
    SliceCompact c;
    c.first_slice_segment_in_pic_flag = c.no_output_of_prior_pics_flag = c.pic_parameter_set_id = c.dependent_slice_segment_flag = 0;
    c.slice_segment_address = c.slice_type = c.pic_output_flag = c.slice_pic_order_cnt_lsb = 0;
    c.short_term_ref_pic_set_sps_flag = c.short_term_ref_pic_set_idx = c.num_long_term_pics = c.slice_temporal_mvp_enabled_flag = 0;
    c.num_ref_idx_l0_active_minus1 = c.num_ref_idx_l1_active_minus1 = c.slice_qp_delta = c.num_entry_point_offsets = 0;
    return c;
}

/* parse_one_nal's slice branch (hbs_parse.h) into a sink: same walk, same cursor, same record */
template <int kMode>
HBS_D void parse_slice_into_sink(ParserT<kMode>& ps, int nal_unit_type, int consumed, ParsedNal* out, SliceCompact* compact,
                                 const hevc_pps_t* last_pps, const hevc_sps_t* last_sps,
                                 const hevc_pps_t* zero_pps, const hevc_sps_t* zero_sps)
{
    SliceSink s;
    s.first_slice_segment_in_pic_flag = s.no_output_of_prior_pics_flag = s.pic_parameter_set_id = s.dependent_slice_segment_flag = 0;
    s.slice_segment_address = s.slice_type = s.pic_output_flag = s.slice_pic_order_cnt_lsb = s.short_term_ref_pic_set_sps_flag = 0;
    s.short_term_ref_pic_set_idx = s.num_long_term_pics = s.slice_temporal_mvp_enabled_flag = 0;
    s.num_ref_idx_l0_active_minus1 = s.num_ref_idx_l1_active_minus1 = s.slice_qp_delta = s.num_entry_point_offsets = 0;
    s.used_by_curr_pic_lt_flag.bits = 0u;
    s.pwt.luma_weight_l0_flag.bits = s.pwt.chroma_weight_l0_flag.bits = s.pwt.luma_weight_l1_flag.bits = s.pwt.chroma_weight_l1_flag.bits = 0u;
    s.lt_idx_sps.clear();
    ps.slice_segment_header(&s, nal_unit_type, last_pps, last_sps, zero_pps, zero_sps);
    out->slice_data_off = (ps.b.pos >> 3) + 1u;                              /* hevc_stream.c:608-616, as parse_one_nal */
    out->slice_data_size = (int32_t)ps.b.size - (int32_t)(ps.b.pos >> 3) - 1;
    ps.b.trailing(0u, 0u);                                                   /* (never traced: the compact parse has no trace mode) */
    out->rc = ps.b.overrun() ? -1 : consumed;
    *compact = compact_of(s);
}

} // namespace hbs
#endif
