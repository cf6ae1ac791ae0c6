/*
 * hevc_stream.h -- data model and legacy entry points of the HEVC header layer,
 * ABI-compatible with leslie-wang/hevcbitstream (reference hevc_stream.h:21-574).
 *
 * An application written against the reference (its hevc_analyze.c included)
 * compiles and links against this header + libhevcbitstream unchanged: every
 * struct below has the reference's member names, order and sizes (all members
 * are `int`; sizes are asserted against the reference's in
 * tests/golden/abi_layout.json).  Unlike the reference, the layout is written
 * once as field tables (HBS_*_FIELDS): the same tables give the C structs here,
 * the field dumps of the parity tests, and the name strings of the
 * read_debug trace.
 *
 *   F(name)            int name;
 *   A(name, n)         int name[n];
 *   A2(name, n, m)     int name[n][m];
 *   S(type, name)      type name;
 *   SA(type, name, n)  type name[n];
 */
#ifndef _HEVC_STREAM_H
#define _HEVC_STREAM_H        1

#include <stdint.h>
#include <stdio.h>

#include "bs.h"
#include "h264_sei.h"

#ifdef __cplusplus
extern "C" {
#endif

/* array bounds: reference hevc_stream.h:21-35 */
#define MAX_NUM_SUBLAYERS            32
#define MAX_NUM_HRD_PARAM            10
#define MAX_CPB_CNT                  32
#define MAX_NUM_NEGATIVE_PICS        32
#define MAX_NUM_POSITIVE_PICS        32
#define MAX_NUM_REF_PICS_L0          32
#define MAX_NUM_REF_PICS_L1          32
#define MAX_NUM_SHORT_TERM_REF_PICS  32
#define MAX_NUM_LONG_TERM_REF_PICS   32
#define MAX_NUM_PALLETTE_PREDICTOR   32
#define MAX_NUM_CHROMA_QP_OFFSET_LST 32
#define MAX_NUM_ENTRY_POINT_OFFSET   32
#define MAX_NUM_TILE_COLUMN          32
#define MAX_NUM_TILE_ROW             32

#define HBS_DECL_F(name)            int name;
#define HBS_DECL_A(name, n)         int name[n];
#define HBS_DECL_A2(name, n, m)     int name[n][m];
#define HBS_DECL_S(type, name)      type name;
#define HBS_DECL_SA(type, name, n)  type name[n];
#define HBS_STRUCT(type, FIELDS) \
    typedef struct { FIELDS(HBS_DECL_F, HBS_DECL_A, HBS_DECL_A2, HBS_DECL_S, HBS_DECL_SA) } type;

/* E.2.3 sub-layer HRD parameters (reference hevc_stream.h:41-48) */
#define HBS_SUB_LAYER_HRD_FIELDS(F, A, A2, S, SA) \
    A(bit_rate_value_minus1, MAX_CPB_CNT) \
    A(cpb_size_value_minus1, MAX_CPB_CNT) \
    A(cpb_size_du_value_minus1, MAX_CPB_CNT) \
    A(bit_rate_du_value_minus1, MAX_CPB_CNT) \
    A(cbr_flag, MAX_CPB_CNT)
HBS_STRUCT(hevc_sub_layer_hrd_t, HBS_SUB_LAYER_HRD_FIELDS)

/* E.2.2 HRD parameters (reference hevc_stream.h:53-75) */
#define HBS_HRD_FIELDS(F, A, A2, S, SA) \
    F(nal_hrd_parameters_present_flag) \
    F(vcl_hrd_parameters_present_flag) \
    F(sub_pic_hrd_params_present_flag) \
    F(tick_divisor_minus2) \
    F(du_cpb_removal_delay_increment_length_minus1) \
    F(sub_pic_cpb_params_in_pic_timing_sei_flag) \
    F(dpb_output_delay_du_length_minus1) \
    F(bit_rate_scale) \
    F(cpb_size_scale) \
    F(cpb_size_du_scale) \
    F(initial_cpb_removal_delay_length_minus1) \
    F(au_cpb_removal_delay_length_minus1) \
    F(dpb_output_delay_length_minus1) \
    A(fixed_pic_rate_general_flag, MAX_NUM_SUBLAYERS) \
    A(fixed_pic_rate_within_cvs_flag, MAX_NUM_SUBLAYERS) \
    A(elemental_duration_in_tc_minus1, MAX_NUM_SUBLAYERS) \
    A(low_delay_hrd_flag, MAX_NUM_SUBLAYERS) \
    A(cpb_cnt_minus1, MAX_NUM_SUBLAYERS) \
    SA(hevc_sub_layer_hrd_t, sub_layer_hrd_nal, MAX_NUM_SUBLAYERS) \
    SA(hevc_sub_layer_hrd_t, sub_layer_hrd_vcl, MAX_NUM_SUBLAYERS)
HBS_STRUCT(hevc_hrd_t, HBS_HRD_FIELDS)

/* 7.3.3 profile, tier and level (reference hevc_stream.h:81-127) */
#define HBS_PTL_FIELDS(F, A, A2, S, SA) \
    F(general_profile_space) \
    F(general_tier_flag) \
    F(general_profile_idc) \
    A(general_profile_compatibility_flag, 32) \
    F(general_progressive_source_flag) \
    F(general_interlaced_source_flag) \
    F(general_non_packed_constraint_flag) \
    F(general_frame_only_constraint_flag) \
    F(general_max_12bit_constraint_flag) \
    F(general_max_10bit_constraint_flag) \
    F(general_max_8bit_constraint_flag) \
    F(general_max_422chroma_constraint_flag) \
    F(general_max_420chroma_constraint_flag) \
    F(general_max_monochrome_constraint_flag) \
    F(general_intra_constraint_flag) \
    F(general_one_picture_only_constraint_flag) \
    F(general_lower_bit_rate_constraint_flag) \
    F(general_max_14bit_constraint_flag) \
    F(general_inbld_flag) \
    F(general_level_idc) \
    A(sub_layer_profile_present_flag, MAX_NUM_SUBLAYERS) \
    A(sub_layer_level_present_flag, MAX_NUM_SUBLAYERS) \
    A(sub_layer_profile_space, MAX_NUM_SUBLAYERS) \
    A(sub_layer_tier_flag, MAX_NUM_SUBLAYERS) \
    A(sub_layer_profile_idc, MAX_NUM_SUBLAYERS) \
    A2(sub_layer_profile_compatibility_flag, MAX_NUM_SUBLAYERS, 32) \
    A(sub_layer_progressive_source_flag, MAX_NUM_SUBLAYERS) \
    A(sub_layer_interlaced_source_flag, MAX_NUM_SUBLAYERS) \
    A(sub_layer_non_packed_constraint_flag, MAX_NUM_SUBLAYERS) \
    A(sub_layer_frame_only_constraint_flag, MAX_NUM_SUBLAYERS) \
    A(sub_layer_max_12bit_constraint_flag, MAX_NUM_SUBLAYERS) \
    A(sub_layer_max_10bit_constraint_flag, MAX_NUM_SUBLAYERS) \
    A(sub_layer_max_8bit_constraint_flag, MAX_NUM_SUBLAYERS) \
    A(sub_layer_max_422chroma_constraint_flag, MAX_NUM_SUBLAYERS) \
    A(sub_layer_max_420chroma_constraint_flag, MAX_NUM_SUBLAYERS) \
    A(sub_layer_max_monochrome_constraint_flag, MAX_NUM_SUBLAYERS) \
    A(sub_layer_intra_constraint_flag, MAX_NUM_SUBLAYERS) \
    A(sub_layer_one_picture_only_constraint_flag, MAX_NUM_SUBLAYERS) \
    A(sub_layer_lower_bit_rate_constraint_flag, MAX_NUM_SUBLAYERS) \
    A(sub_layer_max_14bit_constraint_flag, MAX_NUM_SUBLAYERS) \
    A(sub_layer_inbld_flag, MAX_NUM_SUBLAYERS) \
    A(sub_layer_level_idc, MAX_NUM_SUBLAYERS)
HBS_STRUCT(hevc_profile_tier_level_t, HBS_PTL_FIELDS)

/* 7.3.4 scaling list data (reference hevc_stream.h:133-139) */
#define HBS_SCALING_LIST_FIELDS(F, A, A2, S, SA) \
    A2(scaling_list_pred_mode_flag, 4, 6) \
    A2(scaling_list_pred_matrix_id_delta, 4, 6) \
    A2(scaling_list_dc_coef_minus8, 2, 6) \
    A2(scaling_list_delta_coef, 4, 64)
HBS_STRUCT(hevc_scaling_list_data_t, HBS_SCALING_LIST_FIELDS)

/* 7.3.2.1 video parameter set (reference hevc_stream.h:148-177) */
#define HBS_VPS_FIELDS(F, A, A2, S, SA) \
    F(vps_video_parameter_set_id) \
    F(vps_base_layer_internal_flag) \
    F(vps_base_layer_available_flag) \
    F(vps_max_layers_minus1) \
    F(vps_max_sub_layers_minus1) \
    F(vps_temporal_id_nesting_flag) \
    S(hevc_profile_tier_level_t, ptl) \
    F(vps_sub_layer_ordering_info_present_flag) \
    A(vps_max_dec_pic_buffering_minus1, MAX_NUM_SUBLAYERS) \
    A(vps_max_num_reorder_pics, MAX_NUM_SUBLAYERS) \
    A(vps_max_latency_increase_plus1, MAX_NUM_SUBLAYERS) \
    F(vps_max_layer_id) \
    F(vps_num_layer_sets_minus1) \
    A2(layer_id_included_flag, MAX_NUM_SUBLAYERS, MAX_NUM_SUBLAYERS) \
    F(vps_timing_info_present_flag) \
    F(vps_num_units_in_tick) \
    F(vps_time_scale) \
    F(vps_poc_proportional_to_timing_flag) \
    F(vps_num_ticks_poc_diff_one_minus1) \
    F(vps_num_hrd_parameters) \
    A(hrd_layer_set_idx, MAX_NUM_HRD_PARAM) \
    A(cprms_present_flag, MAX_NUM_HRD_PARAM) \
    SA(hevc_hrd_t, hrd, MAX_NUM_HRD_PARAM) \
    F(vps_extension_flag) \
    F(vps_extension_data_flag)
HBS_STRUCT(hevc_vps_t, HBS_VPS_FIELDS)

/* 7.3.7 short-term reference picture set (reference hevc_stream.h:183-197) */
#define HBS_ST_RPS_FIELDS(F, A, A2, S, SA) \
    F(inter_ref_pic_set_prediction_flag) \
    F(delta_idx_minus1) \
    F(delta_rps_sign) \
    F(abs_delta_rps_minus1) \
    A(used_by_curr_pic_flag, MAX_NUM_SHORT_TERM_REF_PICS) \
    A(use_delta_flag, MAX_NUM_SHORT_TERM_REF_PICS) \
    F(num_negative_pics) \
    F(num_positive_pics) \
    A(delta_poc_s0_minus1, MAX_NUM_NEGATIVE_PICS) \
    A(used_by_curr_pic_s0_flag, MAX_NUM_NEGATIVE_PICS) \
    A(delta_poc_s1_minus1, MAX_NUM_POSITIVE_PICS) \
    A(used_by_curr_pic_s1_flag, MAX_NUM_NEGATIVE_PICS)
HBS_STRUCT(hevc_st_ref_pic_set_t, HBS_ST_RPS_FIELDS)

/* E.2.1 VUI parameters (reference hevc_stream.h:203-245) */
#define HBS_VUI_FIELDS(F, A, A2, S, SA) \
    F(aspect_ratio_info_present_flag) \
    F(aspect_ratio_idc) \
    F(sar_width) \
    F(sar_height) \
    F(overscan_info_present_flag) \
    F(overscan_appropriate_flag) \
    F(video_signal_type_present_flag) \
    F(video_format) \
    F(video_full_range_flag) \
    F(colour_description_present_flag) \
    F(colour_primaries) \
    F(transfer_characteristics) \
    F(matrix_coefficients) \
    F(chroma_loc_info_present_flag) \
    F(chroma_sample_loc_type_top_field) \
    F(chroma_sample_loc_type_bottom_field) \
    F(neutral_chroma_indication_flag) \
    F(field_seq_flag) \
    F(frame_field_info_present_flag) \
    F(default_display_window_flag) \
    F(def_disp_win_left_offset) \
    F(def_disp_win_right_offset) \
    F(def_disp_win_top_offset) \
    F(def_disp_win_bottom_offset) \
    F(vui_timing_info_present_flag) \
    F(vui_num_units_in_tick) \
    F(vui_time_scale) \
    F(vui_poc_proportional_to_timing_flag) \
    F(vui_num_ticks_poc_diff_one_minus1) \
    F(vui_hrd_parameters_present_flag) \
    S(hevc_hrd_t, hrd) \
    F(bitstream_restriction_flag) \
    F(tiles_fixed_structure_flag) \
    F(motion_vectors_over_pic_boundaries_flag) \
    F(restricted_ref_pic_lists_flag) \
    F(min_spatial_segmentation_idc) \
    F(max_bytes_per_pic_denom) \
    F(max_bits_per_min_cu_denom) \
    F(log2_max_mv_length_horizontal) \
    F(log2_max_mv_length_vertical)
HBS_STRUCT(hevc_vui_t, HBS_VUI_FIELDS)

/* 7.3.2.2.2 SPS range extension (reference hevc_stream.h:251-262) */
#define HBS_SPS_RANGE_EXT_FIELDS(F, A, A2, S, SA) \
    F(transform_skip_rotation_enabled_flag) \
    F(transform_skip_context_enabled_flag) \
    F(implicit_rdpcm_enabled_flag) \
    F(explicit_rdpcm_enabled_flag) \
    F(extended_precision_processing_flag) \
    F(intra_smoothing_disabled_flag) \
    F(high_precision_offsets_enabled_flag) \
    F(persistent_rice_adaptation_enabled_flag) \
    F(cabac_bypass_alignment_enabled_flag)
HBS_STRUCT(hevc_sps_range_ext_t, HBS_SPS_RANGE_EXT_FIELDS)

/* 7.3.2.2.3 SPS screen content coding extension: declared, never parsed
 * (reference hevc_stream.h:268-279) */
#define HBS_SPS_SCC_EXT_FIELDS(F, A, A2, S, SA) \
    F(sps_curr_pic_ref_enabled_flag) \
    F(palette_mode_enabled_flag) \
    F(palette_max_size) \
    F(delta_palette_max_predictor_size) \
    F(sps_palette_predictor_initializer_present_flag) \
    F(sps_num_palette_predictor_initializer_minus1) \
    A2(sps_palette_predictor_initializers, 3, MAX_NUM_PALLETTE_PREDICTOR) \
    F(motion_vector_resolution_control_idc) \
    F(intra_boundary_filtering_disabled_flag)
HBS_STRUCT(hevc_sps_scc_ext_t, HBS_SPS_SCC_EXT_FIELDS)

/* 7.3.2.2 sequence parameter set (reference hevc_stream.h:288-346) */
#define HBS_SPS_FIELDS(F, A, A2, S, SA) \
    F(sps_video_parameter_set_id) \
    F(sps_max_sub_layers_minus1) \
    F(sps_temporal_id_nesting_flag) \
    S(hevc_profile_tier_level_t, ptl) \
    F(sps_seq_parameter_set_id) \
    F(chroma_format_idc) \
    F(separate_colour_plane_flag) \
    F(pic_width_in_luma_samples) \
    F(pic_height_in_luma_samples) \
    F(conformance_window_flag) \
    F(conf_win_left_offset) \
    F(conf_win_right_offset) \
    F(conf_win_top_offset) \
    F(conf_win_bottom_offset) \
    F(bit_depth_luma_minus8) \
    F(bit_depth_chroma_minus8) \
    F(log2_max_pic_order_cnt_lsb_minus4) \
    F(sps_sub_layer_ordering_info_present_flag) \
    A(sps_max_dec_pic_buffering_minus1, MAX_NUM_SUBLAYERS) \
    A(sps_max_num_reorder_pics, MAX_NUM_SUBLAYERS) \
    A(sps_max_latency_increase_plus1, MAX_NUM_SUBLAYERS) \
    F(log2_min_luma_coding_block_size_minus3) \
    F(log2_diff_max_min_luma_coding_block_size) \
    F(log2_min_luma_transform_block_size_minus2) \
    F(log2_diff_max_min_luma_transform_block_size) \
    F(max_transform_hierarchy_depth_inter) \
    F(max_transform_hierarchy_depth_intra) \
    F(scaling_list_enabled_flag) \
    F(sps_scaling_list_data_present_flag) \
    S(hevc_scaling_list_data_t, scaling_list_data) \
    F(amp_enabled_flag) \
    F(sample_adaptive_offset_enabled_flag) \
    F(pcm_enabled_flag) \
    F(pcm_sample_bit_depth_luma_minus1) \
    F(pcm_sample_bit_depth_chroma_minus1) \
    F(log2_min_pcm_luma_coding_block_size_minus3) \
    F(log2_diff_max_min_pcm_luma_coding_block_size) \
    F(pcm_loop_filter_disabled_flag) \
    F(num_short_term_ref_pic_sets) \
    SA(hevc_st_ref_pic_set_t, st_ref_pic_set, MAX_NUM_SHORT_TERM_REF_PICS) \
    F(long_term_ref_pics_present_flag) \
    F(num_long_term_ref_pics_sps) \
    A(lt_ref_pic_poc_lsb_sps, MAX_NUM_LONG_TERM_REF_PICS) \
    A(used_by_curr_pic_lt_sps_flag, MAX_NUM_LONG_TERM_REF_PICS) \
    F(sps_temporal_mvp_enabled_flag) \
    F(strong_intra_smoothing_enabled_flag) \
    F(vui_parameters_present_flag) \
    S(hevc_vui_t, vui) \
    F(sps_extension_present_flag) \
    F(sps_range_extension_flag) \
    F(sps_multilayer_extension_flag) \
    F(sps_3d_extension_flag) \
    F(sps_extension_5bits) \
    S(hevc_sps_range_ext_t, sps_range_ext)
HBS_STRUCT(hevc_sps_t, HBS_SPS_FIELDS)

/* 7.3.2.3.2 PPS range extension (reference hevc_stream.h:352-363) */
#define HBS_PPS_RANGE_EXT_FIELDS(F, A, A2, S, SA) \
    F(log2_max_transform_skip_block_size_minus2) \
    F(cross_component_prediction_enabled_flag) \
    F(chroma_qp_offset_list_enabled_flag) \
    F(diff_cu_chroma_qp_offset_depth) \
    F(chroma_qp_offset_list_len_minus1) \
    A(cb_qp_offset_list, MAX_NUM_CHROMA_QP_OFFSET_LST) \
    A(cr_qp_offset_list, MAX_NUM_CHROMA_QP_OFFSET_LST) \
    F(log2_sao_offset_scale_luma) \
    F(log2_sao_offset_scale_chroma)
HBS_STRUCT(hevc_pps_range_ext_t, HBS_PPS_RANGE_EXT_FIELDS)

/* 7.3.2.3 picture parameter set (reference hevc_stream.h:372-421) */
#define HBS_PPS_FIELDS(F, A, A2, S, SA) \
    F(pic_parameter_set_id) \
    F(seq_parameter_set_id) \
    F(dependent_slice_segments_enabled_flag) \
    F(output_flag_present_flag) \
    F(num_extra_slice_header_bits) \
    F(sign_data_hiding_enabled_flag) \
    F(cabac_init_present_flag) \
    F(num_ref_idx_l0_default_active_minus1) \
    F(num_ref_idx_l1_default_active_minus1) \
    F(init_qp_minus26) \
    F(constrained_intra_pred_flag) \
    F(transform_skip_enabled_flag) \
    F(cu_qp_delta_enabled_flag) \
    F(diff_cu_qp_delta_depth) \
    F(pps_cb_qp_offset) \
    F(pps_cr_qp_offset) \
    F(pps_slice_chroma_qp_offsets_present_flag) \
    F(weighted_pred_flag) \
    F(weighted_bipred_flag) \
    F(transquant_bypass_enabled_flag) \
    F(tiles_enabled_flag) \
    F(entropy_coding_sync_enabled_flag) \
    F(num_tile_columns_minus1) \
    F(num_tile_rows_minus1) \
    F(uniform_spacing_flag) \
    A(column_width_minus1, MAX_NUM_TILE_COLUMN) \
    A(row_height_minus1, MAX_NUM_TILE_ROW) \
    F(loop_filter_across_tiles_enabled_flag) \
    F(pps_loop_filter_across_slices_enabled_flag) \
    F(deblocking_filter_control_present_flag) \
    F(deblocking_filter_override_enabled_flag) \
    F(pps_deblocking_filter_disabled_flag) \
    F(pps_beta_offset_div2) \
    F(pps_tc_offset_div2) \
    F(pps_scaling_list_data_present_flag) \
    S(hevc_scaling_list_data_t, scaling_list_data) \
    F(lists_modification_present_flag) \
    F(log2_parallel_merge_level_minus2) \
    F(slice_segment_header_extension_present_flag) \
    F(pps_extension_present_flag) \
    F(pps_range_extension_flag) \
    F(pps_multilayer_extension_flag) \
    F(pps_3d_extension_flag) \
    F(pps_extension_5bits) \
    S(hevc_pps_range_ext_t, pps_range_ext)
HBS_STRUCT(hevc_pps_t, HBS_PPS_FIELDS)

/* 7.3.6.2 reference picture list modification (reference hevc_stream.h:427-433) */
#define HBS_RPLM_FIELDS(F, A, A2, S, SA) \
    F(ref_pic_list_modification_flag_l0) \
    A(list_entry_l0, MAX_NUM_REF_PICS_L0) \
    F(ref_pic_list_modification_flag_l1) \
    A(list_entry_l1, MAX_NUM_REF_PICS_L1)
HBS_STRUCT(hevc_ref_pics_lists_mod_t, HBS_RPLM_FIELDS)

/* 7.3.6.3 weighted prediction parameters (reference hevc_stream.h:439-456) */
#define HBS_PWT_FIELDS(F, A, A2, S, SA) \
    F(luma_log2_weight_denom) \
    F(delta_chroma_log2_weight_denom) \
    A(luma_weight_l0_flag, MAX_NUM_REF_PICS_L0) \
    A(chroma_weight_l0_flag, MAX_NUM_REF_PICS_L0) \
    A(delta_luma_weight_l0, MAX_NUM_REF_PICS_L0) \
    A(luma_offset_l0, MAX_NUM_REF_PICS_L0) \
    A2(delta_chroma_weight_l0, MAX_NUM_REF_PICS_L0, 2) \
    A2(delta_chroma_offset_l0, MAX_NUM_REF_PICS_L0, 2) \
    A(luma_weight_l1_flag, MAX_NUM_REF_PICS_L1) \
    A(chroma_weight_l1_flag, MAX_NUM_REF_PICS_L1) \
    A(delta_luma_weight_l1, MAX_NUM_REF_PICS_L1) \
    A(luma_offset_l1, MAX_NUM_REF_PICS_L1) \
    A2(delta_chroma_weight_l1, MAX_NUM_REF_PICS_L1, 2) \
    A2(delta_chroma_offset_l1, MAX_NUM_REF_PICS_L1, 2)
HBS_STRUCT(hevc_pred_weight_table_t, HBS_PWT_FIELDS)

/* 7.3.6 slice segment header (reference hevc_stream.h:465-515) */
#define HBS_SLICE_HEADER_FIELDS(F, A, A2, S, SA) \
    F(first_slice_segment_in_pic_flag) \
    F(no_output_of_prior_pics_flag) \
    F(pic_parameter_set_id) \
    F(dependent_slice_segment_flag) \
    F(slice_segment_address) \
    F(slice_type) \
    F(pic_output_flag) \
    F(colour_plane_id) \
    F(slice_pic_order_cnt_lsb) \
    F(short_term_ref_pic_set_sps_flag) \
    S(hevc_st_ref_pic_set_t, st_ref_pic_set) \
    F(short_term_ref_pic_set_idx) \
    F(num_long_term_sps) \
    F(num_long_term_pics) \
    A(lt_idx_sps, MAX_NUM_LONG_TERM_REF_PICS) \
    A(poc_lsb_lt, MAX_NUM_LONG_TERM_REF_PICS) \
    A(used_by_curr_pic_lt_flag, MAX_NUM_LONG_TERM_REF_PICS) \
    A(delta_poc_msb_present_flag, MAX_NUM_LONG_TERM_REF_PICS) \
    A(delta_poc_msb_cycle_lt, MAX_NUM_LONG_TERM_REF_PICS) \
    F(slice_temporal_mvp_enabled_flag) \
    F(slice_sao_luma_flag) \
    F(slice_sao_chroma_flag) \
    F(num_ref_idx_active_override_flag) \
    F(num_ref_idx_l0_active_minus1) \
    F(num_ref_idx_l1_active_minus1) \
    S(hevc_ref_pics_lists_mod_t, rpld) \
    F(mvd_l1_zero_flag) \
    F(cabac_init_flag) \
    F(collocated_from_l0_flag) \
    F(collocated_ref_idx) \
    S(hevc_pred_weight_table_t, pwt) \
    F(five_minus_max_num_merge_cand) \
    F(slice_qp_delta) \
    F(slice_cb_qp_offset) \
    F(slice_cr_qp_offset) \
    F(cu_chroma_qp_offset_enabled_flag) \
    F(deblocking_filter_override_flag) \
    F(slice_deblocking_filter_disabled_flag) \
    F(slice_beta_offset_div2) \
    F(slice_tc_offset_div2) \
    F(slice_loop_filter_across_slices_enabled_flag) \
    F(num_entry_point_offsets) \
    F(offset_len_minus1) \
    A(entry_point_offset_minus1, MAX_NUM_ENTRY_POINT_OFFSET) \
    F(slice_segment_header_extension_length)
HBS_STRUCT(hevc_slice_header_t, HBS_SLICE_HEADER_FIELDS)

/* 7.3.1 NAL unit header (reference hevc_stream.h:524-530) */
#define HBS_NAL_FIELDS(F, A, A2, S, SA) \
    F(forbidden_zero_bit) \
    F(nal_unit_type) \
    F(nal_layer_id) \
    F(nal_temporal_id_plus1)
HBS_STRUCT(hevc_nal_t, HBS_NAL_FIELDS)

/* slice payload handed back by the slice reader (reference hevc_stream.h:532-536) */
typedef struct
{
    int rbsp_size;
    uint8_t* rbsp_buf;
} hevc_slice_data_rbsp_t;

/* 7.3.5 access unit delimiter (reference hevc_stream.h:544-547) */
#define HBS_AUD_FIELDS(F, A, A2, S, SA) \
    F(primary_pic_type)
HBS_STRUCT(hevc_aud_t, HBS_AUD_FIELDS)

/* the parser object (reference hevc_stream.h:556-569): the structures of the
 * NAL that was read last, plus the id-indexed parameter-set tables */
typedef struct
{
    hevc_nal_t* nal;
    hevc_vps_t* vps;
    hevc_sps_t* sps;
    hevc_pps_t* pps;
    hevc_aud_t* aud;
    hevc_slice_header_t* sh;

    hevc_slice_data_rbsp_t* slice_data;

    hevc_sps_t* sps_table[32];
    hevc_pps_t* pps_table[256];
} hevc_stream_t;

/* reference hevc_stream.h:571-574, hevc_nal.c:34-114 */
hevc_stream_t* hevc_new();
void hevc_free(hevc_stream_t* h);
int read_debug_hevc_nal_unit(hevc_stream_t* h, uint8_t* buf, int size);
/* exported by the reference without a prototype (hevc_stream.c:155, :1249; hevc_nal.c:97) */
int read_hevc_nal_unit(hevc_stream_t* h, uint8_t* buf, int size);
int write_hevc_nal_unit(hevc_stream_t* h, uint8_t* buf, int size);
int peek_hevc_nal_unit(hevc_stream_t* h, uint8_t* buf, int size);

/* Table 7-1 NAL unit type codes (reference hevc_stream.h:577-619) */
#define HEVC_NAL_UNIT_TYPE_TRAIL_N                    0
#define HEVC_NAL_UNIT_TYPE_TRAIL_R                    1
#define HEVC_NAL_UNIT_TYPE_TSA_N                      2
#define HEVC_NAL_UNIT_TYPE_TSA_R                      3
#define HEVC_NAL_UNIT_TYPE_STSA_N                     4
#define HEVC_NAL_UNIT_TYPE_STSA_R                     5
#define HEVC_NAL_UNIT_TYPE_RADL_N                     6
#define HEVC_NAL_UNIT_TYPE_RADL_R                     7
#define HEVC_NAL_UNIT_TYPE_RASL_N                     8
#define HEVC_NAL_UNIT_TYPE_RASL_R                     9
#define HEVC_NAL_UNIT_TYPE_RSV_VCL_N10               10
#define HEVC_NAL_UNIT_TYPE_RSV_VCL_R11               11
#define HEVC_NAL_UNIT_TYPE_RSV_VCL_N12               12
#define HEVC_NAL_UNIT_TYPE_RSV_VCL_R13               13
#define HEVC_NAL_UNIT_TYPE_RSV_VCL_N14               14
#define HEVC_NAL_UNIT_TYPE_RSV_VCL_R15               15
#define HEVC_NAL_UNIT_TYPE_BLA_W_LP                  16
#define HEVC_NAL_UNIT_TYPE_BLA_W_RADL                17
#define HEVC_NAL_UNIT_TYPE_BLA_N_LP                  18
#define HEVC_NAL_UNIT_TYPE_IDR_W_RADL                19
#define HEVC_NAL_UNIT_TYPE_IDR_N_LP                  20
#define HEVC_NAL_UNIT_TYPE_CRA_NUT                   21
#define HEVC_NAL_UNIT_TYPE_RSV_IRAP_VCL22            22
#define HEVC_NAL_UNIT_TYPE_RSV_IRAP_VCL23            23
#define HEVC_NAL_UNIT_TYPE_RSV_VCL24                 24
#define HEVC_NAL_UNIT_TYPE_RSV_VCL25                 25
#define HEVC_NAL_UNIT_TYPE_RSV_VCL26                 26
#define HEVC_NAL_UNIT_TYPE_RSV_VCL27                 27
#define HEVC_NAL_UNIT_TYPE_RSV_VCL28                 28
#define HEVC_NAL_UNIT_TYPE_RSV_VCL29                 29
#define HEVC_NAL_UNIT_TYPE_RSV_VCL30                 30
#define HEVC_NAL_UNIT_TYPE_RSV_VCL31                 31
#define HEVC_NAL_UNIT_TYPE_VPS_NUT                   32
#define HEVC_NAL_UNIT_TYPE_SPS_NUT                   33
#define HEVC_NAL_UNIT_TYPE_PPS_NUT                   34
#define HEVC_NAL_UNIT_TYPE_AUD_NUT                   35
#define HEVC_NAL_UNIT_TYPE_EOS_NUT                   36
#define HEVC_NAL_UNIT_TYPE_EOB_NUT                   37
#define HEVC_NAL_UNIT_TYPE_FD_NUT                    38
#define HEVC_NAL_UNIT_TYPE_PREFIX_SEI_NUT            39
#define HEVC_NAL_UNIT_TYPE_SUFFIX_SEI_NUT            40
#define MAX_HEVC_VAL_UNIT_TYPE                       40

/* Table 7-7 slice_type values (reference hevc_stream.h:625-627) */
#define HEVC_SLICE_TYPE_B        0
#define HEVC_SLICE_TYPE_P        1
#define HEVC_SLICE_TYPE_I        2

#define HEVC_PROFILE_BASELINE  66
#define HEVC_PROFILE_MAIN      77
#define HEVC_PROFILE_EXTENDED  88
#define HEVC_PROFILE_HIGH     100

/* destination of the "!! Found NAL" / debug_bytes lines (reference h264_stream.c:33) */
extern FILE* h264_dbgfile;

#ifdef __cplusplus
}
#endif

#endif
