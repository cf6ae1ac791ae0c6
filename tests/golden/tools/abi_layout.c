#include <stdio.h>
#include <stddef.h>
#include "hevc_stream.h"
#define SZ(t) printf("\"%s\": %zu,\n", #t, sizeof(t))
#define OFF(t,f) printf("\"%s.%s\": %zu,\n", #t, #f, offsetof(t,f))
int main(){
 printf("{\n");
 SZ(bs_t); SZ(hevc_nal_t); SZ(hevc_sub_layer_hrd_t); SZ(hevc_hrd_t); SZ(hevc_profile_tier_level_t); SZ(hevc_scaling_list_data_t);
 SZ(hevc_vps_t); SZ(hevc_st_ref_pic_set_t); SZ(hevc_vui_t); SZ(hevc_sps_range_ext_t); SZ(hevc_sps_scc_ext_t); SZ(hevc_sps_t);
 SZ(hevc_pps_range_ext_t); SZ(hevc_pps_t); SZ(hevc_ref_pics_lists_mod_t); SZ(hevc_pred_weight_table_t); SZ(hevc_slice_header_t);
 SZ(hevc_slice_data_rbsp_t); SZ(hevc_aud_t); SZ(hevc_stream_t); SZ(sei_t);
 OFF(hevc_vps_t, ptl); OFF(hevc_vps_t, vps_sub_layer_ordering_info_present_flag); OFF(hevc_vps_t, layer_id_included_flag); OFF(hevc_vps_t, hrd); OFF(hevc_vps_t, vps_extension_data_flag);
 OFF(hevc_hrd_t, sub_layer_hrd_nal); OFF(hevc_hrd_t, sub_layer_hrd_vcl); OFF(hevc_hrd_t, cpb_cnt_minus1);
 OFF(hevc_profile_tier_level_t, general_level_idc); OFF(hevc_profile_tier_level_t, sub_layer_profile_compatibility_flag); OFF(hevc_profile_tier_level_t, sub_layer_level_idc);
 OFF(hevc_sps_t, ptl); OFF(hevc_sps_t, sps_seq_parameter_set_id); OFF(hevc_sps_t, scaling_list_data); OFF(hevc_sps_t, st_ref_pic_set); OFF(hevc_sps_t, vui); OFF(hevc_sps_t, sps_range_ext); OFF(hevc_sps_t, num_short_term_ref_pic_sets);
 OFF(hevc_vui_t, hrd); OFF(hevc_vui_t, log2_max_mv_length_vertical);
 OFF(hevc_pps_t, scaling_list_data); OFF(hevc_pps_t, pps_range_ext); OFF(hevc_pps_t, column_width_minus1); OFF(hevc_pps_t, pps_extension_5bits);
 OFF(hevc_slice_header_t, st_ref_pic_set); OFF(hevc_slice_header_t, rpld); OFF(hevc_slice_header_t, pwt); OFF(hevc_slice_header_t, collocated_from_l0_flag); OFF(hevc_slice_header_t, entry_point_offset_minus1); OFF(hevc_slice_header_t, slice_segment_header_extension_length);
 OFF(hevc_st_ref_pic_set_t, num_negative_pics); OFF(hevc_st_ref_pic_set_t, used_by_curr_pic_s1_flag);
 OFF(hevc_pred_weight_table_t, luma_weight_l1_flag); OFF(hevc_pred_weight_table_t, delta_chroma_offset_l1);
 OFF(hevc_stream_t, slice_data); OFF(hevc_stream_t, sps_table); OFF(hevc_stream_t, pps_table);
 printf("\"_end\": 0\n}\n"); return 0; }
