/* Prints the field layout (name, byte offset, dims, nested type) of every
 * struct of include/hevc_stream.h as JSON, from its field tables.
 * gcc -Iinclude tests/golden/tools/field_layout.c && ./a.out > tests/golden/field_layout.json */
#include <stdio.h>
#include <stddef.h>
#include "hevc_stream.h"

#define P_F(name)           printf("   [\"%s\", %zu, [], null],\n", #name, offsetof(T, name));
#define P_A(name, n)        printf("   [\"%s\", %zu, [%d], null],\n", #name, offsetof(T, name), (int)(n));
#define P_A2(name, n, m)    printf("   [\"%s\", %zu, [%d, %d], null],\n", #name, offsetof(T, name), (int)(n), (int)(m));
#define P_S(type, name)     printf("   [\"%s\", %zu, [], \"%s\"],\n", #name, offsetof(T, name), #type);
#define P_SA(type, name, n) printf("   [\"%s\", %zu, [%d], \"%s\"],\n", #name, offsetof(T, name), (int)(n), #type);
#define DUMP(type, FIELDS) { printf(" \"%s\": {\"size\": %zu, \"fields\": [\n", #type, sizeof(type)); FIELDS(P_F, P_A, P_A2, P_S, P_SA) printf("   null]},\n"); }

int main(void)
{
    printf("{\n");
#define T hevc_sub_layer_hrd_t
    DUMP(hevc_sub_layer_hrd_t, HBS_SUB_LAYER_HRD_FIELDS)
#undef T
#define T hevc_hrd_t
    DUMP(hevc_hrd_t, HBS_HRD_FIELDS)
#undef T
#define T hevc_profile_tier_level_t
    DUMP(hevc_profile_tier_level_t, HBS_PTL_FIELDS)
#undef T
#define T hevc_scaling_list_data_t
    DUMP(hevc_scaling_list_data_t, HBS_SCALING_LIST_FIELDS)
#undef T
#define T hevc_vps_t
    DUMP(hevc_vps_t, HBS_VPS_FIELDS)
#undef T
#define T hevc_st_ref_pic_set_t
    DUMP(hevc_st_ref_pic_set_t, HBS_ST_RPS_FIELDS)
#undef T
#define T hevc_vui_t
    DUMP(hevc_vui_t, HBS_VUI_FIELDS)
#undef T
#define T hevc_sps_range_ext_t
    DUMP(hevc_sps_range_ext_t, HBS_SPS_RANGE_EXT_FIELDS)
#undef T
#define T hevc_sps_t
    DUMP(hevc_sps_t, HBS_SPS_FIELDS)
#undef T
#define T hevc_pps_range_ext_t
    DUMP(hevc_pps_range_ext_t, HBS_PPS_RANGE_EXT_FIELDS)
#undef T
#define T hevc_pps_t
    DUMP(hevc_pps_t, HBS_PPS_FIELDS)
#undef T
#define T hevc_ref_pics_lists_mod_t
    DUMP(hevc_ref_pics_lists_mod_t, HBS_RPLM_FIELDS)
#undef T
#define T hevc_pred_weight_table_t
    DUMP(hevc_pred_weight_table_t, HBS_PWT_FIELDS)
#undef T
#define T hevc_slice_header_t
    DUMP(hevc_slice_header_t, HBS_SLICE_HEADER_FIELDS)
#undef T
#define T hevc_nal_t
    DUMP(hevc_nal_t, HBS_NAL_FIELDS)
#undef T
    printf(" \"_end\": null\n}\n");
    return 0;
}
