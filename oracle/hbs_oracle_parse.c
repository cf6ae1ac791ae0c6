/*
 * hbs_oracle_parse.c -- TEST INFRASTRUCTURE ONLY (see hbs_oracle.h).
 *
 * CPU restatement of the read direction of the reference's HEVC header layer:
 * read_hevc_nal_unit (hevc_stream.c:155-240) and the syntax readers it
 * dispatches to (:243-1218), including their departures from H.265 that a
 * drop-in must reproduce.  Field by field it does what the
 * reference does; the structs are the ABI-identical ones of
 * include/hevc_stream.h.
 *
 * Where the reference has undefined behaviour this file is bounded instead and
 * the generators stay inside the defined envelope:
 *   - active PPS/SPS: the reference indexes the single h->pps / h->sps objects
 *     with the ids (:800-801); only id 0 is defined (= last parsed PPS/SPS).
 *     Other ids select an all-zero parameter set here.
 *   - the derived RPS tables (:26-32) have 32 rows; rows outside 0..31 read as
 *     zero and are not written.
 *   - ceil(log2(n)) for n <= 0 gives 0 bits (x86 result of the reference).
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include "hbs_oracle.h"
#include "hbs_oracle_bits.h"
#include "../include/hevc_stream.h"
#include "../include/h264_stream.h"

#define ROWS 32

struct orc_hevc {
    hevc_stream_t h;
    /* hevc_stream.c:26-32 (file-static there: one parser at a time) */
    int NumDeltaPocs[ROWS], NumNegativePics[ROWS], NumPositivePics[ROWS];
    int DeltaPocS0[ROWS][32], UsedByCurrPicS0[ROWS][32];
    int DeltaPocS1[ROWS][32], UsedByCurrPicS1[ROWS][32];
    /* per-call scratch */
    uint8_t* rbsp;
    int rbsp_cap;
    int rbsp_size;
    int slice_data_off;          /* offset of the slice payload copy in rbsp, or -1 */
    hevc_pps_t zero_pps;
    hevc_sps_t zero_sps;
};

orc_hevc* orc_hevc_new(void)
{
    int i;
    orc_hevc* o = (orc_hevc*)calloc(1, sizeof(orc_hevc));
    /* hevc_nal.c:34-57 */
    o->h.nal = (hevc_nal_t*)calloc(1, sizeof(hevc_nal_t));
    for (i = 0; i < 32; i++) o->h.sps_table[i] = (hevc_sps_t*)calloc(1, sizeof(hevc_sps_t));
    for (i = 0; i < 256; i++) o->h.pps_table[i] = (hevc_pps_t*)calloc(1, sizeof(hevc_pps_t));
    o->h.vps = (hevc_vps_t*)calloc(1, sizeof(hevc_vps_t));
    o->h.sps = (hevc_sps_t*)calloc(1, sizeof(hevc_sps_t));
    o->h.pps = (hevc_pps_t*)calloc(1, sizeof(hevc_pps_t));
    o->h.aud = (hevc_aud_t*)calloc(1, sizeof(hevc_aud_t));
    o->h.sh = (hevc_slice_header_t*)calloc(1, sizeof(hevc_slice_header_t));
    o->h.slice_data = (hevc_slice_data_rbsp_t*)calloc(1, sizeof(hevc_slice_data_rbsp_t));
    o->slice_data_off = -1;
    return o;
}

void orc_hevc_free(orc_hevc* o)
{
    int i;
    if (!o) return;
    free(o->h.nal);
    for (i = 0; i < 32; i++) free(o->h.sps_table[i]);
    for (i = 0; i < 256; i++) free(o->h.pps_table[i]);
    free(o->h.slice_data); free(o->h.sh); free(o->h.aud); free(o->h.pps); free(o->h.sps); free(o->h.vps);
    free(o->rbsp);
    free(o);
}

void* orc_hevc_stream_ptr(orc_hevc* o) { return &o->h; }
const uint8_t* orc_hevc_rbsp(orc_hevc* o, int* size) { *size = o->rbsp_size; return o->rbsp; }
int orc_hevc_slice_data_off(orc_hevc* o) { return o->slice_data_off; }
/* the derived tables as they stand (hevc_stream.c:26-32): 3 x 32 counts, then 4 x 32 x 32 values, in the order of the struct above */
const int* orc_hevc_tables(orc_hevc* o) { return o->NumDeltaPocs; }

static int row_ok(int r) { return r >= 0 && r < ROWS; }
static int col_ok(int c) { return c >= 0 && c < 32; }

/* x86 result of (int)ceil(log2(n)) as the reference computes it (:122,:832,:842,:954,:962) */
static int ceil_log2_ref(int n)
{
    if (n <= 0) return 0;
    return (int)ceil(log2((double)n));
}

/* hevc_stream.c:115-123 */
static int slice_address_bits(const hevc_sps_t* sps)
{
    int MinCbLog2SizeY = sps->log2_min_luma_coding_block_size_minus3 + 3;
    int CtbLog2SizeY = MinCbLog2SizeY + sps->log2_diff_max_min_luma_coding_block_size;
    int CtbSizeY, PicWidthInCtbsY, PicHeightInCtbsY;
    if (CtbLog2SizeY < 0 || CtbLog2SizeY > 30) return 0;       /* bounded: shift is undefined there */
    CtbSizeY = 1 << CtbLog2SizeY;
    PicWidthInCtbsY = (int)ceil(sps->pic_width_in_luma_samples * 1.0f / CtbSizeY);
    PicHeightInCtbsY = (int)ceil(sps->pic_height_in_luma_samples * 1.0f / CtbSizeY);
    return ceil_log2_ref(PicWidthInCtbsY * PicHeightInCtbsY);
}

/* hevc_stream.c:35-59 */
static int num_pic_total_curr(orc_hevc* o, const hevc_sps_t* sps, const hevc_slice_header_t* sh)
{
    int i, n = 0;
    int CurrRpsIdx = sps->num_short_term_ref_pic_sets;
    if (sh->short_term_ref_pic_set_sps_flag) CurrRpsIdx = sh->short_term_ref_pic_set_idx;
    if (row_ok(CurrRpsIdx)) {
        for (i = 0; i < o->NumNegativePics[CurrRpsIdx] && i < 32; i++)
            if (o->UsedByCurrPicS0[CurrRpsIdx][i]) n++;
        for (i = 0; i < o->NumPositivePics[CurrRpsIdx] && i < 32; i++)
            if (o->UsedByCurrPicS1[CurrRpsIdx][i]) n++;
    }
    for (i = 0; i < sh->num_long_term_sps + sh->num_long_term_pics && i < 32; i++) {
        int used = 0;
        if (i < sh->num_long_term_sps) {
            int k = sh->lt_idx_sps[i];
            used = col_ok(k) ? sps->used_by_curr_pic_lt_sps_flag[k] : 0;
        } else {
            used = sh->used_by_curr_pic_lt_flag[i];
        }
        if (used) n++;
    }
    return n;
}

/* hevc_stream.c:61-113 */
static void update_num_delta_pocs(orc_hevc* o, const hevc_st_ref_pic_set_t* rps, int stRpsIdx)
{
    int RefRpsIdx = stRpsIdx - (rps->delta_idx_minus1 + 1);
    if (!row_ok(stRpsIdx)) return;
    if (rps->inter_ref_pic_set_prediction_flag) {
        int i, j, dPoc;
        int deltaRps = (1 - 2 * rps->delta_rps_sign) * (rps->abs_delta_rps_minus1 + 1);
        int refNeg = row_ok(RefRpsIdx) ? o->NumNegativePics[RefRpsIdx] : 0;
        int refPos = row_ok(RefRpsIdx) ? o->NumPositivePics[RefRpsIdx] : 0;
        int refNum = row_ok(RefRpsIdx) ? o->NumDeltaPocs[RefRpsIdx] : 0;
#define RPS_FLAG(arr, k) (col_ok(k) ? rps->arr[k] : 0)
        i = 0;
        for (j = refPos - 1; j >= 0; j--) {
            if (!col_ok(j)) continue;
            dPoc = o->DeltaPocS1[RefRpsIdx][j] + deltaRps;
            if (dPoc < 0 && RPS_FLAG(use_delta_flag, refNeg + j)) {
                if (col_ok(i)) { o->DeltaPocS0[stRpsIdx][i] = dPoc; o->UsedByCurrPicS0[stRpsIdx][i] = RPS_FLAG(used_by_curr_pic_flag, refNeg + j); }
                i++;
            }
        }
        if (deltaRps < 0 && RPS_FLAG(use_delta_flag, refNum)) {
            if (col_ok(i)) { o->DeltaPocS0[stRpsIdx][i] = deltaRps; o->UsedByCurrPicS0[stRpsIdx][i] = RPS_FLAG(used_by_curr_pic_flag, refNum); }
            i++;
        }
        for (j = 0; j < refNeg; j++) {
            if (!col_ok(j)) continue;
            dPoc = o->DeltaPocS0[RefRpsIdx][j] + deltaRps;
            if (dPoc < 0 && RPS_FLAG(use_delta_flag, j)) {
                if (col_ok(i)) { o->DeltaPocS0[stRpsIdx][i] = dPoc; o->UsedByCurrPicS0[stRpsIdx][i] = RPS_FLAG(used_by_curr_pic_flag, j); }
                i++;
            }
        }
        o->NumNegativePics[stRpsIdx] = i;
        i = 0;
        for (j = refNeg - 1; j >= 0; j--) {
            if (!col_ok(j)) continue;
            dPoc = o->DeltaPocS0[RefRpsIdx][j] + deltaRps;
            if (dPoc > 0 && RPS_FLAG(use_delta_flag, j)) {
                if (col_ok(i)) { o->DeltaPocS1[stRpsIdx][i] = dPoc; o->UsedByCurrPicS1[stRpsIdx][i] = RPS_FLAG(used_by_curr_pic_flag, j); }
                i++;
            }
        }
        if (deltaRps > 0 && RPS_FLAG(use_delta_flag, refNum)) {
            if (col_ok(i)) { o->DeltaPocS1[stRpsIdx][i] = deltaRps; o->UsedByCurrPicS1[stRpsIdx][i] = RPS_FLAG(used_by_curr_pic_flag, refNum); }
            i++;
        }
        for (j = 0; j < refPos; j++) {
            if (!col_ok(j)) continue;
            dPoc = o->DeltaPocS1[RefRpsIdx][j] + deltaRps;
            if (dPoc > 0 && RPS_FLAG(use_delta_flag, refNeg + j)) {
                if (col_ok(i)) { o->DeltaPocS1[stRpsIdx][i] = dPoc; o->UsedByCurrPicS1[stRpsIdx][i] = RPS_FLAG(used_by_curr_pic_flag, refNeg + j); }
                i++;
            }
        }
        o->NumPositivePics[stRpsIdx] = i;
#undef RPS_FLAG
    } else {
        o->NumNegativePics[stRpsIdx] = rps->num_negative_pics;
        o->NumPositivePics[stRpsIdx] = rps->num_positive_pics;
    }
    o->NumDeltaPocs[stRpsIdx] = o->NumNegativePics[stRpsIdx] + o->NumPositivePics[stRpsIdx];
}

/* hevc_stream.c:630-649: one bit skipped, then up to the byte boundary; a whole
 * byte when already aligned.  Values are not checked. */
static void trailing_bits(obs_t* b)
{
    obs_skip(b, 1);
    while (!obs_aligned(b)) obs_skip(b, 1);
}

/* hevc_stream.c:652-755 */
static void read_ptl(hevc_profile_tier_level_t* ptl, obs_t* b, int profilePresentFlag, int maxNumSubLayersMinus1)
{
    int i, j;
    if (!profilePresentFlag) return;
    ptl->general_profile_space = obs_u(b, 2);
    ptl->general_tier_flag = obs_u1(b);
    ptl->general_profile_idc = obs_u(b, 5);
    for (i = 0; i < 32; i++) ptl->general_profile_compatibility_flag[i] = obs_u1(b);
    ptl->general_progressive_source_flag = obs_u1(b);
    ptl->general_interlaced_source_flag = obs_u1(b);
    ptl->general_non_packed_constraint_flag = obs_u1(b);
    ptl->general_frame_only_constraint_flag = obs_u1(b);
    if (ptl->general_profile_idc == 4 || ptl->general_profile_compatibility_flag[4] ||
        ptl->general_profile_idc == 5 || ptl->general_profile_compatibility_flag[5] ||
        ptl->general_profile_idc == 6 || ptl->general_profile_compatibility_flag[6] ||
        ptl->general_profile_idc == 7 || ptl->general_profile_compatibility_flag[7]) {
        ptl->general_max_12bit_constraint_flag = obs_u1(b);
        ptl->general_max_10bit_constraint_flag = obs_u1(b);
        ptl->general_max_8bit_constraint_flag = obs_u1(b);
        ptl->general_max_422chroma_constraint_flag = obs_u1(b);
        ptl->general_max_420chroma_constraint_flag = obs_u1(b);
        ptl->general_max_monochrome_constraint_flag = obs_u1(b);
        ptl->general_intra_constraint_flag = obs_u1(b);
        ptl->general_one_picture_only_constraint_flag = obs_u1(b);
        ptl->general_lower_bit_rate_constraint_flag = obs_u1(b);
        obs_skip(b, 34);
    } else {
        obs_skip(b, 43);
    }
    if ((ptl->general_profile_idc >= 1 && ptl->general_profile_idc <= 5) ||
        ptl->general_profile_compatibility_flag[1] || ptl->general_profile_compatibility_flag[2] ||
        ptl->general_profile_compatibility_flag[3] || ptl->general_profile_compatibility_flag[4] ||
        ptl->general_profile_compatibility_flag[5]) {
        ptl->general_inbld_flag = obs_u1(b);
    } else {
        obs_skip(b, 1);
    }
    ptl->general_level_idc = obs_u8(b);
    for (i = 0; i < maxNumSubLayersMinus1; i++) {
        ptl->sub_layer_profile_present_flag[i] = obs_u1(b);
        ptl->sub_layer_level_present_flag[i] = obs_u1(b);
    }
    if (maxNumSubLayersMinus1 > 0)
        for (i = maxNumSubLayersMinus1; i < 8; i++) obs_skip(b, 2);
    for (i = 0; i < maxNumSubLayersMinus1; i++) {
        if (ptl->sub_layer_profile_present_flag[i]) {
            ptl->sub_layer_profile_space[i] = obs_u(b, 2);
            ptl->sub_layer_tier_flag[i] = obs_u1(b);
            ptl->sub_layer_profile_idc[i] = obs_u(b, 5);
            for (j = 0; j < 32; j++) ptl->sub_layer_profile_compatibility_flag[i][j] = obs_u(b, 1);
            ptl->sub_layer_progressive_source_flag[i] = obs_u1(b);
            ptl->sub_layer_interlaced_source_flag[i] = obs_u1(b);
            ptl->sub_layer_non_packed_constraint_flag[i] = obs_u1(b);
            ptl->sub_layer_frame_only_constraint_flag[i] = obs_u1(b);
            if (ptl->sub_layer_profile_idc[i] == 4 || ptl->sub_layer_profile_compatibility_flag[i][4] ||
                ptl->sub_layer_profile_idc[i] == 5 || ptl->sub_layer_profile_compatibility_flag[i][5] ||
                ptl->sub_layer_profile_idc[i] == 6 || ptl->sub_layer_profile_compatibility_flag[i][6] ||
                ptl->sub_layer_profile_idc[i] == 7 || ptl->sub_layer_profile_compatibility_flag[i][7]) {
                ptl->sub_layer_max_12bit_constraint_flag[i] = obs_u1(b);
                ptl->sub_layer_max_10bit_constraint_flag[i] = obs_u1(b);
                ptl->sub_layer_max_8bit_constraint_flag[i] = obs_u1(b);
                ptl->sub_layer_max_422chroma_constraint_flag[i] = obs_u1(b);
                ptl->sub_layer_max_420chroma_constraint_flag[i] = obs_u1(b);
                ptl->sub_layer_max_monochrome_constraint_flag[i] = obs_u1(b);
                ptl->sub_layer_intra_constraint_flag[i] = obs_u1(b);
                ptl->sub_layer_one_picture_only_constraint_flag[i] = obs_u1(b);
                ptl->sub_layer_lower_bit_rate_constraint_flag[i] = obs_u1(b);
                obs_skip(b, 34);
            } else {
                obs_skip(b, 43);
            }
            /* :739-744 tests row POINTERS sub_layer_profile_compatibility_flag[1..5]
             * (array decay), which are never null: the flag is always read */
            ptl->sub_layer_inbld_flag[i] = obs_u1(b);
        }
        if (ptl->sub_layer_level_present_flag[i]) ptl->sub_layer_level_idc[i] = obs_u8(b);
    }
}

/* hevc_stream.c:758-779: every delta coefficient lands in the same element */
static void read_scaling_list(hevc_scaling_list_data_t* sld, obs_t* b)
{
    int sizeId, matrixId, i;
    for (sizeId = 0; sizeId < 4; sizeId++)
        for (matrixId = 0; matrixId < 6; matrixId += (sizeId == 3) ? 3 : 1) {
            sld->scaling_list_pred_mode_flag[sizeId][matrixId] = obs_u1(b);
            if (!sld->scaling_list_pred_mode_flag[sizeId][matrixId]) {
                sld->scaling_list_pred_matrix_id_delta[sizeId][matrixId] = obs_ue(b);
            } else {
                int coefNum = (1 << (4 + (sizeId << 1))) < 64 ? (1 << (4 + (sizeId << 1))) : 64;
                if (sizeId > 1) sld->scaling_list_dc_coef_minus8[sizeId - 2][matrixId] = obs_se(b);
                for (i = 0; i < coefNum; i++) sld->scaling_list_delta_coef[sizeId][matrixId] = obs_se(b);
            }
        }
}

/* hevc_stream.c:1207-1218: i <= CpbCnt, one entry more than H.265 */
static void read_sub_layer_hrd(hevc_sub_layer_hrd_t* s, obs_t* b, int CpbCnt, int sub_pic)
{
    int i;
    for (i = 0; i <= CpbCnt; i++) {
        int k = i < MAX_CPB_CNT ? i : MAX_CPB_CNT - 1;      /* bounded; the envelope keeps CpbCnt < 32 */
        s->bit_rate_value_minus1[k] = obs_ue(b);
        s->cpb_size_value_minus1[k] = obs_ue(b);
        if (sub_pic) {
            s->cpb_size_du_value_minus1[k] = obs_ue(b);
            s->bit_rate_du_value_minus1[k] = obs_ue(b);
        }
        s->cbr_flag[k] = obs_u1(b);
    }
}

/* hevc_stream.c:1160-1204 */
static void read_hrd(hevc_hrd_t* hrd, obs_t* b, int commonInfPresentFlag, int maxNumSubLayersMinus1)
{
    int i;
    if (commonInfPresentFlag) {
        hrd->nal_hrd_parameters_present_flag = obs_u1(b);
        hrd->vcl_hrd_parameters_present_flag = obs_u1(b);
        if (hrd->nal_hrd_parameters_present_flag || hrd->vcl_hrd_parameters_present_flag) {
            hrd->sub_pic_hrd_params_present_flag = obs_u1(b);
            if (hrd->sub_pic_hrd_params_present_flag) {
                hrd->tick_divisor_minus2 = obs_u8(b);
                hrd->du_cpb_removal_delay_increment_length_minus1 = obs_u(b, 5);
                hrd->sub_pic_cpb_params_in_pic_timing_sei_flag = obs_u1(b);
                hrd->dpb_output_delay_du_length_minus1 = obs_u(b, 5);
            }
            hrd->bit_rate_scale = obs_u(b, 4);
            hrd->cpb_size_scale = obs_u(b, 4);
            if (hrd->sub_pic_hrd_params_present_flag) hrd->cpb_size_du_scale = obs_u(b, 4);
            hrd->initial_cpb_removal_delay_length_minus1 = obs_u(b, 5);
            hrd->au_cpb_removal_delay_length_minus1 = obs_u(b, 5);
            hrd->dpb_output_delay_length_minus1 = obs_u(b, 5);
        }
    }
    for (i = 0; i <= maxNumSubLayersMinus1; i++) {
        hrd->fixed_pic_rate_general_flag[i] = obs_u1(b);
        if (!hrd->fixed_pic_rate_general_flag[i]) hrd->fixed_pic_rate_within_cvs_flag[i] = obs_u1(b);
        if (hrd->fixed_pic_rate_within_cvs_flag[i]) hrd->elemental_duration_in_tc_minus1[i] = obs_ue(b);
        else hrd->low_delay_hrd_flag[i] = obs_u1(b);
        if (hrd->low_delay_hrd_flag[i]) hrd->cpb_cnt_minus1[i] = obs_ue(b);       /* :1194: when the flag is 1 */
        if (hrd->nal_hrd_parameters_present_flag)
            read_sub_layer_hrd(&hrd->sub_layer_hrd_nal[i], b, hrd->cpb_cnt_minus1[i] + 1, hrd->sub_pic_hrd_params_present_flag);
        if (hrd->vcl_hrd_parameters_present_flag)
            read_sub_layer_hrd(&hrd->sub_layer_hrd_vcl[i], b, hrd->cpb_cnt_minus1[i] + 1, hrd->sub_pic_hrd_params_present_flag);
    }
}

/* hevc_stream.c:1032-1085 */
static void read_st_ref_pic_set(orc_hevc* o, hevc_st_ref_pic_set_t* rps, obs_t* b, int stRpsIdx, int num_sets)
{
    int i, j;
    if (stRpsIdx != 0) rps->inter_ref_pic_set_prediction_flag = obs_u1(b);
    if (rps->inter_ref_pic_set_prediction_flag) {
        int RefRpsIdx, lim;
        if (stRpsIdx == num_sets) rps->delta_idx_minus1 = obs_ue(b);
        rps->delta_rps_sign = obs_u1(b);
        rps->abs_delta_rps_minus1 = obs_ue(b);
        RefRpsIdx = stRpsIdx - (rps->delta_idx_minus1 + 1);
        lim = row_ok(RefRpsIdx) ? o->NumDeltaPocs[RefRpsIdx] : 0;
        for (j = 0; j <= lim; j++) {                          /* :1048: <=, as in H.265 */
            int k = col_ok(j) ? j : 31;
            rps->used_by_curr_pic_flag[k] = obs_u1(b);
            if (!rps->used_by_curr_pic_flag[k]) rps->use_delta_flag[k] = obs_u1(b);
        }
    } else {
        rps->num_negative_pics = obs_ue(b);
        rps->num_positive_pics = obs_ue(b);
        for (i = 0; i < rps->num_negative_pics; i++) {
            int k = col_ok(i) ? i : 31;
            rps->delta_poc_s0_minus1[k] = obs_ue(b);
            rps->used_by_curr_pic_s0_flag[k] = obs_u1(b);
            if (row_ok(stRpsIdx)) {
                o->UsedByCurrPicS0[stRpsIdx][k] = rps->used_by_curr_pic_s0_flag[k];
                if (i == 0) o->DeltaPocS0[stRpsIdx][k] = -1 * (rps->delta_poc_s0_minus1[k] + 1);
                else o->DeltaPocS0[stRpsIdx][k] = o->DeltaPocS0[stRpsIdx][k - 1 >= 0 ? k - 1 : 0] - (rps->delta_poc_s0_minus1[k] + 1);
            }
        }
        for (i = 0; i < rps->num_positive_pics; i++) {
            int k = col_ok(i) ? i : 31;
            rps->delta_poc_s1_minus1[k] = obs_ue(b);
            rps->used_by_curr_pic_s1_flag[k] = obs_u1(b);
            if (row_ok(stRpsIdx)) {
                o->UsedByCurrPicS1[stRpsIdx][k] = rps->used_by_curr_pic_s1_flag[k];
                if (i == 0) o->DeltaPocS1[stRpsIdx][k] = rps->delta_poc_s1_minus1[k] + 1;
                else o->DeltaPocS1[stRpsIdx][k] = o->DeltaPocS1[stRpsIdx][k - 1 >= 0 ? k - 1 : 0] + (rps->delta_poc_s1_minus1[k] + 1);
            }
        }
    }
    update_num_delta_pocs(o, rps, stRpsIdx);
}

/* hevc_stream.c:1088-1157 */
static void read_vui(hevc_sps_t* sps, obs_t* b)
{
    hevc_vui_t* vui = &sps->vui;
    vui->aspect_ratio_info_present_flag = obs_u1(b);
    if (vui->aspect_ratio_info_present_flag) {
        vui->aspect_ratio_idc = obs_u8(b);
        if (vui->aspect_ratio_idc == SAR_Extended) {
            vui->sar_width = obs_u(b, 16);
            vui->sar_height = obs_u(b, 16);
        }
    }
    vui->overscan_info_present_flag = obs_u1(b);
    if (vui->overscan_info_present_flag) vui->overscan_appropriate_flag = obs_u1(b);
    vui->video_signal_type_present_flag = obs_u1(b);
    if (vui->video_signal_type_present_flag) {
        vui->video_format = obs_u(b, 3);
        vui->video_full_range_flag = obs_u1(b);
        vui->colour_description_present_flag = obs_u1(b);
        if (vui->colour_description_present_flag) {
            vui->colour_primaries = obs_u8(b);
            vui->transfer_characteristics = obs_u8(b);
            vui->matrix_coefficients = obs_u8(b);
        }
    }
    vui->chroma_loc_info_present_flag = obs_u1(b);
    if (vui->chroma_loc_info_present_flag) {
        vui->chroma_sample_loc_type_top_field = obs_ue(b);
        vui->chroma_sample_loc_type_bottom_field = obs_ue(b);
    }
    vui->neutral_chroma_indication_flag = obs_u1(b);
    vui->field_seq_flag = obs_u1(b);
    vui->frame_field_info_present_flag = obs_u1(b);
    vui->default_display_window_flag = obs_u1(b);
    if (vui->default_display_window_flag) {
        vui->def_disp_win_left_offset = obs_ue(b);
        vui->def_disp_win_right_offset = obs_ue(b);
        vui->def_disp_win_top_offset = obs_ue(b);
        vui->def_disp_win_bottom_offset = obs_ue(b);
    }
    vui->vui_timing_info_present_flag = obs_u1(b);
    if (vui->vui_timing_info_present_flag) {
        vui->vui_num_units_in_tick = obs_u(b, 32);
        vui->vui_time_scale = obs_u(b, 32);
        vui->vui_poc_proportional_to_timing_flag = obs_u1(b);
        if (vui->vui_poc_proportional_to_timing_flag) vui->vui_num_ticks_poc_diff_one_minus1 = obs_ue(b);
        vui->vui_hrd_parameters_present_flag = obs_u1(b);
        if (vui->vui_hrd_parameters_present_flag) read_hrd(&vui->hrd, b, 1, sps->sps_max_sub_layers_minus1);
    }
    vui->bitstream_restriction_flag = obs_u1(b);
    if (vui->bitstream_restriction_flag) {
        vui->tiles_fixed_structure_flag = obs_u1(b);
        vui->motion_vectors_over_pic_boundaries_flag = obs_u1(b);
        vui->restricted_ref_pic_lists_flag = obs_u1(b);
        vui->min_spatial_segmentation_idc = obs_ue(b);
        vui->max_bytes_per_pic_denom = obs_ue(b);
        vui->max_bits_per_min_cu_denom = obs_ue(b);
        vui->log2_max_mv_length_horizontal = obs_ue(b);
        vui->log2_max_mv_length_vertical = obs_ue(b);
    }
}

/* hevc_stream.c:243-300 */
static void read_vps(orc_hevc* o, obs_t* b)
{
    int i, j;
    hevc_vps_t* vps = o->h.vps;
    memset(vps, 0, sizeof(hevc_vps_t));
    vps->vps_video_parameter_set_id = obs_u(b, 4);
    vps->vps_base_layer_internal_flag = obs_u1(b);
    vps->vps_base_layer_available_flag = obs_u1(b);
    vps->vps_max_layers_minus1 = obs_u(b, 6);
    vps->vps_max_sub_layers_minus1 = obs_u(b, 3);
    vps->vps_temporal_id_nesting_flag = obs_u1(b);
    obs_skip(b, 16);
    read_ptl(&vps->ptl, b, 1, vps->vps_max_sub_layers_minus1);
    vps->vps_sub_layer_ordering_info_present_flag = obs_u1(b);
    for (i = (vps->vps_sub_layer_ordering_info_present_flag ? 0 : vps->vps_max_sub_layers_minus1);
         i <= vps->vps_max_sub_layers_minus1; i++) {
        vps->vps_max_dec_pic_buffering_minus1[i] = obs_ue(b);
        vps->vps_max_num_reorder_pics[i] = obs_ue(b);
        vps->vps_max_latency_increase_plus1[i] = obs_ue(b);
    }
    vps->vps_max_layer_id = obs_u(b, 6);
    vps->vps_num_layer_sets_minus1 = obs_ue(b);
    for (i = 1; i <= vps->vps_num_layer_sets_minus1; i++)
        for (j = 0; j <= vps->vps_max_layer_id; j++) {
            int v = obs_u1(b);
            if (i < MAX_NUM_SUBLAYERS && j < MAX_NUM_SUBLAYERS) vps->layer_id_included_flag[i][j] = v;   /* bounded */
        }
    vps->vps_timing_info_present_flag = obs_u1(b);
    if (vps->vps_timing_info_present_flag) {
        vps->vps_num_units_in_tick = obs_u(b, 32);
        vps->vps_time_scale = obs_u(b, 32);
        vps->vps_poc_proportional_to_timing_flag = obs_u1(b);
        if (vps->vps_poc_proportional_to_timing_flag) vps->vps_num_ticks_poc_diff_one_minus1 = obs_ue(b);
        vps->vps_num_hrd_parameters = obs_ue(b);
        for (i = 0; i < vps->vps_num_hrd_parameters; i++) {
            int k = i < MAX_NUM_HRD_PARAM ? i : MAX_NUM_HRD_PARAM - 1;                                   /* bounded */
            vps->hrd_layer_set_idx[k] = obs_ue(b);
            if (i > 0) vps->cprms_present_flag[k] = obs_u1(b);
            read_hrd(&vps->hrd[k], b, vps->cprms_present_flag[k], vps->vps_max_sub_layers_minus1);
        }
    }
    vps->vps_extension_flag = obs_u1(b);
    trailing_bits(b);
}

/* hevc_stream.c:404-415 */
static void read_sps_range_ext(hevc_sps_range_ext_t* e, obs_t* b)
{
    e->transform_skip_rotation_enabled_flag = obs_u1(b);
    e->transform_skip_context_enabled_flag = obs_u1(b);
    e->implicit_rdpcm_enabled_flag = obs_u1(b);
    e->explicit_rdpcm_enabled_flag = obs_u1(b);
    e->extended_precision_processing_flag = obs_u1(b);
    e->intra_smoothing_disabled_flag = obs_u1(b);
    e->high_precision_offsets_enabled_flag = obs_u1(b);
    e->persistent_rice_adaptation_enabled_flag = obs_u1(b);
    e->cabac_bypass_alignment_enabled_flag = obs_u1(b);
}

/* hevc_stream.c:303-401: no rbsp_trailing_bits; copy into sps_table[id] */
static void read_sps(orc_hevc* o, obs_t* b)
{
    int i;
    hevc_sps_t* sps = o->h.sps;
    memset(sps, 0, sizeof(hevc_sps_t));
    sps->sps_video_parameter_set_id = obs_u(b, 4);
    sps->sps_max_sub_layers_minus1 = obs_u(b, 3);
    sps->sps_temporal_id_nesting_flag = obs_u1(b);
    read_ptl(&sps->ptl, b, 1, sps->sps_max_sub_layers_minus1);
    sps->sps_seq_parameter_set_id = obs_ue(b);
    sps->chroma_format_idc = obs_ue(b);
    if (sps->chroma_format_idc == 3) sps->separate_colour_plane_flag = obs_u1(b);
    sps->pic_width_in_luma_samples = obs_ue(b);
    sps->pic_height_in_luma_samples = obs_ue(b);
    sps->conformance_window_flag = obs_u1(b);
    if (sps->conformance_window_flag) {
        sps->conf_win_left_offset = obs_ue(b);
        sps->conf_win_right_offset = obs_ue(b);
        sps->conf_win_top_offset = obs_ue(b);
        sps->conf_win_bottom_offset = obs_ue(b);
    }
    sps->bit_depth_luma_minus8 = obs_ue(b);
    sps->bit_depth_chroma_minus8 = obs_ue(b);
    sps->log2_max_pic_order_cnt_lsb_minus4 = obs_ue(b);
    sps->sps_sub_layer_ordering_info_present_flag = obs_u1(b);
    for (i = (sps->sps_sub_layer_ordering_info_present_flag ? 0 : sps->sps_max_sub_layers_minus1);
         i <= sps->sps_max_sub_layers_minus1; i++) {
        sps->sps_max_dec_pic_buffering_minus1[i] = obs_ue(b);
        sps->sps_max_num_reorder_pics[i] = obs_ue(b);
        sps->sps_max_latency_increase_plus1[i] = obs_ue(b);
    }
    sps->log2_min_luma_coding_block_size_minus3 = obs_ue(b);
    sps->log2_diff_max_min_luma_coding_block_size = obs_ue(b);
    sps->log2_min_luma_transform_block_size_minus2 = obs_ue(b);
    sps->log2_diff_max_min_luma_transform_block_size = obs_ue(b);
    sps->max_transform_hierarchy_depth_inter = obs_ue(b);
    sps->max_transform_hierarchy_depth_intra = obs_ue(b);
    sps->scaling_list_enabled_flag = obs_u1(b);
    if (sps->scaling_list_enabled_flag) {
        sps->sps_scaling_list_data_present_flag = obs_u1(b);
        if (sps->sps_scaling_list_data_present_flag) read_scaling_list(&sps->scaling_list_data, b);
    }
    sps->amp_enabled_flag = obs_u1(b);
    sps->sample_adaptive_offset_enabled_flag = obs_u1(b);
    sps->pcm_enabled_flag = obs_u1(b);
    if (sps->pcm_enabled_flag) {
        sps->pcm_sample_bit_depth_luma_minus1 = obs_u(b, 4);
        sps->pcm_sample_bit_depth_chroma_minus1 = obs_u(b, 4);
        sps->log2_min_pcm_luma_coding_block_size_minus3 = obs_ue(b);
        sps->log2_diff_max_min_pcm_luma_coding_block_size = obs_ue(b);
        sps->pcm_loop_filter_disabled_flag = obs_u1(b);
    }
    sps->num_short_term_ref_pic_sets = obs_ue(b);
    for (i = 0; i < sps->num_short_term_ref_pic_sets; i++) {
        int k = i < MAX_NUM_SHORT_TERM_REF_PICS ? i : MAX_NUM_SHORT_TERM_REF_PICS - 1;                   /* bounded */
        read_st_ref_pic_set(o, &sps->st_ref_pic_set[k], b, i, sps->num_short_term_ref_pic_sets);
    }
    sps->long_term_ref_pics_present_flag = obs_u1(b);
    if (sps->long_term_ref_pics_present_flag) {
        sps->num_long_term_ref_pics_sps = obs_ue(b);
        for (i = 0; i < sps->num_long_term_ref_pics_sps; i++) {
            int k = col_ok(i) ? i : 31;
            sps->lt_ref_pic_poc_lsb_sps[k] = obs_u(b, sps->log2_max_pic_order_cnt_lsb_minus4 + 4);
            sps->used_by_curr_pic_lt_sps_flag[k] = obs_u1(b);
        }
    }
    sps->sps_temporal_mvp_enabled_flag = obs_u1(b);
    sps->strong_intra_smoothing_enabled_flag = obs_u1(b);
    sps->vui_parameters_present_flag = obs_u1(b);
    if (sps->vui_parameters_present_flag) read_vui(sps, b);
    sps->sps_extension_present_flag = obs_u1(b);
    if (sps->sps_extension_present_flag) {
        sps->sps_range_extension_flag = obs_u1(b);
        sps->sps_multilayer_extension_flag = obs_u1(b);
        sps->sps_3d_extension_flag = obs_u1(b);
        sps->sps_extension_5bits = obs_u(b, 5);
    }
    if (sps->sps_range_extension_flag) read_sps_range_ext(&sps->sps_range_ext, b);
    if (sps->sps_seq_parameter_set_id >= 0 && sps->sps_seq_parameter_set_id < 32)                        /* bounded */
        memcpy(o->h.sps_table[sps->sps_seq_parameter_set_id], sps, sizeof(hevc_sps_t));
}

/* hevc_stream.c:503-521 */
static void read_pps_range_ext(hevc_pps_t* pps, obs_t* b)
{
    int i;
    hevc_pps_range_ext_t* e = &pps->pps_range_ext;
    if (pps->transform_skip_enabled_flag) e->log2_max_transform_skip_block_size_minus2 = obs_ue(b);
    e->cross_component_prediction_enabled_flag = obs_u1(b);
    e->chroma_qp_offset_list_enabled_flag = obs_u1(b);
    if (e->chroma_qp_offset_list_enabled_flag) {
        e->diff_cu_chroma_qp_offset_depth = obs_ue(b);
        e->chroma_qp_offset_list_len_minus1 = obs_ue(b);
        for (i = 0; i <= e->chroma_qp_offset_list_len_minus1; i++) {
            int k = col_ok(i) ? i : 31;
            e->cb_qp_offset_list[k] = obs_se(b);
            e->cr_qp_offset_list[k] = obs_se(b);
        }
    }
    e->log2_sao_offset_scale_luma = obs_ue(b);
    e->log2_sao_offset_scale_chroma = obs_ue(b);
}

/* hevc_stream.c:419-500 */
static void read_pps(orc_hevc* o, obs_t* b)
{
    int i;
    hevc_pps_t* pps = o->h.pps;
    memset(pps, 0, sizeof(hevc_pps_t));
    pps->pic_parameter_set_id = obs_ue(b);
    pps->seq_parameter_set_id = obs_ue(b);
    pps->dependent_slice_segments_enabled_flag = obs_u1(b);
    pps->output_flag_present_flag = obs_u1(b);
    pps->num_extra_slice_header_bits = obs_u(b, 3);
    pps->sign_data_hiding_enabled_flag = obs_u1(b);
    pps->cabac_init_present_flag = obs_u1(b);
    pps->num_ref_idx_l0_default_active_minus1 = obs_ue(b);
    pps->num_ref_idx_l1_default_active_minus1 = obs_ue(b);
    pps->init_qp_minus26 = obs_se(b);
    pps->constrained_intra_pred_flag = obs_u1(b);
    pps->transform_skip_enabled_flag = obs_u1(b);
    pps->cu_qp_delta_enabled_flag = obs_u1(b);
    if (pps->cu_qp_delta_enabled_flag) pps->diff_cu_qp_delta_depth = obs_ue(b);
    pps->pps_cb_qp_offset = obs_se(b);
    pps->pps_cr_qp_offset = obs_se(b);
    pps->pps_slice_chroma_qp_offsets_present_flag = obs_u1(b);
    pps->weighted_pred_flag = obs_u1(b);
    pps->weighted_bipred_flag = obs_u1(b);
    pps->transquant_bypass_enabled_flag = obs_u1(b);
    pps->tiles_enabled_flag = obs_u1(b);
    pps->entropy_coding_sync_enabled_flag = obs_u1(b);
    if (pps->tiles_enabled_flag) {
        pps->num_tile_columns_minus1 = obs_ue(b);
        pps->num_tile_rows_minus1 = obs_ue(b);
        pps->uniform_spacing_flag = obs_u1(b);
        if (!pps->uniform_spacing_flag) {
            for (i = 0; i < pps->num_tile_columns_minus1; i++) pps->column_width_minus1[col_ok(i) ? i : 31] = obs_ue(b);
            for (i = 0; i < pps->num_tile_rows_minus1; i++) pps->row_height_minus1[col_ok(i) ? i : 31] = obs_ue(b);
        }
        pps->loop_filter_across_tiles_enabled_flag = obs_u1(b);
    }
    pps->pps_loop_filter_across_slices_enabled_flag = obs_u1(b);
    pps->deblocking_filter_control_present_flag = obs_u1(b);
    if (pps->deblocking_filter_control_present_flag) {
        pps->deblocking_filter_override_enabled_flag = obs_u1(b);
        pps->pps_deblocking_filter_disabled_flag = obs_u1(b);
        if (pps->pps_deblocking_filter_disabled_flag) {          /* :471: when the flag is 1 */
            pps->pps_beta_offset_div2 = obs_se(b);
            pps->pps_tc_offset_div2 = obs_se(b);
        }
    }
    pps->pps_scaling_list_data_present_flag = obs_u1(b);
    if (pps->pps_scaling_list_data_present_flag) read_scaling_list(&pps->scaling_list_data, b);
    pps->lists_modification_present_flag = obs_u1(b);
    pps->log2_parallel_merge_level_minus2 = obs_ue(b);
    pps->slice_segment_header_extension_present_flag = obs_u1(b);
    pps->pps_extension_present_flag = obs_u1(b);
    if (pps->pps_extension_present_flag) {
        pps->pps_range_extension_flag = obs_u1(b);
        pps->pps_multilayer_extension_flag = obs_u1(b);
        pps->pps_3d_extension_flag = obs_u1(b);
        pps->pps_extension_5bits = obs_u1(b);                    /* :488: one bit */
    }
    if (pps->pps_range_extension_flag) read_pps_range_ext(pps, b);
    trailing_bits(b);
    if (pps->pic_parameter_set_id >= 0 && pps->pic_parameter_set_id < 256)                               /* bounded */
        memcpy(o->h.pps_table[pps->pic_parameter_set_id], pps, sizeof(hevc_pps_t));
}

/* hevc_stream.c:944-966: the l1 flag is never read (:959) */
static void read_rplm(orc_hevc* o, const hevc_sps_t* sps, obs_t* b)
{
    int i;
    hevc_slice_header_t* sh = o->h.sh;
    sh->rpld.ref_pic_list_modification_flag_l0 = obs_u1(b);
    if (sh->rpld.ref_pic_list_modification_flag_l0)
        for (i = 0; i <= sh->num_ref_idx_l0_active_minus1; i++)
            sh->rpld.list_entry_l0[col_ok(i) ? i : 31] = obs_u(b, ceil_log2_ref(num_pic_total_curr(o, sps, sh)));
    if (sh->slice_type == HEVC_SLICE_TYPE_B) {
        if (sh->rpld.ref_pic_list_modification_flag_l1)
            for (i = 0; i <= sh->num_ref_idx_l1_active_minus1; i++)
                sh->rpld.list_entry_l1[col_ok(i) ? i : 31] = obs_u(b, ceil_log2_ref(num_pic_total_curr(o, sps, sh)));
    }
}

/* hevc_stream.c:969-1029 */
static void read_pwt(orc_hevc* o, const hevc_sps_t* sps, obs_t* b)
{
    int i, j;
    hevc_slice_header_t* sh = o->h.sh;
    hevc_pred_weight_table_t* pwt = &sh->pwt;
    int ChromaArrayType = 0;
    pwt->luma_log2_weight_denom = obs_ue(b);
    if (sps->separate_colour_plane_flag == 0) ChromaArrayType = sps->chroma_format_idc;
    if (ChromaArrayType != 0) pwt->delta_chroma_log2_weight_denom = obs_se(b);
#define K(i) (col_ok(i) ? (i) : 31)
    for (i = 0; i <= sh->num_ref_idx_l0_active_minus1; i++) pwt->luma_weight_l0_flag[K(i)] = obs_u1(b);
    if (ChromaArrayType != 0)
        for (i = 0; i <= sh->num_ref_idx_l0_active_minus1; i++) pwt->chroma_weight_l0_flag[K(i)] = obs_u1(b);
    for (i = 0; i <= sh->num_ref_idx_l0_active_minus1; i++) {
        if (pwt->luma_weight_l0_flag[K(i)]) {
            pwt->delta_luma_weight_l0[K(i)] = obs_se(b);
            pwt->luma_offset_l0[K(i)] = obs_se(b);
        }
        if (pwt->chroma_weight_l0_flag[K(i)])
            for (j = 0; j < 2; j++) {
                pwt->delta_chroma_weight_l0[K(i)][j] = obs_se(b);
                pwt->delta_chroma_offset_l0[K(i)][j] = obs_se(b);
            }
    }
    if (sh->slice_type == HEVC_SLICE_TYPE_B) {
        for (i = 0; i <= sh->num_ref_idx_l1_active_minus1; i++) pwt->luma_weight_l1_flag[K(i)] = obs_u1(b);
        if (ChromaArrayType != 0)
            for (i = 0; i <= sh->num_ref_idx_l1_active_minus1; i++) pwt->chroma_weight_l1_flag[K(i)] = obs_u1(b);
        for (i = 0; i <= sh->num_ref_idx_l1_active_minus1; i++) {
            if (pwt->luma_weight_l1_flag[K(i)]) {
                pwt->delta_luma_weight_l1[K(i)] = obs_se(b);
                pwt->luma_offset_l1[K(i)] = obs_se(b);
            }
            if (pwt->chroma_weight_l1_flag[K(i)])
                for (j = 0; j < 2; j++) {
                    pwt->delta_chroma_weight_l1[K(i)][j] = obs_se(b);
                    pwt->delta_chroma_offset_l1[K(i)][j] = obs_se(b);
                }
        }
    }
#undef K
}

/* hevc_stream.c:782-941 */
static void read_slice_header(orc_hevc* o, obs_t* b)
{
    int i;
    hevc_slice_header_t* sh = o->h.sh;
    const hevc_nal_t* nal = o->h.nal;
    const hevc_pps_t* pps;
    const hevc_sps_t* sps;

    memset(sh, 0, sizeof(hevc_slice_header_t));                 /* init_slice_hevc :19-24 */
    sh->collocated_from_l0_flag = 1;

    sh->first_slice_segment_in_pic_flag = obs_u1(b);
    if (nal->nal_unit_type >= HEVC_NAL_UNIT_TYPE_BLA_W_LP && nal->nal_unit_type <= HEVC_NAL_UNIT_TYPE_RSV_IRAP_VCL23)
        sh->no_output_of_prior_pics_flag = obs_u1(b);
    sh->pic_parameter_set_id = obs_ue(b);

    pps = (sh->pic_parameter_set_id == 0) ? o->h.pps : &o->zero_pps;               /* :800, bounded */
    sps = (pps->seq_parameter_set_id == 0) ? o->h.sps : &o->zero_sps;               /* :801, bounded */

    sh->num_ref_idx_l0_active_minus1 = pps->num_ref_idx_l0_default_active_minus1;
    sh->num_ref_idx_l1_active_minus1 = pps->num_ref_idx_l1_default_active_minus1;

    if (!sh->first_slice_segment_in_pic_flag) {
        if (pps->dependent_slice_segments_enabled_flag) sh->dependent_slice_segment_flag = obs_u1(b);
        sh->slice_segment_address = obs_u(b, slice_address_bits(sps));
    }
    if (!sh->dependent_slice_segment_flag) {
        for (i = 0; i < pps->num_extra_slice_header_bits; i++) obs_skip(b, 1);
        sh->slice_type = obs_ue(b);
        if (pps->output_flag_present_flag) sh->pic_output_flag = obs_u1(b);
        if (sps->separate_colour_plane_flag == 1) sh->colour_plane_id = obs_u(b, 2);
        if (nal->nal_unit_type != HEVC_NAL_UNIT_TYPE_IDR_W_RADL && nal->nal_unit_type != HEVC_NAL_UNIT_TYPE_IDR_N_LP) {
            sh->slice_pic_order_cnt_lsb = obs_u(b, sps->log2_max_pic_order_cnt_lsb_minus4 + 4);
            sh->short_term_ref_pic_set_sps_flag = obs_u1(b);
            if (!sh->short_term_ref_pic_set_sps_flag)
                read_st_ref_pic_set(o, &sh->st_ref_pic_set, b, sps->num_short_term_ref_pic_sets, sps->num_short_term_ref_pic_sets);
            else if (sps->num_short_term_ref_pic_sets > 1)
                sh->short_term_ref_pic_set_idx = obs_u(b, ceil_log2_ref(sps->num_short_term_ref_pic_sets));
            if (sps->long_term_ref_pics_present_flag) {
                if (sps->num_long_term_ref_pics_sps > 0) sh->num_long_term_sps = obs_ue(b);
                sh->num_long_term_pics = obs_ue(b);
                for (i = 0; i < sh->num_long_term_sps + sh->num_long_term_pics; i++) {
                    int k = col_ok(i) ? i : 31;
                    if (i < sh->num_long_term_sps) {
                        if (sps->num_long_term_ref_pics_sps > 1)
                            sh->lt_idx_sps[k] = obs_u(b, ceil_log2_ref(sps->num_long_term_ref_pics_sps));
                    } else {
                        sh->poc_lsb_lt[k] = obs_u(b, sps->log2_max_pic_order_cnt_lsb_minus4 + 4);
                        sh->used_by_curr_pic_lt_flag[k] = obs_u1(b);
                    }
                    sh->delta_poc_msb_present_flag[k] = obs_u1(b);
                    if (sh->delta_poc_msb_present_flag[k]) sh->delta_poc_msb_cycle_lt[k] = obs_ue(b);
                }
            }
            if (sps->sps_temporal_mvp_enabled_flag) sh->slice_temporal_mvp_enabled_flag = obs_u1(b);
        }
        if (sps->sample_adaptive_offset_enabled_flag) {
            int ChromaArrayType = 0;
            sh->slice_sao_luma_flag = obs_u1(b);
            if (sps->separate_colour_plane_flag == 0) ChromaArrayType = sps->chroma_format_idc;
            if (ChromaArrayType != 0) sh->slice_sao_chroma_flag = obs_u1(b);
        }
        if (sh->slice_type == HEVC_SLICE_TYPE_P || sh->slice_type == HEVC_SLICE_TYPE_B) {
            sh->num_ref_idx_active_override_flag = obs_u1(b);
            if (sh->num_ref_idx_active_override_flag) {
                sh->num_ref_idx_l0_active_minus1 = obs_ue(b);
                if (sh->slice_type == HEVC_SLICE_TYPE_B) sh->num_ref_idx_l1_active_minus1 = obs_ue(b);
            }
            if (pps->lists_modification_present_flag && num_pic_total_curr(o, sps, sh) > 1) read_rplm(o, sps, b);
            if (sh->slice_type == HEVC_SLICE_TYPE_B) sh->mvd_l1_zero_flag = obs_u1(b);
            if (pps->cabac_init_present_flag) sh->cabac_init_flag = obs_u1(b);
            if (sh->slice_temporal_mvp_enabled_flag) {
                if (sh->slice_type == HEVC_SLICE_TYPE_B) sh->collocated_from_l0_flag = obs_u1(b);
                if ((sh->collocated_from_l0_flag && sh->num_ref_idx_l0_active_minus1 > 0) ||
                    (!sh->collocated_from_l0_flag && sh->num_ref_idx_l1_active_minus1 > 0))
                    sh->collocated_ref_idx = obs_ue(b);
            }
            if ((pps->weighted_pred_flag && sh->slice_type == HEVC_SLICE_TYPE_P) ||
                (pps->weighted_bipred_flag && sh->slice_type == HEVC_SLICE_TYPE_B))
                read_pwt(o, sps, b);
            sh->five_minus_max_num_merge_cand = obs_ue(b);
        }
        sh->slice_qp_delta = obs_se(b);
        if (pps->pps_slice_chroma_qp_offsets_present_flag) {
            sh->slice_cb_qp_offset = obs_se(b);
            sh->slice_cr_qp_offset = obs_se(b);
        }
        if (pps->pps_range_ext.chroma_qp_offset_list_enabled_flag) sh->cu_chroma_qp_offset_enabled_flag = obs_u1(b);
        if (pps->deblocking_filter_override_enabled_flag) sh->deblocking_filter_override_flag = obs_u1(b);
        if (sh->deblocking_filter_override_flag) {
            sh->slice_deblocking_filter_disabled_flag = obs_u1(b);
            if (!sh->slice_deblocking_filter_disabled_flag) {
                sh->slice_beta_offset_div2 = obs_se(b);
                sh->slice_tc_offset_div2 = obs_se(b);
            }
        }
        if (pps->pps_loop_filter_across_slices_enabled_flag &&
            (sh->slice_sao_luma_flag || sh->slice_sao_chroma_flag || !sh->slice_deblocking_filter_disabled_flag))
            sh->slice_loop_filter_across_slices_enabled_flag = obs_u1(b);
    }
    if (pps->tiles_enabled_flag || pps->entropy_coding_sync_enabled_flag) {
        sh->num_entry_point_offsets = obs_ue(b);
        if (sh->num_entry_point_offsets > 0) {
            sh->offset_len_minus1 = obs_ue(b);
            for (i = 0; i < sh->num_entry_point_offsets; i++) {
                int v = obs_u(b, sh->offset_len_minus1 + 1);
                if (i < MAX_NUM_ENTRY_POINT_OFFSET) sh->entry_point_offset_minus1[i] = v;               /* bounded (:929) */
            }
        }
    }
    if (pps->slice_segment_header_extension_present_flag) {
        sh->slice_segment_header_extension_length = obs_ue(b);
        for (i = 0; i < sh->slice_segment_header_extension_length; i++) obs_skip(b, 8);
    }
    trailing_bits(b);                                            /* byte_alignment :640-649, same shape */
}

/* hevc_stream.c:600-617 */
static void read_slice_layer(orc_hevc* o, obs_t* b)
{
    read_slice_header(o, b);
    /* :608: sptr = p + !!bits_left, and bits_left is never 0: one byte is skipped */
    o->slice_data_off = (int)(b->p - b->start) + 1;
    o->h.slice_data->rbsp_size = (int)(b->end - b->p) - 1;
    o->h.slice_data->rbsp_buf = (o->h.slice_data->rbsp_size >= 0) ? o->rbsp + o->slice_data_off : NULL;
    trailing_bits(b);                                            /* :616 */
}

/* hevc_stream.c:155-240 */
int orc_read_hevc_nal_unit(orc_hevc* o, const uint8_t* buf, int size)
{
    hevc_nal_t* nal = o->h.nal;
    int nal_size = size, rbsp_size = size, rc;
    obs_t bs, *b = &bs;

    if (size > o->rbsp_cap) {
        free(o->rbsp);
        o->rbsp_cap = size + 64;
        o->rbsp = (uint8_t*)malloc((size_t)o->rbsp_cap);
    }
    if (size > 0) memset(o->rbsp, 0, (size_t)size);              /* calloc :161 */
    o->slice_data_off = -1;
    o->rbsp_size = 0;
    rc = orc_nal_to_rbsp(buf, &nal_size, o->rbsp, &rbsp_size);
    if (rc < 0) return -1;
    o->rbsp_size = rbsp_size;

    obs_init(b, o->rbsp, rbsp_size);
    obs_skip(b, 1);
    nal->nal_unit_type = obs_u(b, 6);
    nal->nal_layer_id = obs_u(b, 6);
    nal->nal_temporal_id_plus1 = obs_u(b, 3);

    switch (nal->nal_unit_type) {
    case HEVC_NAL_UNIT_TYPE_TRAIL_N: case HEVC_NAL_UNIT_TYPE_TRAIL_R:
    case HEVC_NAL_UNIT_TYPE_TSA_N: case HEVC_NAL_UNIT_TYPE_TSA_R:
    case HEVC_NAL_UNIT_TYPE_STSA_N: case HEVC_NAL_UNIT_TYPE_STSA_R:
    case HEVC_NAL_UNIT_TYPE_RADL_N: case HEVC_NAL_UNIT_TYPE_RADL_R:
    case HEVC_NAL_UNIT_TYPE_RASL_N: case HEVC_NAL_UNIT_TYPE_RASL_R:
    case HEVC_NAL_UNIT_TYPE_BLA_W_LP: case HEVC_NAL_UNIT_TYPE_BLA_W_RADL: case HEVC_NAL_UNIT_TYPE_BLA_N_LP:
    case HEVC_NAL_UNIT_TYPE_IDR_W_RADL: case HEVC_NAL_UNIT_TYPE_IDR_N_LP: case HEVC_NAL_UNIT_TYPE_CRA_NUT:
        read_slice_layer(o, b);
        break;
    case HEVC_NAL_UNIT_TYPE_VPS_NUT: read_vps(o, b); break;
    case HEVC_NAL_UNIT_TYPE_SPS_NUT: read_sps(o, b); break;
    case HEVC_NAL_UNIT_TYPE_PPS_NUT: read_pps(o, b); break;
    default:
        return -1;                                               /* :221-222, after nal was updated */
    }
    if (obs_overrun(b)) return -1;                               /* :225 */
    return nal_size;                                             /* :239 */
}

/* ---- extension: the readers the reference has but never calls --------------------------------------------- */

/* h264_stream.c:88-98 */
static int ff_coded_number(obs_t* b)
{
    int n1 = 0, n2;
    do { n2 = (int)obs_u8(b); n1 += n2; } while (n2 == 0xff);
    return n1;
}

/* h264_stream.c:62-84 */
static int more_rbsp_data(const obs_t* bs)
{
    obs_t t = *bs;
    if (obs_eof(bs)) return 0;
    if (obs_u1(&t) == 0) return 1;                 /* bs_peek_u1: no rbsp_stop_bit yet */
    while (!obs_eof(&t)) if (obs_u1(&t) == 1) return 1;
    return 0;
}

/* bs.h:365-370 */
static uint32_t next_bits(const obs_t* bs, int n)
{
    obs_t t = *bs;
    return obs_u(&t, n);
}

int orc_read_extended_nal(const uint8_t* nal_buf, int size, orc_ext_nal* out, int* nal_unit_type)
{
    int nal_size = size, rbsp_size = size, type;
    uint8_t* rbsp = (uint8_t*)calloc(1, (size_t)(size > 0 ? size : 1));
    obs_t bs, *b = &bs;
    memset(out, 0, sizeof(*out));
    if (orc_nal_to_rbsp(nal_buf, &nal_size, rbsp, &rbsp_size) < 0) { free(rbsp); *nal_unit_type = -1; return -1; }
    obs_init(b, rbsp, rbsp_size);
    obs_skip(b, 1);                                 /* hevc_stream.c:176-179 */
    type = (int)obs_u(b, 6);
    (void)obs_u(b, 6);
    (void)obs_u(b, 3);
    *nal_unit_type = type;
    switch (type) {
    case 35:                                        /* :573-577 */
        out->primary_pic_type = (int32_t)obs_u(b, 3);
        trailing_bits(b);
        break;
    case 36: case 37:                               /* :580-587 */
        break;
    case 38:                                        /* :590-597 */
        while (next_bits(b, 8) == 0xFF) { obs_skip(b, 8); out->filler_bytes++; }
        trailing_bits(b);
        break;
    case 39: case 40:                               /* :524-563 */
        do {
            const int pt = ff_coded_number(b), ps = ff_coded_number(b);
            int i;
            if (out->num_sei_messages < ORC_SEI_MAX) {
                out->sei[out->num_sei_messages].payloadType = pt;
                out->sei[out->num_sei_messages].payloadSize = ps;
                out->sei[out->num_sei_messages].payload_off = (uint32_t)obs_pos(b);
            }
            out->num_sei_messages++;
            for (i = 0; i < ps; i++) (void)obs_u8(b);   /* h264_sei.c:80-81 */
        } while (more_rbsp_data(b));
        trailing_bits(b);
        break;
    default:
        free(rbsp);
        return -2;
    }
    {
        const int over = obs_overrun(b);
        free(rbsp);
        return over ? -1 : nal_size;                /* :225, :239 */
    }
}
