/*
 * h264_stream.h -- legacy byte-layer entry points, drop-in for the reference's
 * header of the same name (reference h264_stream.h:54-71).
 *
 * Each function is a thin host wrapper over the batch API of
 * hevcbitstream_amd.h: the caller's buffer is copied to the GPU, the HIP kernel
 * runs, the answer comes back.  They exist so that code written against the
 * reference (hevc_analyze.c) links unmodified; throughput work should call
 * hbs_index_extract / hbs_emit_annexb on device-resident streams instead.
 * There is no CPU implementation behind them: without a gfx950 GPU they print one
 * diagnostic on stderr and return the reference's failure values (-1, or 0 NALs found).
 */
#ifndef _H264_STREAM_H
#define _H264_STREAM_H        1

#include <stdint.h>
#include <stdio.h>
#include <assert.h>

#include "bs.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Table E-1 sample aspect ratio indicators (reference h264_stream.h:35-52) */
#define SAR_Unspecified  0
#define SAR_1_1          1
#define SAR_12_11        2
#define SAR_10_11        3
#define SAR_16_11        4
#define SAR_40_33        5
#define SAR_24_11        6
#define SAR_20_11        7
#define SAR_32_11        8
#define SAR_80_33        9
#define SAR_18_11       10
#define SAR_15_11       11
#define SAR_64_33       12
#define SAR_160_99      13
#define SAR_Extended   255

/* reference h264_nal.c:38-76: first NAL in buf[0,size): returns its length and
 * sets *nal_start / *nal_end; 0 = no start code (or an empty NAL); -1 = start
 * found but no end: *nal_end = size */
extern int find_nal_unit(uint8_t* buf, int size, int* nal_start, int* nal_end);
/* reference h264_nal.c:92-132 */
extern int rbsp_to_nal(const uint8_t* rbsp_buf, const int* rbsp_size, uint8_t* nal_buf, int* nal_size);
/* reference h264_nal.c:147-200 */
extern int nal_to_rbsp(const uint8_t* nal_buf, int* nal_size, uint8_t* rbsp_buf, int* rbsp_size);

/* reference h264_stream.c:117-126: hex dump to h264_dbgfile (stdout when NULL) */
extern void debug_bytes(uint8_t* buf, int len);

/* destination of debug_bytes and of hevc_analyze's "!! Found NAL" lines (h264_stream.c:33) */
extern FILE* h264_dbgfile;

#ifdef __cplusplus
}
#endif

#endif
