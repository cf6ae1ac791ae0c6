/*
 * h264_sei.h -- SEI container type of the legacy API (reference h264_sei.h:37-49).
 * HEVC SEI parsing is compiled out in the reference (HAVE_SEI is never defined,
 * hevc_stream.c:203-207): SEI NAL units make read_hevc_nal_unit() return -1.
 * Only the type is kept, for source compatibility of code that names it.
 */
#ifndef _H264_SEI_H
#define _H264_SEI_H        1

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct
{
    int payloadType;
    int payloadSize;
    uint8_t* payload;
} sei_t;

#ifdef __cplusplus
}
#endif

#endif
