/*
 * bs.h -- bit cursor type of the legacy API (reference bs.h:34-40).
 *
 * The reference's header is an all-inline bit reader/writer; applications of
 * the library only meet `bs_t*` in the prototypes of h264_stream.h.  Here the
 * type is kept layout-identical (32 bytes on LP64) and the bit I/O itself is
 * internal to the library: on the GPU it is the whole-field reader / writer
 * `BitIOT` of hevcbitstream_amd/csrc/hbs_parse.h over the 64-bit windows of
 * hevcbitstream_amd/csrc/hbs_bitfast.h, in the test oracle a bit-serial
 * restatement (oracle/hbs_oracle_bits.h).
 */
#ifndef _H264_BS_H
#define _H264_BS_H        1

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct
{
    uint8_t* start;      /* first byte of the buffer            */
    uint8_t* p;          /* byte the cursor is in               */
    uint8_t* end;        /* one past the last byte              */
    int bits_left;       /* unread bits of *p: 8 .. 1           */
} bs_t;

#ifdef __cplusplus
}
#endif

#endif
