/*
 * hbs_parse_ext.h -- the NAL types read_hevc_nal_unit() does not dispatch (hevc_stream.c:221-222 returns -1
 * for them): access unit delimiter (35), end of sequence / bitstream (36, 37), filler data (38), SEI (39, 40).
 * Opt-in (hbs_parse_extended): the default hbs_parse_headers keeps the reference's -1.
 *
 * The readers exist in the reference without a caller: read_hevc_access_unit_delimiter_rbsp (hevc_stream.c:573-577),
 * read_hevc_end_of_seq_rbsp / read_end_of_bitstream_rbsp (:580-587, empty), read_filler_data_rbsp (:590-597), and
 * for SEI the loop of read_sei_rbsp / read_sei_message (:524-563, behind HAVE_SEI) over _read_ff_coded_number
 * (h264_stream.c:88-98), read_sei_payload (h264_sei.c:69-87: the payload bytes, opaque) and more_rbsp_data
 * (h264_stream.c:62-84).  They are restated here on bytes -- every element of these NALs starts on a byte
 * boundary except the AUD's three bits -- with bs.h's rules at the end of the buffer: bits past the end read as 0,
 * the cursor keeps advancing, and a cursor beyond the end (bs_overrun, bs.h:119) turns the result into -1 as
 * hevc_stream.c:225 does for the types it does dispatch.  Compiles for gfx950 and, under tests/sim, for the host.
 */
#ifndef HBS_PARSE_EXT_H
#define HBS_PARSE_EXT_H

#include "hbs_common.h"

namespace hbs {

HBS_HD bool is_extended_nal_type(int t) { return t >= 35 && t <= 40; }

/* what one NAL of those types reads into; rc = bytes of the NAL consumed, or -1 (cursor beyond the RBSP) */
HBS_HD int32_t read_extended_nal(const uint8_t* rbsp, uint32_t size, int nal_unit_type, int32_t consumed, hbs_ext_nal* out)
{
    out->num_sei_messages = 0;
    out->primary_pic_type = 0;
    out->filler_bytes = 0;
    out->reserved = 0;
    for (int i = 0; i < HBS_SEI_MAX_MESSAGES; ++i) { out->sei[i].payloadType = 0; out->sei[i].payloadSize = 0; out->sei[i].payload_off = 0; out->sei[i].reserved = 0; }
    uint64_t pos = 2;                    /* byte cursor behind the NAL header (hevc_stream.c:176-179 read 16 bits) */
    if (nal_unit_type == 35) {
        /* primary_pic_type u(3), then rbsp_trailing_bits: a one and zeros to the byte boundary -- 8 bits in all */
        out->primary_pic_type = pos < size ? (int32_t)(rbsp[pos] >> 5) : 0;
        pos += 1;
    } else if (nal_unit_type == 38) {
        while (pos < size && rbsp[pos] == 0xFFu) { ++pos; ++out->filler_bytes; }      /* bs_next_bits past the end is 0: the loop ends there */
        pos += 1;                        /* rbsp_trailing_bits from a byte boundary: one whole byte */
    } else if (nal_unit_type == 39 || nal_unit_type == 40) {
        /* position of the last set bit of the RBSP: more_rbsp_data() asks whether a one follows the next bit */
        int64_t last_one_byte = (int64_t)size - 1;
        while (last_one_byte >= 0 && rbsp[last_one_byte] == 0) --last_one_byte;
        uint32_t last_one_bit = 0;       /* 0 = most significant */
        if (last_one_byte >= 0) { const uint32_t v = rbsp[last_one_byte]; last_one_bit = 7u - (uint32_t)__builtin_ctz(v); }
        bool more;
        do {
            uint32_t type = 0, sz = 0, b;
            do { b = pos < size ? rbsp[pos] : 0u; ++pos; type += b; } while (b == 0xFFu);     /* _read_ff_coded_number */
            do { b = pos < size ? rbsp[pos] : 0u; ++pos; sz += b; } while (b == 0xFFu);
            if (out->num_sei_messages < HBS_SEI_MAX_MESSAGES) {
                hbs_sei_message* m = &out->sei[out->num_sei_messages];
                m->payloadType = (int32_t)type; m->payloadSize = (int32_t)sz;
                m->payload_off = pos < size ? (uint32_t)pos : size;
            }
            ++out->num_sei_messages;
            pos += sz;                   /* the payload: payloadSize calls of bs_read_u8, whatever is there */
            /* more_rbsp_data (h264_stream.c:62-84) at a byte boundary */
            if (pos >= size) more = false;
            else if (!(rbsp[pos] & 0x80u)) more = true;                                          /* no stop bit yet */
            else more = last_one_byte > (int64_t)pos || (last_one_byte == (int64_t)pos && last_one_bit > 0u);   /* a later one: it was not the stop bit */
        } while (more);
        pos += 1;                        /* rbsp_trailing_bits */
    }
    /* 36, 37: nothing to read */
    return pos > (uint64_t)size ? -1 : consumed;            /* bs_overrun, hevc_stream.c:225; :239 */
}

} // namespace hbs
#endif
