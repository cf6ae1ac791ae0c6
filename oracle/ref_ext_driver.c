/*
 * ref_ext_driver.c -- TEST INFRASTRUCTURE, not product code; built only where /root/reference exists (it includes the
 * reference's own headers from there, -I$(REF); nothing of them is copied).
 *
 * Drives the readers the reference ships but never dispatches, as read_hevc_nal_unit would if it had a case for
 * them: nal_to_rbsp, bs_new over the RBSP, the 16 header bits (hevc_stream.c:176-179), then
 *   35      read_hevc_access_unit_delimiter_rbsp (hevc_stream.c:573-577)           -- the real function
 *   38      read_filler_data_rbsp (:590-597)                                       -- the real function
 *   39, 40  the message loop of read_sei_rbsp / read_sei_message (:524-563, compiled out behind HAVE_SEI) spelled
 *           out here over the real _read_ff_coded_number (h264_stream.c:88), read_sei_payload (h264_sei.c:69),
 *           more_rbsp_data (h264_stream.c:62) and read_hevc_rbsp_trailing_bits
 * and returns what :225 / :239 would: -1 on bs_overrun, else the NAL bytes consumed.
 */
#include <stdlib.h>
#include <string.h>
#include "bs.h"
#include "h264_stream.h"
#include "hevc_stream.h"
#include "h264_sei.h"

void read_hevc_access_unit_delimiter_rbsp(hevc_stream_t* h, bs_t* b);
void read_filler_data_rbsp(bs_t* b);
void read_hevc_rbsp_trailing_bits(bs_t* b);
int _read_ff_coded_number(bs_t* b);
void read_sei_payload(sei_t* s, bs_t* b);
int more_rbsp_data(bs_t* bs);

#define REF_SEI_MAX 6
typedef struct {
    int32_t num_sei_messages, primary_pic_type;
    uint32_t filler_bytes, reserved;
    struct { int32_t payloadType, payloadSize; uint32_t payload_off, reserved; } sei[REF_SEI_MAX];
} ref_ext_nal;

int ref_read_extended_nal(const uint8_t* nal_buf, int size, ref_ext_nal* out, int* nal_unit_type)
{
    int nal_size = size, rbsp_size = size, type, rc;
    uint8_t* rbsp = (uint8_t*)calloc(1, (size_t)(size > 0 ? size : 1));
    hevc_stream_t* h;
    bs_t* b;
    memset(out, 0, sizeof(*out));
    if (nal_to_rbsp(nal_buf, &nal_size, rbsp, &rbsp_size) < 0) { free(rbsp); *nal_unit_type = -1; return -1; }
    h = hevc_new();
    b = bs_new(rbsp, rbsp_size);
    bs_skip_u(b, 1);
    type = bs_read_u(b, 6);
    (void)bs_read_u(b, 6);
    (void)bs_read_u(b, 3);
    *nal_unit_type = type;
    switch (type) {
    case 35:
        read_hevc_access_unit_delimiter_rbsp(h, b);
        out->primary_pic_type = h->aud->primary_pic_type;
        break;
    case 36: case 37:
        break;
    case 38: {
        const uint8_t* p0 = b->p;
        read_filler_data_rbsp(b);
        /* ff bytes skipped = cursor movement minus the trailing byte, where the cursor stayed inside the buffer */
        out->filler_bytes = (uint32_t)((b->p - p0) > 0 ? (b->p - p0) - 1 : 0);
        break;
    }
    case 39: case 40:
        do {
            sei_t* s = sei_new();
            s->payloadType = _read_ff_coded_number(b);
            s->payloadSize = _read_ff_coded_number(b);
            if (out->num_sei_messages < REF_SEI_MAX) {
                out->sei[out->num_sei_messages].payloadType = s->payloadType;
                out->sei[out->num_sei_messages].payloadSize = s->payloadSize;
                out->sei[out->num_sei_messages].payload_off = (uint32_t)(b->p > b->end ? b->end - b->start : b->p - b->start);
            }
            out->num_sei_messages++;
            read_sei_payload(s, b);
            sei_free(s);
        } while (more_rbsp_data(b));
        read_hevc_rbsp_trailing_bits(b);
        break;
    default:
        bs_free(b); hevc_free(h); free(rbsp);
        return -2;
    }
    rc = bs_overrun(b) ? -1 : nal_size;
    bs_free(b);
    hevc_free(h);
    free(rbsp);
    return rc;
}
