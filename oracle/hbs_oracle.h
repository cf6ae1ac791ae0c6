/*
 * hbs_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99) of the hot path of leslie-wang/hevcbitstream:
 * the Annex-B byte layer (h264_nal.c), the bit reader (bs.h, read half) and the
 * HEVC header readers (hevc_stream.c, read direction).  It is the checker that
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg compare the
 * HIP path against.  Nothing under hevcbitstream_amd/ may include, link or call
 * it; the product library fails loudly when its HIP code object is missing.
 *
 * Parity pin: every function here is checked (tests/test_oracle_vs_ref.py, in
 * the dev container) against the REAL reference compiled from /root/reference
 * by oracle/Makefile into oracle/_ref/libhevcref.so, and against the golden
 * vectors under tests/golden/ that were generated from that build.
 *
 * Convention for reads past the end of a buffer: the reference performs a few
 * unchecked reads at buf[size..size+2] (h264_nal.c:47-48,65-66); this
 * restatement defines such bytes as 0xFF.  The reference is compared on
 * buffers that really are followed by 0xFF bytes.
 */
#ifndef HBS_ORACLE_H
#define HBS_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- Annex-B byte layer (reference: h264_nal.c) ------------------------- */

/* h264_nal.c:38-76, same contract, int sizes. */
int orc_find_nal_unit(const uint8_t* buf, int size, int* nal_start, int* nal_end);
/* 64-bit restatement of the same loop for streams > 2 GiB. */
int64_t orc_find_nal_unit64(const uint8_t* buf, int64_t size, int64_t* nal_start, int64_t* nal_end);
/* h264_nal.c:147-200 */
int orc_nal_to_rbsp(const uint8_t* nal_buf, int* nal_size, uint8_t* rbsp_buf, int* rbsp_size);
/* h264_nal.c:92-132 */
int orc_rbsp_to_nal(const uint8_t* rbsp_buf, const int* rbsp_size, uint8_t* nal_buf, int* nal_size);

/* One record of the whole-stream index: what the NAL loop of
 * hevc_analyze.c:135-205 visits when the whole stream is one window.
 * Layout is identical to hbs_nal_entry in include/hevcbitstream_amd.h. */
typedef struct {
    uint64_t start;     /* first payload byte (after 00 00 01)                */
    uint64_t end;       /* one past the last payload byte                     */
    uint64_t rbsp_off;  /* offset of this NAL's RBSP bytes in the RBSP arena  */
    uint32_t rbsp_len;  /* RBSP bytes (NAL length minus emulation bytes)      */
    int32_t  status;    /* ORC_ST_* flags                                     */
} orc_nal_entry;

#define ORC_ST_ERROR        1  /* nal_to_rbsp would return -1                  */
#define ORC_ST_TRAILING03   2  /* NAL ends in 00 00 03: consumed = len-1       */
#define ORC_ST_UNTERMINATED 4  /* last NAL: find_nal_unit returned -1          */

/* Walk the whole stream the way hevc_analyze.c:135-205 does (find_nal_unit
 * until it returns <= 0, then the "last NAL" of the -1 path), filling up to
 * `cap` entries (start/end/status only).  Returns the number of NALs.
 * *stop_reason: 0 = stream exhausted (ret 0), -1 = last NAL unterminated,
 * 1 = stopped at an empty NAL (ret 0 with a start code found). */
int64_t orc_index_stream(const uint8_t* buf, int64_t size, orc_nal_entry* out, int64_t cap, int* stop_reason);

/* nal_to_rbsp over every indexed NAL into a packed arena (rbsp_off/rbsp_len/
 * status filled).  NALs whose conversion fails get ORC_ST_ERROR and their
 * arena bytes are unspecified (the pattern-rule length is still reserved).
 * Returns total arena bytes. */
int64_t orc_extract_rbsp(const uint8_t* buf, orc_nal_entry* idx, int64_t n, uint8_t* arena, int64_t arena_cap);

/* Inverse: re-emit Annex-B from the RBSP arena; gap_k (zeros + 01 before NAL
 * k) is taken from start_k - end_{k-1} (start_0 for the first).  Returns
 * bytes written. */
int64_t orc_emit_annexb(const uint8_t* arena, const orc_nal_entry* idx, int64_t n, uint8_t* out, int64_t out_cap);

/* ---- HEVC header layer (reference: hevc_stream.c read direction) --------- */

typedef struct orc_hevc orc_hevc;
/* hevc_nal.c:34-57 / :64-91 */
orc_hevc* orc_hevc_new(void);
void orc_hevc_free(orc_hevc* o);
/* the parser object (include/hevc_stream.h layout, as the reference's hevc_stream_t) */
struct hevc_stream_s;
void* orc_hevc_stream_ptr(orc_hevc* o);
/* hevc_stream.c:155-240 */
int orc_read_hevc_nal_unit(orc_hevc* o, const uint8_t* buf, int size);
/* RBSP of the NAL read last, and where its slice payload copy starts (-1: none) */
const uint8_t* orc_hevc_rbsp(orc_hevc* o, int* size);
int orc_hevc_slice_data_off(orc_hevc* o);
const int* orc_hevc_tables(orc_hevc* o);      /* the derived RPS tables as they stand: 3 x 32 counts, then 4 x 32 x 32 values */

/* ---- the NAL types read_hevc_nal_unit never dispatches (35..40): what their unused readers would read -------
 * hevc_stream.c:573-577 (AUD), :580-587 (EOS / EOB: nothing), :590-597 (filler data), :524-563 (SEI message loop,
 * behind HAVE_SEI) with h264_stream.c:62-98 (more_rbsp_data, _read_ff_coded_number) and h264_sei.c:69-87 (opaque
 * payload).  Same frame as read_hevc_nal_unit: nal_to_rbsp, 16 header bits, the reader, -1 on bs_overrun.
 * Returns -2 when the NAL is of another type. */
#define ORC_SEI_MAX 6
typedef struct {
    int32_t num_sei_messages, primary_pic_type;
    uint32_t filler_bytes, reserved;
    struct { int32_t payloadType, payloadSize; uint32_t payload_off, reserved; } sei[ORC_SEI_MAX];
} orc_ext_nal;
int orc_read_extended_nal(const uint8_t* nal_buf, int size, orc_ext_nal* out, int* nal_unit_type);

#ifdef __cplusplus
}
#endif
#endif
