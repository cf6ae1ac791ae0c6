/*
 * hevcbitstream_amd.h -- batch C ABI of the MI355X (gfx950) Annex-B indexer /
 * RBSP extractor.  Plain C: pointers and sizes only, no C++/torch types.
 *
 * This is the 64-bit, whole-stream form of the reference's per-NAL byte layer:
 *
 *   hbs_index_extract   replaces the loop  find_nal_unit() + nal_to_rbsp()
 *                       reference: h264_nal.c:38-76 (find_nal_unit),
 *                       h264_nal.c:147-200 (nal_to_rbsp), driven as in
 *                       hevc_analyze.c:135-205 and hevc_stream.c:161-165
 *   hbs_emit_annexb     replaces  rbsp_to_nal() per NAL + start-code emission
 *                       reference: h264_nal.c:92-132, hevc_stream.c:1324-1327
 *   hbs_parse_headers   replaces  read_hevc_nal_unit() per NAL
 *                       reference: hevc_stream.c:155-240 and the readers it
 *                       dispatches to (:243-1218)
 *   hbs_index_parse     find_nal_unit over the stream + read_hevc_nal_unit per NAL without
 *                       an RBSP arena (each header is stripped from the stream by itself)
 *   hbs_parse_headers_trace   the same with the per-field trace read_debug_hevc_nal_unit
 *                       prints (hevc_stream.c:2343-3434)
 *   hbs_write_headers   replaces  write_hevc_nal_unit() per NAL up to rbsp_to_nal
 *                       reference: hevc_stream.c:1249-1327 and the writers behind it
 *   hbs_index_extract_host    hbs_index_extract for a stream in HOST memory of any
 *                       length, windowed (replaces the reader of hevc_analyze.c:124-210)
 *   hbs_synth_*         synthetic stream S(seed, n_nals, mode) of SURVEY.md
 *                       8(d) (no reference counterpart: it ships no streams)
 *
 * All `d_` pointers are DEVICE pointers of the context's GPU (hipMalloc'ed or
 * torch-allocated), 16-byte aligned.  Calls enqueue work on the context's HIP
 * stream and return without synchronising unless stated otherwise.  Return
 * value: 0 on success, a negative HBS_E_* code otherwise.  There is no CPU
 * fallback: without a usable gfx950 device every call fails with
 * HBS_E_NO_DEVICE.
 *
 * The legacy single-NAL symbols of the reference (find_nal_unit, nal_to_rbsp,
 * rbsp_to_nal, read_hevc_nal_unit, ...) are declared in h264_stream.h /
 * hevc_stream.h next to this file and are thin host wrappers over this API.
 */
#ifndef HEVCBITSTREAM_AMD_H
#define HEVCBITSTREAM_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HBS_E_NO_DEVICE   (-1)   /* no gfx950 GPU / HIP runtime failure at init  */
#define HBS_E_HIP         (-2)   /* a HIP call failed (see hbs_last_error)       */
#define HBS_E_ARG         (-3)   /* bad argument (alignment, null, capacity 0)   */
#define HBS_E_CAPACITY    (-4)   /* index / arena / output capacity too small    */
#define HBS_E_TIMEOUT     (-5)   /* in-kernel look-back wait gave up (bug guard) */
#define HBS_E_DEPTH       (-6)   /* hbs_parse_headers_compact / _materialize on an out-of-spec stream whose answer needs the
                                    sequential parse (a chain of slice-own RPS sets deeper than 3): take hbs_parse_headers */

/* per-NAL status flags */
#define HBS_ST_ERROR        1    /* nal_to_rbsp() would return -1 (h264_nal.c:156-167) */
#define HBS_ST_TRAILING03   2    /* NAL ends in 00 00 03: the 03 is dropped and
                                    nal_to_rbsp() consumes len-1 (h264_nal.c:170-173) */
#define HBS_ST_UNTERMINATED 4    /* last NAL: find_nal_unit() returned -1 with
                                    nal_end = size (h264_nal.c:71)                */

/* One NAL of the whole-stream index: what find_nal_unit() reports for it when
 * the stream is walked as in hevc_analyze.c:135-205, with 64-bit offsets, plus
 * where nal_to_rbsp() of it went. 32 bytes. */
typedef struct hbs_nal_entry {
    uint64_t start;      /* offset of the first payload byte (after 00 00 01)   */
    uint64_t end;        /* offset one past the last payload byte               */
    uint64_t rbsp_off;   /* offset of this NAL's RBSP in the RBSP arena         */
    uint32_t rbsp_len;   /* RBSP bytes = (end-start) - emulation-prevention bytes */
    int32_t  status;     /* HBS_ST_* flags                                      */
} hbs_nal_entry;

/* Result block of hbs_index_extract (written on the DEVICE; copy it with
 * hbs_read_summary). */
typedef struct hbs_summary {
    uint64_t nal_count;     /* NALs in the index (what the reference loop visits) */
    uint64_t nal_found;     /* start codes found before truncation/capacity clip  */
    uint64_t rbsp_bytes;    /* bytes written to the RBSP arena                    */
    uint64_t stream_bytes;  /* bytes scanned                                      */
    int32_t  stop_reason;   /* 0: no more start codes; -1: last NAL unterminated;
                               1: stopped at an empty NAL (find_nal_unit == 0 with
                               a start code found, hevc_analyze.c:135)            */
    int32_t  error;         /* 0 or HBS_E_CAPACITY / HBS_E_TIMEOUT / HBS_E_ARG    */
    uint64_t reserved[3];
} hbs_summary;

typedef struct hbs_ctx hbs_ctx;

/* Create a context on HIP device `device` (one per GPU / per rank).
 *
 * Threads and streams.  A context owns ONE set of scratch on its GPU (run header, look-back words, ticket, the workspaces of
 * K3 / K4 / K5, the window buffers of hbs_index_extract_host) and binds to ONE stream at a time, so:
 *   - a context is used by one thread at a time (the library takes no lock in the batch API: callers that share a
 *     context serialise their calls themselves);
 *   - DIFFERENT contexts are independent: any number of threads, one context each, may call at the same time on the
 *     same or on different GPUs (tests/test_gpu_legacy.py::test_batch_api_two_contexts_two_threads);
 *   - all work of a context is ordered by the stream it is bound to when the call is made.  Re-binding the stream
 *     (hbs_ctx_set_stream) while work enqueued through the previous one may still be running is the caller's to order
 *     (an event between the two streams): the scratch is shared by both.  hbs_index_extract_host uses two private streams
 *     and returns only when they are idle.
 * The legacy single-NAL symbols (find_nal_unit ... write_hevc_nal_unit) share one internal context behind a process-wide
 * lock: safe from any thread, serialised (hbs_legacy.c). */
int  hbs_ctx_create(hbs_ctx** out, int device);
void hbs_ctx_destroy(hbs_ctx* ctx);
/* A new context enqueues on a non-blocking stream of its own.  set_stream
 * makes it use the caller's HIP stream instead (hipStream_t passed as void*;
 * NULL = the HIP null stream); use_own_stream goes back. */
int  hbs_ctx_set_stream(hbs_ctx* ctx, void* hip_stream);
int  hbs_ctx_use_own_stream(hbs_ctx* ctx);
void* hbs_ctx_get_stream(hbs_ctx* ctx);
int  hbs_ctx_synchronize(hbs_ctx* ctx);
/* Measurement aid: when enabled, hbs_index_extract records HIP events on the
 * context's stream around its dominant kernel (the fused scan/extract kernel)
 * only (for an index-only call: its four kernels); hbs_ctx_kernel_ms waits for the
 * last such launch and returns its duration, hbs_ctx_kernel_ms_back(back) that of the
 * call `back` calls earlier (the event pairs of the last 64 timed calls are kept, so a
 * benchmark can read every step of its timed loop after the loop, without a wait
 * inside it).  hbs_ctx_grid reports the persistent grid used. */
int  hbs_ctx_enable_timing(hbs_ctx* ctx, int on);
int  hbs_ctx_kernel_ms(hbs_ctx* ctx, float* ms);
int  hbs_ctx_kernel_ms_back(hbs_ctx* ctx, int back, float* ms);
int  hbs_ctx_grid(hbs_ctx* ctx, int* blocks, int* blocks_per_cu);
/* The scan + extract kernels are persistent and fill the GPU: a kernel of another stream (RCCL's, in hbs_gather_index running
 * beside the next scan) finds no CU before the scan ends.  `spare` workgroup slots are left free from now on (0 = none, the
 * default; a multi-GPU caller that overlaps the index gather with the next scan wants ~32 of the 512: with 8 or 16 RCCL still
 * waited for the scan to end, with 32 a one-rank gather of 54 MB took 0.14 ms beside the scan instead of 5.3 ms behind it).
 * Covers every scan kernel: the register-tile and LDS-image kernels launch `spare` workgroups fewer, the index-only streaming
 * kernel 4 x `spare` one-wavefront workgroups fewer.  An HBS_GRID_BLOCKS debugging cap stays a ceiling of its own. */
int  hbs_ctx_reserve_workgroups(hbs_ctx* ctx, int spare);
/* Tile hand-out of the persistent kernels (scan + extraction, arena-tile emit).  Default (0): every tile by atomic ticket, in
 * arrival order -- a workgroup's look-back only ever waits for tiles that a RUNNING workgroup has claimed, so calls of several
 * contexts or processes may share a device.  on = 1: the caller states that while this context's calls run, no other persistent
 * kernel uses the device; a workgroup's first tile is then its own number (later ones by ticket), which saves the start-of-call
 * queue on the ticket word (~1 % of a 1 GiB call, nothing at 16 GiB).  With static first tiles, forward progress assumes every
 * workgroup of the grid is resident at once: do NOT set it when two contexts (or ranks) scan one device concurrently. */
int  hbs_ctx_set_device_exclusive(hbs_ctx* ctx, int on);
/* Implementations of the scan kernel, with identical results:
 * 4 = event-sparse, tile held in registers: 48 rows of 1 KiB per wavefront, 192 KiB tiles, up to 512 candidate chunks a tile
 *     (hbs_scan4.hip; the fastest on coded video, where zero pairs are rare, and the slowest on zero-heavy data),
 * 6 = the same kernel with 24 rows per wavefront: 96 KiB tiles, up to 1024 candidate chunks a tile, all four wavefronts on them
 *     (hbs_scan4_r24.hip, round 6; streams that are dense but regular -- NALs of ~120 to ~450 bytes --, with or without an arena),
 * 2 = tile staged in an LDS image (hbs_scan.hip; same speed on any data),
 * 5 = index only (no RBSP arena asked for): nothing has to stay in registers, so the bytes are
 *     streamed and only the flagged chunks are looked at again (hbs_scan5.hip); with an arena it means 4,
 * 0 = automatic, the default: a density probe (64 windows of 16 KiB) runs in front and the kernel is picked from it on the
 *     device, without a host round trip: 4 up to one candidate chunk in 44, 6 up to one in 6.5, 2 beyond; when no arena is asked
 *     for and the stream is 1 GiB or more: 5 up to one in 9, 2 beyond.
 * Environment HBS_KERNEL=0|2|4|5|6 sets the default.  hbs_ctx_last_kernel waits for the last
 * hbs_index_extract and says which kernel ran it. */
int  hbs_ctx_set_kernel(hbs_ctx* ctx, int variant);
int  hbs_ctx_get_kernel(hbs_ctx* ctx);
/* Kernel 4 walks a DENSE tile (one with more than 512 candidate chunks in its 192 KiB: padding, zero stuffing) chunk by chunk,
 * and every tile behind it waits for its count.  From round 5 such tiles are counted AHEAD of the main kernel: the call's first
 * launch samples every tile, a small kernel counts the ones the sample marks, and the main kernel takes those counts instead of
 * walking the tile a first time (hbs_scan4.hip, "dense tiles counted ahead").  mode 1 (default): streams of 3 GiB and more;
 * 0: never; 2: any stream that has more than one tile.  Results are identical in all three.  Environment HBS_COUNT_AHEAD=0|1|2
 * sets the default.  The table costs 76 bytes of device memory per 192 KiB of stream; nothing of it lives on the host, so a call captured into a HIP graph may be replayed.
 * The table is allocated by the first call that needs it (and again by a call on a longer stream): such a call must not be made
 * inside a stream capture -- make one call of at least that size before capturing (round 5's advice).
 * Since round 6 the same mode governs hbs_emit_annexb's arena-tile kernel, which samples the arena and counts the dense tiles it
 * lists ahead in the same way (two launches in front of the main pass): mode 1 = arenas of 3 GiB and more. */
int  hbs_ctx_set_count_ahead(hbs_ctx* ctx, int mode);
int  hbs_ctx_last_kernel(hbs_ctx* ctx);
/* Text of the last HIP/driver error seen by this context. */
const char* hbs_last_error(hbs_ctx* ctx);
/* Library/self description: "hevcbitstream_amd <ver> gfx950 ..." */
const char* hbs_version(void);

/*
 * Start-code scan + NAL index + RBSP extraction over one stream resident in
 * HBM (single pass: every stream byte is read once, every RBSP byte written
 * once).
 *
 *   d_stream, stream_bytes   Annex-B bytes
 *   d_index, index_cap       out: hbs_nal_entry[index_cap]; the call zeroes it
 *   d_rbsp, rbsp_cap         out: packed RBSP arena (NAL k at rbsp_off, rbsp_len);
 *                            NULL = index only (rbsp_off/rbsp_len still filled)
 *   d_summary                out: hbs_summary on the device
 *
 * Semantics (bit-exact with the reference on the same bytes):
 *   - entries are the NALs the loop `while (find_nal_unit(p, sz, &s, &e) > 0)`
 *     of hevc_analyze.c:135-177 visits over the whole stream, followed by the
 *     "last NAL" of the -1 path (:190-205), offsets relative to d_stream;
 *   - RBSP of NAL k is what nal_to_rbsp() writes for it; for a NAL it rejects
 *     (HBS_ST_ERROR) the arena holds every byte except 00 00 03 emulation
 *     bytes (the reference leaves its output unspecified there);
 *   - bytes past the end of the stream are taken as 0xFF where the reference
 *     reads them unchecked (h264_nal.c:47-48, 65-66).
 */
int hbs_index_extract(hbs_ctx* ctx,
                      const uint8_t* d_stream, uint64_t stream_bytes,
                      hbs_nal_entry* d_index, uint64_t index_cap,
                      uint8_t* d_rbsp, uint64_t rbsp_cap,
                      hbs_summary* d_summary);

/*
 * The same for a stream in HOST memory of any length (larger than device memory is fine): the
 * stream is uploaded window by window (window_bytes each, >= 4096, rounded down to 16), the
 * upload of one window overlapping the scan and the download of the previous one; index and RBSP
 * arrive in host memory with offsets relative to h_stream / h_rbsp, identical to what
 * hbs_index_extract returns for the whole stream.  Replaces the windowed reader of
 * hevc_analyze.c:124-210 (and gets a NAL that straddles two reads right).
 * A NAL (with the zeros in front of it) that does not fit the window makes the window GROW (round 5): the walk goes on from
 * the end of the last complete NAL with device windows of twice the size, as often as it takes, up to the ceiling set with
 * hbs_ctx_set_ingest_window_max (default 1 GiB; a ceiling at or below window_bytes: no growth); h_summary->reserved[0] = the
 * window size the call ended with (0: never grown).  Past the ceiling: HBS_E_CAPACITY in h_summary->error, reserved[2] = 1
 * and reserved[1] = the stream offset of the NAL that did not fit -- everything in front of it has been delivered.  (The
 * reference's fixed 32 MiB reader parses such a NAL cut short, hevc_analyze.c:126,190-209.)  A window must not hold more than
 * window_bytes/16 NALs (HBS_E_CAPACITY otherwise).  Page-locked host buffers make the transfers asynchronous; pageable ones work.
 */
int hbs_index_extract_host(hbs_ctx* ctx, const uint8_t* h_stream, uint64_t stream_bytes, uint64_t window_bytes,
                           hbs_nal_entry* h_index, uint64_t index_cap,
                           uint8_t* h_rbsp, uint64_t rbsp_cap, hbs_summary* h_summary);
/* Device memory: a window of W bytes holds two stream buffers of 2 W, an RBSP buffer of 2 W (when h_rbsp is asked for) and an
 * index of 2 W / 32 entries -- about 8 W in all; every growth step frees them and allocates the next size, so at the default
 * ceiling a call may hold ~8 GiB on the device.  Lower the ceiling where that is too much.  When the call returns a hard error
 * (non-zero return value) h_summary still says what the runs before the failing one delivered (nal_count, rbsp_bytes). */
int hbs_ctx_set_ingest_window_max(hbs_ctx* ctx, uint64_t max_window_bytes /* 0: the default, 1 GiB */);

/*
 * K3: re-emit Annex-B from an RBSP arena: for every NAL, the bytes between the
 * previous NAL and this one (zeros and the 01 of the start code) followed by
 * rbsp_to_nal() of its RBSP (h264_nal.c:92-132: a 03 is inserted in front of
 * any byte <= 3 that follows two zeros; nothing is appended after a trailing
 * 00 00).
 *
 *   d_rbsp, rbsp_bytes                the arena and its size: nothing at or behind d_rbsp + rbsp_bytes
 *                                     is read -- an entry with rbsp_off + rbsp_len > rbsp_bytes ends the
 *                                     call with HBS_E_ARG in the summary before any byte is read through
 *                                     the index -- and the NALs must add up to at most rbsp_bytes
 *                                     (they do unless entries overlap; else HBS_E_CAPACITY)
 *   d_index_in[k].rbsp_off/rbsp_len   where NAL k's RBSP lives in d_rbsp
 *   gap_mode 0   gap of NAL k = d_index_in[k].start - d_index_in[k-1].end
 *                (start of NAL 0 for k = 0): what hbs_index_extract recorded
 *   gap_mode 1   synthetic rule: 00 00 00 01 when k % 4 == 0, else 00 00 01
 *   d_index_out  (optional) entries with start/end in the emitted stream
 *   d_summary    stream_bytes = bytes emitted; error = HBS_E_CAPACITY if
 *                out_cap was too small.  hbs_annexb_bound(rbsp_bytes, n_nals) is always enough for
 *                gap_mode 1; with gap_mode 0 the recorded gaps (zero bytes between NALs, any number
 *                of them) come on top: hbs_annexb_bound_gaps(rbsp_bytes, n_nals, gap_bytes) with
 *                gap_bytes = the sum of the gaps -- at most the `start` of the last entry of the
 *                index the gaps are taken from
 *
 * Emitting what hbs_index_extract extracted reproduces the input stream byte
 * for byte when every NAL was accepted (no HBS_ST_ERROR), none ended in
 * 00 00 03 (HBS_ST_TRAILING03: the reference drops that byte for good), the
 * bytes between NALs were zeros + 01, and the stream ends with its last NAL.
 */
int hbs_emit_annexb(hbs_ctx* ctx, const uint8_t* d_rbsp, uint64_t rbsp_bytes,
                    const hbs_nal_entry* d_index_in, uint64_t n_nals, int gap_mode,
                    uint8_t* d_out, uint64_t out_cap, hbs_nal_entry* d_index_out, hbs_summary* d_summary);
uint64_t hbs_annexb_bound(uint64_t rbsp_bytes, uint64_t n_nals);
uint64_t hbs_annexb_bound_gaps(uint64_t rbsp_bytes, uint64_t n_nals, uint64_t gap_bytes);

/*
 * K4: header parse, one NAL per lane (64 per wavefront), over the RBSP arena and index that
 * hbs_index_extract produced.  For NAL k it does what read_hevc_nal_unit()
 * does after nal_to_rbsp (hevc_stream.c:175-239): NAL header, then by type the
 * VPS / SPS / PPS / slice-segment-header reader, into structs laid out exactly
 * as hevc_stream.h's (include/hevc_stream.h).  The state the reference threads
 * through one mutable parser object is resolved per NAL: a slice is read
 * against the last SPS and PPS that precede it in the stream.
 *
 *   d_parsed[k]     rc (read_hevc_nal_unit's return value: consumed NAL bytes,
 *                   or -1: bad emulation pattern, unsupported type -- AUD, SEI,
 *                   EOS, ... -- or bit-reader overrun), the hevc_nal_t fields
 *                   (-1 when nal_to_rbsp already failed), where its struct is in
 *                   d_structs, and for slices h->slice_data: payload size and
 *                   where the payload starts inside the NAL's RBSP
 *   d_structs       struct arena: hevc_vps_t / hevc_sps_t / hevc_pps_t /
 *                   hevc_slice_header_t per NAL at d_parsed[k].struct_off
 *                   (an SPS slot is followed by its derived RPS tables);
 *                   NULL = plan only
 *   d_summary       reserved[0] = arena bytes needed; error HBS_E_CAPACITY when
 *                   structs_cap was smaller (NALs that did not fit keep
 *                   struct_off = ~0)
 */
typedef struct hbs_parsed_nal {
    int32_t  rc;
    int32_t  nal_unit_type, nal_layer_id, nal_temporal_id_plus1;
    uint64_t struct_off;
    int32_t  slice_data_size;
    uint32_t slice_data_off;
} hbs_parsed_nal;

int hbs_parse_headers(hbs_ctx* ctx, const uint8_t* d_rbsp, const hbs_nal_entry* d_index, uint64_t n_nals,
                      hbs_parsed_nal* d_parsed, uint8_t* d_structs, uint64_t structs_cap, hbs_summary* d_summary);
/*
 * ---- compact header parse (round 5) -------------------------------------------------------------------------------
 * hbs_parse_headers writes one hevc_slice_header_t per slice: 4 024 bytes, cleared and then filled (reference struct
 * hevc_stream.h:465-515, reader hevc_stream.c:782-941) -- 405 MB for the 100 k NALs of a 4K30 stream, of which a caller that
 * indexes a stream reads a handful of members.  hbs_parse_headers_compact walks every slice header exactly as
 * hbs_parse_headers does (same bits, same order, same derived tables, out-of-spec slices walked again exactly) but WITHOUT a
 * struct: per slice a 64-byte hbs_slice_compact -- sixteen members of hevc_slice_header_t, each equal to the member of the
 * same name in the struct hbs_parse_headers fills -- next to the same hbs_parsed_nal (rc, NAL header, slice_data_off /
 * slice_data_size; struct_off = ~0 for slices).  Parameter sets are parsed into d_structs as always (the slices need them;
 * a VPS + SPS + PPS group is ~0.5 MB).  d_structs = NULL: plan only (d_summary->reserved[0] = arena bytes needed).
 *
 * hbs_parse_materialize is the same call with a list of NAL numbers (device memory, any order): the listed NALs that are
 * slices are walked into full hevc_slice_header_t slots in d_structs behind the parameter sets (d_parsed[k].struct_off says
 * where), every other slice into its compact record as before.  "The full struct on demand": call it with the NALs a caller
 * wants to look at closely; the structs are those of hbs_parse_headers, member for member.
 *
 * No trace and no parser-state output in these calls.  Errors in d_summary->error: HBS_E_CAPACITY (structs_cap), HBS_E_DEPTH
 * (an out-of-spec stream whose exact answer needs the sequential parse: a chain of slice-own RPS sets deeper than three,
 * never seen in 12 000 fuzzed streams; hbs_parse_headers handles it).
 */
typedef struct hbs_slice_compact {
    int32_t first_slice_segment_in_pic_flag, no_output_of_prior_pics_flag, pic_parameter_set_id, dependent_slice_segment_flag;
    int32_t slice_segment_address, slice_type, pic_output_flag, slice_pic_order_cnt_lsb;
    int32_t short_term_ref_pic_set_sps_flag, short_term_ref_pic_set_idx, num_long_term_pics, slice_temporal_mvp_enabled_flag;
    int32_t num_ref_idx_l0_active_minus1, num_ref_idx_l1_active_minus1, slice_qp_delta, num_entry_point_offsets;
} hbs_slice_compact;

int hbs_parse_headers_compact(hbs_ctx* ctx, const uint8_t* d_rbsp, const hbs_nal_entry* d_index, uint64_t n_nals,
                              hbs_parsed_nal* d_parsed, hbs_slice_compact* d_compact, uint8_t* d_structs, uint64_t structs_cap,
                              const uint8_t* d_initial_sps_slot, const uint8_t* d_initial_pps, hbs_summary* d_summary);
int hbs_parse_materialize(hbs_ctx* ctx, const uint8_t* d_rbsp, const hbs_nal_entry* d_index, uint64_t n_nals,
                          hbs_parsed_nal* d_parsed, hbs_slice_compact* d_compact, uint8_t* d_structs, uint64_t structs_cap,
                          const uint8_t* d_initial_sps_slot, const uint8_t* d_initial_pps,
                          const uint64_t* d_nal_list, uint64_t n_list, hbs_summary* d_summary);

/*
 * Opt-in extension: the NAL types read_hevc_nal_unit() returns -1 for without reading them
 * (hevc_stream.c:221-222) -- access unit delimiter 35, end of sequence 36, end of bitstream 37, filler data 38,
 * prefix / suffix SEI 39 / 40 -- read the way the reference's own, never dispatched readers would
 * (read_hevc_access_unit_delimiter_rbsp hevc_stream.c:573-577, read_filler_data_rbsp :590-597, the SEI message loop
 * :524-563 with h264_sei.c's opaque payloads).  Call it behind hbs_parse_headers on the same arrays: for every NAL of
 * those types d_parsed[k].rc becomes the bytes consumed (or -1 when the cursor ran past the RBSP, as :225 does) and
 * d_ext[k] is filled; other NALs' records are zeroed and their d_parsed entries left alone.  hbs_parse_headers by
 * itself keeps the reference's -1.
 */
#define HBS_SEI_MAX_MESSAGES 6
typedef struct hbs_sei_message {
    int32_t  payloadType, payloadSize;     /* sei_t, h264_sei.h:38-47 */
    uint32_t payload_off;                  /* where the payload bytes start inside the NAL's RBSP (not copied) */
    uint32_t reserved;
} hbs_sei_message;
typedef struct hbs_ext_nal {
    int32_t  num_sei_messages;             /* messages the NAL holds; the first HBS_SEI_MAX_MESSAGES are recorded */
    int32_t  primary_pic_type;             /* hevc_aud_t */
    uint32_t filler_bytes;                 /* ff_byte count of a filler data NAL */
    uint32_t reserved;
    hbs_sei_message sei[HBS_SEI_MAX_MESSAGES];
} hbs_ext_nal;
int hbs_parse_extended(hbs_ctx* ctx, const uint8_t* d_rbsp, const hbs_nal_entry* d_index, uint64_t n_nals,
                       hbs_parsed_nal* d_parsed, hbs_ext_nal* d_ext);

/*
 * hbs_index_parse: BASELINE config 3 without an RBSP arena -- find_nal_unit over the whole stream (the index-only scan:
 * start / end / rbsp_off / rbsp_len / status exactly as hbs_index_extract with d_rbsp = NULL), then read_hevc_nal_unit's
 * parse of every NAL, reading each NAL's header straight from the stream: only the bytes the readers can look at are
 * stripped of their emulation prevention bytes (nal_to_rbsp's rule, hevc_stream.c:161-179 / h264_nal.c:147-200) into a
 * small window per NAL -- `header_window` RBSP bytes of a slice segment (0 = 512; 64 ... 65536, a multiple of 16), all of
 * a VPS / SPS / PPS -- instead of the whole stream into an arena.  d_parsed and d_structs come out exactly as from
 * hbs_index_extract + hbs_parse_headers (slice_data_off / slice_data_size still describe the NAL's RBSP, which is not
 * materialised); d_payload_off (optional, one uint64 per NAL) receives the STREAM offset of the first payload byte of
 * every parsed slice (~0 for other NALs).  A slice header that does not end at least 8 bytes inside its window -- hundreds
 * of entry points -- is reported, never guessed: d_parse_summary->error = HBS_E_CAPACITY and that NAL's rc = INT32_MIN;
 * call again with a larger window, or take the arena path.  The call waits once, for the scan's NAL count (returned in
 * *nal_count_out when not NULL); the parse is enqueued behind it.  On the 2.1 GiB 4K30 sequence of bench.py: scan 1 B/B
 * + ~3 % for the windows, against 2 B/B for the arena.
 */
int hbs_index_parse(hbs_ctx* ctx, const uint8_t* d_stream, uint64_t stream_bytes,
                    hbs_nal_entry* d_index, uint64_t index_cap, uint32_t header_window,
                    hbs_parsed_nal* d_parsed, uint8_t* d_structs, uint64_t structs_cap, uint64_t* d_payload_off,
                    hbs_summary* d_scan_summary, hbs_summary* d_parse_summary, uint64_t* nal_count_out);
/* the same with the compact parse behind the scan (hbs_parse_headers_compact): d_compact[k] for every NAL, no slice structs */
int hbs_index_parse_compact(hbs_ctx* ctx, const uint8_t* d_stream, uint64_t stream_bytes,
                    hbs_nal_entry* d_index, uint64_t index_cap, uint32_t header_window,
                    hbs_parsed_nal* d_parsed, hbs_slice_compact* d_compact, uint8_t* d_structs, uint64_t structs_cap, uint64_t* d_payload_off,
                    hbs_summary* d_scan_summary, hbs_summary* d_parse_summary, uint64_t* nal_count_out);

/* Same, for a batch that continues an earlier one: d_initial_sps_slot (an SPS
 * slot = hevc_sps_t followed at hbs_sps_tables_offset() by its derived RPS
 * tables, hbs_sps_slot_bytes() in all) and d_initial_pps (hevc_pps_t) are the
 * parameter sets in force before NAL 0; NULL = none parsed yet. */
int hbs_parse_headers_ctx(hbs_ctx* ctx, const uint8_t* d_rbsp, const hbs_nal_entry* d_index, uint64_t n_nals,
                          hbs_parsed_nal* d_parsed, uint8_t* d_structs, uint64_t structs_cap,
                          const uint8_t* d_initial_sps_slot, const uint8_t* d_initial_pps, hbs_summary* d_summary);

/* Same, and additionally the per-field trace the reference's read_debug_* readers print
 * (hevc_stream.c:2343-3434): for NAL k, d_trace[k * trace_cap ...] receives one record per syntax
 * element read behind the NAL header, in reading order; d_trace_count[k] = how many it produced
 * (records beyond trace_cap are counted, not stored).  `site` identifies the syntax element (the key
 * of the name table the legacy read_debug_hevc_nal_unit prints with), `pos` the bit position of
 * the RBSP cursor before the read, `value` what was read. */
typedef struct hbs_trace_rec { uint32_t site; uint32_t pos; int32_t value; } hbs_trace_rec;
int hbs_parse_headers_trace(hbs_ctx* ctx, const uint8_t* d_rbsp, const hbs_nal_entry* d_index, uint64_t n_nals,
                            hbs_parsed_nal* d_parsed, uint8_t* d_structs, uint64_t structs_cap,
                            const uint8_t* d_initial_sps_slot, const uint8_t* d_initial_pps,
                            hbs_trace_rec* d_trace, uint32_t trace_cap, uint32_t* d_trace_count, hbs_summary* d_summary);
uint64_t hbs_sps_slot_bytes(void);
/* The same, and what the reference would hold BEHIND the last NAL of the batch, for a caller that goes on NAL by NAL or with
 * another batch (the legacy symbols do: they serve the loop of hevc_analyze.c:135-177 from one batch per buffer): the SPS in
 * force with the 32 rows of the derived RPS tables (hevc_stream.c:26-32) -- each row what the last NAL that wrote it left,
 * also rows beyond the SPS's own sets -- into d_state_sps_slot (hbs_sps_slot_bytes()), and the PPS in force into d_state_pps
 * (sizeof(hevc_pps_t)).  Either may be the buffer the initial context came from.  Both NULL: hbs_parse_headers_trace.
 * summary.reserved[1] != 0: a row depends on a chain of more than three slices' own sets; that row was left untouched.
 * Needs d_structs and n_nals >= 1. */
int hbs_parse_headers_state(hbs_ctx* ctx, const uint8_t* d_rbsp, const hbs_nal_entry* d_index, uint64_t n_nals,
                            hbs_parsed_nal* d_parsed, uint8_t* d_structs, uint64_t structs_cap,
                            const uint8_t* d_initial_sps_slot, const uint8_t* d_initial_pps,
                            hbs_trace_rec* d_trace, uint32_t trace_cap, uint32_t* d_trace_count, hbs_summary* d_summary,
                            uint8_t* d_state_sps_slot, uint8_t* d_state_pps);
/* One NAL at a time, exactly as the reference does it (what the legacy symbols use): with this on, a call with
 * n_nals == 1 and a d_initial_sps_slot treats the RPS tables behind that SPS as THE tables (hevc_stream.c:26-32):
 * an SPS writes its rows into them and leaves the others, a slice's own set lands in them, and a slice that names a
 * row nobody of its SPS wrote reads what is there.  The slot is then read AND written by the call.  Batches
 * (n_nals > 1) parse their NALs independently of one another either way. */
int hbs_ctx_set_sequential_parse(hbs_ctx* ctx, int on);
uint64_t hbs_sps_tables_offset(void);
/* How hbs_emit_annexb works: -1 (default) picked per call -- one single-workgroup launch for a handful of small
 * NALs (<= 256 NALs, <= 32 KiB of RBSP: the legacy rbsp_to_nal); otherwise a single pass: by ARENA TILES when the
 * index's NALs lie back to back in the arena in index order (what hbs_index_extract and hbs_write_headers produce;
 * checked on the device together with a few size limits, hbs_emit.hip: k3t_check; a tile that turns out dense in zero
 * pairs -- padding, cabac_zero_words -- is walked by rows by its own workgroup), else by NALs (items of <= 12 KiB);
 * or, on zero-heavy payload (density probe on the device), count / scan / emit; arenas whose mean NAL is below 448
 * bytes: the arena tiles when they apply (up to 1024 NAL starts per 192 KiB), else a lane per NAL.  0 pins the single pass by NALs,
 * 1 the three steps, 2 the arena tiles whenever the index allows them (whatever the arena's size and density).
 * The bytes are the same whichever runs (h264_nal.c:92-132). */
int hbs_ctx_set_emit_path(hbs_ctx* ctx, int path);
/* Diagnostic: 1 when the arena-tile kernel did the whole of the last hbs_emit_annexb on this context (the index was eligible
 * and no tile was handed to another kernel), 0 when another path did (waits for the call to finish). */
int hbs_ctx_last_emit_by_tiles(hbs_ctx* ctx);

/*
 * Synthetic workload S(seed, n_nals, mode) of SURVEY.md 8(d), generated in HBM:
 * RBSP of NAL k is 8192 + mix(k) % 4097 pseudo-random bytes (mode 0 uniform,
 * mode 1 "zero-heavy": ~10 % 00 and ~5 % 01..03), first bytes 02 01, last byte
 * 80.  Fills d_rbsp (packed) and d_index[k].rbsp_off/rbsp_len; follow with
 * hbs_emit_annexb(..., gap_mode 1, ...) to obtain the Annex-B stream.
 * d_summary->stream_bytes receives the RBSP bytes written.
 */
int hbs_synth_rbsp(hbs_ctx* ctx, uint64_t seed, uint64_t n_nals, int mode,
                   uint8_t* d_rbsp, uint64_t rbsp_cap, hbs_nal_entry* d_index, hbs_summary* d_summary);
uint64_t hbs_synth_rbsp_bound(uint64_t n_nals);

/*
 * K5: the syntax writers behind write_hevc_nal_unit (hevc_stream.c:1249-1327): NAL k's struct
 * (at d_structs + d_parsed[k].struct_off, of the type d_parsed[k] names; layout as hbs_parse_headers
 * leaves it, an SPS followed by its derived tables) is serialised into d_rbsp_out + k * rbsp_cap
 * (rbsp_cap bytes per NAL, zero-filled first; the reference uses size * 3 / 4 of the caller's buffer).
 * Slices are written against the last SPS / PPS in front of them in the batch (or the initial ones).
 * d_written[k]: rc 0 / -1 (unsupported type, or wrote past rbsp_cap), the whole bytes written, and
 * what the reference's writer leaves in h->slice_data->rbsp_size.  hbs_emit_annexb turns the RBSP
 * into NAL bytes (rbsp_to_nal).  Quirks of the reference's writers are kept: see hbs_parse.h.
 */
typedef struct hbs_written_nal { int32_t rc; uint32_t rbsp_size; int32_t slice_data_size; uint32_t pad; } hbs_written_nal;
int hbs_write_headers(hbs_ctx* ctx, const hbs_parsed_nal* d_parsed, uint64_t n_nals, uint8_t* d_structs,
                      const uint8_t* d_initial_sps_slot, const uint8_t* d_initial_pps,
                      uint8_t* d_rbsp_out, uint32_t rbsp_cap, hbs_written_nal* d_written);

/*
 * ---- Several GPUs: one process per GPU, one context each ------------------------------------------
 * Every rank indexes its own bytes; the one exchange of the path is the gather of the NAL index: the counts (8 bytes per
 * rank) to everybody, then exactly count x 32 bytes per rank, to `root` or (root = -1) to every rank, over RCCL (xGMI inside
 * a node).  Stream bytes and RBSP arenas never travel.  RCCL is looked up at run time (librccl.so.1; the copy the process
 * already carries, if any), so the library has no link-time dependency on it.  Environment HBS_RCCL_LIB=<path> (read once, when
 * the first communicator is made) names the library to take RCCL's entry points from instead -- a site's own RCCL build; this
 * project's tests point it at a shared-memory stand-in to run world > 1 on one GPU.  A path that does not load is an error.
 *
 *   hbs_comm_unique_id   rank 0 makes the 128-byte id; the caller hands it to every rank by its own means (MPI, a file, a socket,
 *                        torch.distributed ...)
 *   hbs_comm_create      every rank, collectively: the communicator (ncclCommInitRank)
 *   hbs_comm_adopt       instead: wrap an ncclComm_t the application already owns (same RCCL copy); not destroyed by hbs_comm_destroy
 *   hbs_gather_index     collective, on the context's stream; returns when the sizes are known (one wait for the 8-byte counts),
 *                        the payload then moves asynchronously on that stream.  d_index: this rank's n_local entries;
 *                        stream_base / rbsp_base: added to start, end / rbsp_off of this rank's entries on the way out (0 for
 *                        independent streams; the part's cut offset for parts of ONE stream); d_all (cap_all entries; receivers
 *                        only): the ranks' entries back to back in rank order; counts_out[world] (host): entries per rank.
 *                        Errors are collective: every rank first learns every rank's count, the capacity of every receiver
 *                        and whether its local preparations succeeded (32 bytes per rank, one all-gather), and all take the
 *                        same decision -- HBS_E_CAPACITY on EVERY rank if the entries do not fit some receiver's d_all,
 *                        HBS_E_HIP on every rank if one of them failed -- before any payload call is posted, so that no rank
 *                        is left inside a collective the others never enter.  world <= 1024.
 *   hbs_gather_parts     the same for parts of ONE stream (below): `stopped` = this part's scan ended at an empty NAL
 *                        (hbs_summary.stop_reason == 1).  The reference's loop over the whole stream ends there
 *                        (hevc_analyze.c:135), so the parts behind the first one that stopped contribute no entries
 *                        (counts_out says 0 for them) and the gathered index is the whole stream's.
 */
#define HBS_COMM_ID_BYTES 128
typedef struct hbs_comm hbs_comm;
int  hbs_comm_unique_id(uint8_t id[HBS_COMM_ID_BYTES]);
int  hbs_comm_create(hbs_ctx* ctx, const uint8_t id[HBS_COMM_ID_BYTES], int rank, int world, hbs_comm** out);
int  hbs_comm_adopt(hbs_ctx* ctx, void* nccl_comm, int rank, int world, hbs_comm** out);
void hbs_comm_destroy(hbs_comm* comm);
int  hbs_comm_rank(const hbs_comm* comm);
int  hbs_comm_world(const hbs_comm* comm);
/* how many workgroup slots hbs_ctx_reserve_workgroups should leave free for this communicator's exchange to run beside the
 * next scan: 8 per peer, between 32 and 64 (of 512) */
int  hbs_comm_reserve_hint(const hbs_comm* comm);
int  hbs_gather_index(hbs_ctx* ctx, hbs_comm* comm, const hbs_nal_entry* d_index, uint64_t n_local,
                      uint64_t stream_base, uint64_t rbsp_base, int root,
                      hbs_nal_entry* d_all, uint64_t cap_all, uint64_t* counts_out);
int  hbs_gather_parts(hbs_ctx* ctx, hbs_comm* comm, const hbs_nal_entry* d_index, uint64_t n_local, int stopped,
                      uint64_t stream_base, uint64_t rbsp_base, int root,
                      hbs_nal_entry* d_all, uint64_t cap_all, uint64_t* counts_out);
int  hbs_ctx_device(hbs_ctx* ctx);
/*
 * ONE stream over several GPUs.  A part begins at the first start code (00 00 01) at or after its nominal boundary:
 * hbs_find_cut_host(bytes, n, from) returns that offset in a HOST buffer (plain C, no GPU involved; ~0: none with 8 bytes behind
 * it -- the part in front then runs to the end).  Both neighbours of a boundary apply the same rule to the same bytes, so they
 * agree without a collective.  A rank uploads its part [cut_r, cut_r+1) FOLLOWED BY the next 8 bytes of the stream (they
 * terminate its last NAL as they do in the whole stream, h264_nal.c:64-72), runs hbs_index_extract on that, and drops the NAL the
 * halo opens with hbs_trim_part(part_bytes = cut_r+1 - cut_r): n_kept entries and rbsp_kept arena bytes are the part's.  The last
 * part has no halo and nothing to trim.  hbs_gather_parts(stopped = (stop_reason == 1), stream_base = cut_r, rbsp_base = RBSP bytes
 * of the parts in front or 0) then yields the whole stream's index, empty NALs in the stream included; RBSP arenas stay where they are.
 * (A part whose scan stopped at an empty NAL has nothing to trim either: its walk never reached the halo.)
 */
uint64_t hbs_find_cut_host(const uint8_t* bytes, uint64_t n, uint64_t from);
int  hbs_trim_part(hbs_ctx* ctx, const hbs_nal_entry* d_index, uint64_t nal_count, uint64_t rbsp_bytes, uint64_t part_bytes,
                   uint64_t* n_kept, uint64_t* rbsp_kept);

/* Device-memory helpers for callers without HIP headers (the legacy C layer):
 * allocate / free on the context's GPU, synchronising copies, async fill. */
int hbs_dev_alloc(hbs_ctx* ctx, uint64_t bytes, void** out);
int hbs_dev_free(hbs_ctx* ctx, void* p);
int hbs_copy_to_device(hbs_ctx* ctx, void* d_dst, const void* h_src, uint64_t bytes);
int hbs_copy_to_host(hbs_ctx* ctx, void* h_dst, const void* d_src, uint64_t bytes);
int hbs_fill_device(hbs_ctx* ctx, void* d_dst, int value, uint64_t bytes);
/* page-locked host memory, a host-to-device copy that does not wait (source must be page-locked and
 * stay untouched until the stream has passed it), device-to-device copy on the context's stream */
int hbs_host_alloc(hbs_ctx* ctx, uint64_t bytes, void** out);
int hbs_host_free(hbs_ctx* ctx, void* p);
int hbs_copy_to_device_async(hbs_ctx* ctx, void* d_dst, const void* h_src, uint64_t bytes);
int hbs_copy_device(hbs_ctx* ctx, void* d_dst, const void* d_src, uint64_t bytes);

/*
 * Output buffers placed against the input they are written from.
 *
 * On MI355X a kernel that reads one large buffer and writes another in long bursts (hbs_index_extract: stream -> RBSP arena;
 * hbs_emit_annexb: arena -> stream) runs ~4-5 % slower when both buffers lie in the same one of two classes of physical
 * memory (16 GiB: 6.20 against 5.90 ms; DESIGN.md section 3, profiles/r04/placement_*.txt) -- decided when the buffers are
 * allocated, the same for every offset inside them, and invisible to HIP.  hbs_pair_alloc returns `bytes` of device memory
 * whose placement against `d_peer` (16-byte aligned, its final size and location; contents do not matter and are not changed)
 * has been MEASURED.  Buffers of 1 GiB and more against peers of 512 MiB and more are put together from 1 GiB physical
 * chunks (hipMemCreate / hipMemMap) of a per-device POOL:
 *   - every chunk the pool creates is classed once, by two content-free copies with the kernels' access pattern (half a GiB
 *     each, ~1 ms, on the context's stream) against the pool's reference chunk and against itself;
 *   - each GiB piece of the peer is classed the same way (one copy; by table lookup when the peer is itself such a buffer),
 *     and chunk k of the buffer is one of the class its peer piece is NOT in;
 *   - hbs_pair_free unmaps the buffer and puts its chunks back on the pool's free list, class attached; chunks of the class
 *     nobody wanted stay there too.  A later hbs_pair_alloc takes them: no probes but the peer's, typically 16-20 ms for
 *     16 GiB where the first call of a process takes 0.05-5 s (it has to find memory of both classes: up to `chunks + 88` new
 *     chunks and 128 GiB of unmapped ballast to skip runs of one class, released when the call ends, 24 GiB of the device
 *     left free throughout).
 *   - the free list is bounded: what exceeds HBS_PAIR_POOL_KEEP_GIB (default 48) GiB when a call ends goes back to the driver,
 *     hbs_pair_pool_trim(ctx, keep_bytes) releases on demand, hbs_pair_pool_stats reports.
 * What a caller should know about such a buffer: the size is rounded up to whole GiB; it is virtual-memory-API memory -- no
 * hipFree (hbs_pair_free only), no HIP IPC handle; its addresses come from one 32 TiB reservation per process that is used
 * front to back and never again (a range that had been unmapped and mapped again served stale physical memory on this stack):
 * ~17 GiB of addresses per 16 GiB buffer, about 1 900 such allocations per process, after which hbs_pair_alloc silently takes
 * the plain way below.  One hbs_pair_alloc at a time per process (they share the pool, the address reservation and the probes).
 * Smaller buffers, smaller peers, d_peer == NULL, HBS_PAIR_PLAIN=1, or when the virtual-memory calls fail: ordinary hipMalloc
 * memory -- with a peer to measure against, up to six whole allocations are tried and the one with the most fast pieces kept.
 * hbs_pair_free may be called from any thread (a destructor, a garbage collector): it leaves the caller's current device as it
 * found it and waits only for the context's stream (the whole device when ctx is NULL).
 * Plain hipMalloc / torch buffers keep working everywhere; they land in the slow mode about every other time.  No reference
 * counterpart (the reference's buffers are malloc'ed host memory, hevc_analyze.c:100-103).
 */
typedef struct hbs_pair_report {
    uint32_t chunks;                 /* GiB pieces of the buffer (the last one may be partial)                         */
    uint32_t probed;                 /* probe measurements of this call: peer pieces and NEW chunks                    */
    uint32_t rejected;               /* new chunks of a class nobody wanted (they stay on the pool's free list)        */
    uint32_t accepted_fast;          /* pieces of the returned buffer known to pair fast with their peer piece         */
    uint32_t unprobed_after_budget;  /* pieces of the returned buffer that do not (or were not measured)               */
    uint32_t from_pool;              /* pieces that came off the pool's free list (classed by an earlier call)         */
    uint32_t from_table;             /* peer pieces classed by lookup (the peer is itself a buffer of the pool)        */
    uint32_t reserved;
} hbs_pair_report;
int hbs_pair_alloc(hbs_ctx* ctx, const void* d_peer, uint64_t peer_bytes, uint64_t bytes, void** out, hbs_pair_report* report /* may be NULL */);
int hbs_pair_free(hbs_ctx* ctx, void* ptr);
uint64_t hbs_pair_pool_trim(hbs_ctx* ctx, uint64_t keep_bytes);
int hbs_pair_pool_stats(hbs_ctx* ctx, uint64_t out[4] /* chunks created, chunks classed, free chunks of class 0, of class 1 */);

/* Synchronising copy of a device hbs_summary to the host. */
int hbs_read_summary(hbs_ctx* ctx, const hbs_summary* d_summary, hbs_summary* h_summary);

/* Upper bound of the scratch the context will hold for a stream this long
 * (look-back descriptors; allocated lazily, reused between calls). */
uint64_t hbs_workspace_bytes(uint64_t stream_bytes);

/* Device memory the context itself holds right now (look-back descriptors, run header, padded last tile, the K3 / K4 /
 * index-only workspace, the header windows of hbs_index_parse): grow-only scratch, sized by what the calls so far needed --
 * by the NALs FOUND, never by an index capacity.  Caller-owned buffers (stream, index, arena, structs) are not counted. */
uint64_t hbs_ctx_device_bytes(hbs_ctx* ctx);

/* Legacy single-NAL symbols (h264_stream.h / hevc_stream.h) only.  The derived short-term RPS tables are process state in
 * the reference (file-static, zero at program start, hevc_stream.c:26-32) and device state here; this puts them back to
 * "program start", which the reference can only do by starting a new process.  Not part of the reference's API. */
void hbs_legacy_reset_tables(void);
/* Legacy symbols only, a measuring / testing aid: how the loop of a caller was answered so far -- out[0] batches built (one
 * upload + index + extraction + parse of up to 64 MiB each), [1] reads answered from a batch, [2] reads answered one call at a
 * time, [3] batch builds suppressed by the back-off (a batch that is dropped before the calls answered from it were worth its
 * cost makes the next 1, 2, 4, ... find_nal_unit calls of large buffers go without one; hbs_legacy.c). */
void hbs_legacy_batch_stats(uint64_t out[4]);

#ifdef __cplusplus
}
#endif
#endif
