/*
 * hbs_legacy.c -- the reference's single-NAL C API as thin host wrappers over
 * the batch C ABI (include/hevcbitstream_amd.h).  Plain C: it only moves the
 * caller's buffers to and from the GPU and calls hbs_*; every byte is scanned,
 * stripped, inserted or parsed by the HIP kernels.  No CPU implementation of
 * the algorithms exists in this file, and without a gfx950 GPU every call returns
 * its failure value.  A call by itself is launch and copy latency, so a caller's
 * loop over a buffer is answered from ONE batch per buffer ("one batch per buffer"
 * below); calls outside such a loop keep to one upload (page-locked staging), the
 * kernels back to back, and one download of a result block (see need_ctx).
 *
 * Symbols and the reference interface each one replaces:
 *   find_nal_unit               h264_nal.c:38-76       (proto h264_stream.h:54)
 *   nal_to_rbsp                 h264_nal.c:147-200     (proto h264_stream.h:57)
 *   rbsp_to_nal                 h264_nal.c:92-132      (proto h264_stream.h:56)
 *   hevc_new / hevc_free        hevc_nal.c:34-57, :64-91
 *   peek_hevc_nal_unit          hevc_nal.c:97-114
 *   read_hevc_nal_unit          hevc_stream.c:155-240
 *   read_debug_hevc_nal_unit    hevc_stream.c:2343-3434 (per-field trace from the GPU parser's log)
 *   write_hevc_nal_unit         hevc_stream.c:1249-1333 (the GPU parser's walk in write mode, then rbsp_to_nal)
 *   debug_bytes, h264_dbgfile   h264_stream.c:33, :117-126
 *
 * Threads.  In the reference find_nal_unit / nal_to_rbsp / rbsp_to_nal are pure and may be called from any
 * number of threads; here they share one GPU context, its staging buffers and (find_nal_unit) the cache of the last
 * scan's answer -- so every entry point below takes ONE process-wide lock for its duration.  Calls from several
 * threads are safe and serialised (tests/test_gpu_legacy.py runs two threads against each other).  What stays process
 * state, as in the reference (hevc_stream.c:26-32 keeps it in file-static arrays, on which its read_* of DIFFERENT
 * hevc_stream_t objects race): the derived RPS tables of the last SPS read -- here a device buffer, so two parsers used
 * in turn see each other's tables exactly as they do with the reference library.
 * Without a gfx950 GPU the first call prints a diagnostic once and every call returns its failure value (0 from
 * find_nal_unit, -1 from the others) instead of working on the CPU; failures in the middle of a call (device memory
 * exhausted, a lost GPU) still abort().
 */
#define _GNU_SOURCE             /* PTHREAD_MUTEX_RECURSIVE under -std=c99 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <time.h>

#include "../../include/hevcbitstream_amd.h"
#include "../../include/h264_stream.h"
#include "../../include/hevc_stream.h"

FILE* h264_dbgfile = NULL;

/* one lock for everything below (recursive: write_hevc_nal_unit ends in the emit path rbsp_to_nal uses) */
static pthread_mutex_t g_lock;
static pthread_once_t g_lock_once = PTHREAD_ONCE_INIT;
static void lock_init(void)
{
    pthread_mutexattr_t a;
    pthread_mutexattr_init(&a);
    pthread_mutexattr_settype(&a, PTHREAD_MUTEX_RECURSIVE);
    pthread_mutex_init(&g_lock, &a);
    pthread_mutexattr_destroy(&a);
}
static void legacy_lock(void) { pthread_once(&g_lock_once, lock_init); pthread_mutex_lock(&g_lock); }
static void legacy_unlock(void) { pthread_mutex_unlock(&g_lock); }

/* ---- one lazily created GPU context and its scratch ---------------------------------- */

static hbs_ctx* g_ctx = NULL;
static uint8_t* g_dbuf = NULL;       /* input bytes                       */
static uint64_t g_dbuf_cap = 0;
static uint8_t* g_dout = NULL;       /* RBSP arena / emitted stream       */
static uint64_t g_dout_cap = 0;
/* one device block, so that a call's results come back in ONE copy:
 *   [0, 64) summary | [64, 192) 4 parsed records | [192, 256) the parse's summary | [256, 384) 4 index entries |
 *   [512, 512 + R) the NAL's RBSP (R follows the NAL at hand) | [512 + R, ...) one struct slot (VPS-sized)
 * a slice's header struct and its payload are then one copy of 512 + R + 4 KiB away */
static uint8_t* g_dblock = NULL;
static hbs_nal_entry* g_dindex = NULL;
static hbs_summary* g_dsummary = NULL;
static hbs_parsed_nal* g_dparsed = NULL;
static uint8_t* g_dstruct = NULL;
static uint8_t* g_dsps_slot = NULL;  /* SPS in force + its RPS tables     */
static uint8_t* g_dpps = NULL;       /* PPS in force                      */
#define LEGACY_INDEX_CAP 4
#define FIND_INDEX_CAP 64
#define RES_SUMMARY 0
#define RES_PARSED 64
#define RES_SUMMARY2 192                                   /* the parse's own summary: the scan's stays readable */
#define RES_INDEX 256
#define RES_SMALL (RES_INDEX + LEGACY_INDEX_CAP * 32)      /* summary + parsed + index */
#define RES_RBSP 512
#define RES_FIND_BYTES (64 + FIND_INDEX_CAP * 32)          /* find_nal_unit's own block: summary, then FIND_INDEX_CAP entries */
#define SLICE_SLOT ((sizeof(hevc_slice_header_t) + 15) & ~(size_t)15)
static uint64_t g_dblock_cap = 0;
static uint64_t g_res_r = 0;                               /* R of the call at hand */
static uint8_t* g_drbsp = NULL;                            /* g_dblock + RES_RBSP */
static uint8_t* g_hres = NULL;                             /* host mirror of the front of the block */
static uint8_t* g_dfind = NULL;

/* find_nal_unit scans a prefix of the caller's buffer and finds several NALs in it; the canonical loop
 * (hevc_analyze.c:135-177) asks for them one after the other.  The rest of the last scan's answer is kept,
 * with a copy of the bytes it was computed from, and served when the next call starts exactly at the previous
 * NAL's end and the bytes up to the answer's terminator are still the same (memcmp): same bytes, same walk. */
#define FIND_CACHE_MAX (1u << 20)
static const uint8_t* g_fc_base = NULL;
static uint8_t* g_fc_copy = NULL;
static uint64_t g_fc_len = 0;
static hbs_nal_entry g_fc_ent[FIND_INDEX_CAP];
static uint8_t g_hfind[RES_FIND_BYTES];
static uint64_t g_fc_n = 0, g_fc_next = 0;
/* page-locked staging for uploads: [00 00 01 | NAL bytes] goes up in one copy that nobody waits for */
static uint8_t* g_hstage = NULL;
static uint64_t g_hstage_cap = 0;
/* what the device copies of the parameter sets in force were last loaded from (h->sps / h->pps are the
 * caller's to edit between calls: they are compared, and uploaded again only when they differ) */
static uint8_t* g_sps_shadow = NULL;
static uint8_t* g_pps_shadow = NULL;
static int g_sps_shadow_ok = 0, g_pps_shadow_ok = 0;

static void die(const char* what, int rc)
{
    fprintf(stderr, "libhevcbitstream (MI355X build): %s failed (%d%s%s). This library has no CPU path: "
                    "it needs a gfx950 GPU.\n", what, rc, g_ctx ? ": " : "", g_ctx ? hbs_last_error(g_ctx) : "");
    abort();
}

static struct timespec g_t_ready;
static void report_time(void)
{
    struct timespec t1;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    fprintf(stderr, "libhevcbitstream: %.4f s between the context being ready and exit\n",
            (double)(t1.tv_sec - g_t_ready.tv_sec) + 1e-9 * (double)(t1.tv_nsec - g_t_ready.tv_nsec));
}
static int g_no_gpu = 0;
/* 0, or -1: there is no gfx950 GPU to run on (said once; the callers return their failure value) */
static int need_ctx(void)
{
    int rc;
    const char* dev = getenv("HBS_DEVICE");
    if (g_ctx) return 0;
    if (g_no_gpu) return -1;
    rc = hbs_ctx_create(&g_ctx, dev ? atoi(dev) : 0);
    if (rc) {
        g_ctx = NULL; g_no_gpu = 1;
        fprintf(stderr, "libhevcbitstream (MI355X build): hbs_ctx_create failed (%d). This library has no CPU path: it needs a gfx950 GPU; "
                        "every call will fail.\n", rc);
        return -1;
    }
    /* one NAL after the other, one set of derived RPS tables for the process: the reference's semantics to the letter */
    if ((rc = hbs_ctx_set_sequential_parse(g_ctx, 1))) die("hbs_ctx_set_sequential_parse", rc);
    if ((rc = hbs_dev_alloc(g_ctx, RES_FIND_BYTES, (void**)&g_dfind))) die("hbs_dev_alloc", rc);
    if ((rc = hbs_dev_alloc(g_ctx, hbs_sps_slot_bytes(), (void**)&g_dsps_slot))) die("hbs_dev_alloc", rc);
    if ((rc = hbs_dev_alloc(g_ctx, sizeof(hevc_pps_t) + 64, (void**)&g_dpps))) die("hbs_dev_alloc", rc);
    if ((rc = hbs_fill_device(g_ctx, g_dsps_slot, 0, hbs_sps_slot_bytes()))) die("hbs_fill_device", rc);
    if ((rc = hbs_fill_device(g_ctx, g_dpps, 0, sizeof(hevc_pps_t)))) die("hbs_fill_device", rc);
    /* the device-side sets start all-zero, like a fresh hevc_stream_t's (hevc_nal.c:34-57 callocs them) */
    g_sps_shadow = (uint8_t*)calloc(1, sizeof(hevc_sps_t));
    g_pps_shadow = (uint8_t*)calloc(1, sizeof(hevc_pps_t));
    g_sps_shadow_ok = g_pps_shadow_ok = 1;
    if (getenv("HBS_LEGACY_TIMING")) {                          /* measuring aid: the time from here (GPU start-up done) to exit, on stderr */
        clock_gettime(CLOCK_MONOTONIC, &g_t_ready);
        atexit(report_time);
    }
    return 0;
}

/* the result block laid out for a NAL whose RBSP takes at most rbsp_bytes */
static void need_block(uint64_t rbsp_bytes)
{
    int rc;
    const uint64_t r = (rbsp_bytes + 64 + 255) & ~(uint64_t)255;
    const uint64_t want = RES_RBSP + r + sizeof(hevc_vps_t) + 64;
    if (want > g_dblock_cap) {
        if (g_dblock) hbs_dev_free(g_ctx, g_dblock);
        g_dblock_cap = want + r / 2;
        if ((rc = hbs_dev_alloc(g_ctx, g_dblock_cap, (void**)&g_dblock))) die("hbs_dev_alloc", rc);
        free(g_hres);
        g_hres = (uint8_t*)malloc((size_t)(g_dblock_cap - sizeof(hevc_vps_t)) + 8192);
    }
    g_res_r = r;
    g_dsummary = (hbs_summary*)(g_dblock + RES_SUMMARY);
    g_dindex = (hbs_nal_entry*)(g_dblock + RES_INDEX);
    g_dparsed = (hbs_parsed_nal*)(g_dblock + RES_PARSED);
    g_drbsp = g_dblock + RES_RBSP;
    g_dstruct = g_dblock + RES_RBSP + r;
}

/* the pinned staging buffer, at least `bytes` long (the previous call's upload has long been consumed: every
 * wrapper ends with a synchronising copy) */
static void need_stage(uint64_t bytes)
{
    int rc;
    if (bytes + 64 > g_hstage_cap) {
        if (g_hstage) hbs_host_free(g_ctx, g_hstage);
        g_hstage_cap = bytes * 2 + 65536;
        if ((rc = hbs_host_alloc(g_ctx, g_hstage_cap, (void**)&g_hstage))) die("hbs_host_alloc", rc);
    }
}

/* [prefix | bytes] into the device input buffer at offset 0; returns without waiting */
static void upload_input(const uint8_t* prefix, uint64_t prefix_len, const uint8_t* bytes, uint64_t len)
{
    int rc;
    need_stage(prefix_len + len);
    if (prefix_len) memcpy(g_hstage, prefix, prefix_len);
    if (len) memcpy(g_hstage + prefix_len, bytes, len);
    if ((rc = hbs_copy_to_device_async(g_ctx, g_dbuf, g_hstage, prefix_len + len))) die("hbs_copy_to_device_async", rc);
}

/* the device copies of the parameter sets in force follow the caller's object */
static void sync_context(const hevc_stream_t* h)
{
    int rc;
    if (!g_sps_shadow_ok || memcmp(g_sps_shadow, h->sps, sizeof(hevc_sps_t)) != 0) {
        if ((rc = hbs_copy_to_device(g_ctx, g_dsps_slot, h->sps, sizeof(hevc_sps_t)))) die("hbs_copy_to_device", rc);
        memcpy(g_sps_shadow, h->sps, sizeof(hevc_sps_t));
        g_sps_shadow_ok = 1;
    }
    if (!g_pps_shadow_ok || memcmp(g_pps_shadow, h->pps, sizeof(hevc_pps_t)) != 0) {
        if ((rc = hbs_copy_to_device(g_ctx, g_dpps, h->pps, sizeof(hevc_pps_t)))) die("hbs_copy_to_device", rc);
        memcpy(g_pps_shadow, h->pps, sizeof(hevc_pps_t));
        g_pps_shadow_ok = 1;
    }
}

static void need_bufs(uint64_t in_bytes, uint64_t out_bytes)
{
    int rc;
    if (in_bytes + 64 > g_dbuf_cap) {
        if (g_dbuf) hbs_dev_free(g_ctx, g_dbuf);
        g_dbuf_cap = in_bytes + in_bytes / 2 + 4096;
        if ((rc = hbs_dev_alloc(g_ctx, g_dbuf_cap, (void**)&g_dbuf))) die("hbs_dev_alloc", rc);
    }
    if (out_bytes + 64 > g_dout_cap) {
        if (g_dout) hbs_dev_free(g_ctx, g_dout);
        g_dout_cap = out_bytes + out_bytes / 2 + 4096;
        if ((rc = hbs_dev_alloc(g_ctx, g_dout_cap, (void**)&g_dout))) die("hbs_dev_alloc", rc);
    }
}

/* scan + extract the first `bytes` of the device input buffer (no wait) */
static void launch_index(uint64_t bytes, int want_rbsp)
{
    int rc = hbs_index_extract(g_ctx, g_dbuf, bytes, g_dindex, LEGACY_INDEX_CAP,
                               want_rbsp ? g_drbsp : NULL, want_rbsp ? g_res_r : 0, g_dsummary);
    if (rc) die("hbs_index_extract", rc);
}

/* the front `bytes` of the result block to the host (waits for everything enqueued so far) */
static void fetch_results(uint64_t bytes, hbs_summary* sum, hbs_nal_entry* ent)
{
    int rc;
    if ((rc = hbs_copy_to_host(g_ctx, g_hres, g_dblock, bytes))) die("hbs_copy_to_host", rc);
    memcpy(sum, g_hres + RES_SUMMARY, sizeof(*sum));
    memcpy(ent, g_hres + RES_INDEX, LEGACY_INDEX_CAP * sizeof(hbs_nal_entry));
}

/* ---- one batch per buffer (round 4) ------------------------------------------------------------------
 * The reference's callers walk a buffer NAL by NAL (hevc_analyze.c:135-177: find_nal_unit, read_debug_hevc_nal_unit, on to
 * the NAL's end); one GPU round trip per call made that loop twenty times slower than the CPU it replaces.  Now the first
 * find_nal_unit of a buffer of WIN_MIN bytes or more indexes, extracts and PARSES all of it (up to WIN_MAX) as one batch --
 * from the parameter sets and derived tables in force, with the exact re-walk of hbs_parse_fix.h, so every NAL's answer is
 * the sequential parser's -- and brings index, records, structs, RBSP and traces to the host.  The calls that follow are
 * answered from there as long as they are the calls the reference's loop would make: the next NAL, at the address the
 * previous one ended, the same bytes (memcmp against the copy the batch was made from), the caller's h->sps / h->pps still
 * what the previous answers put there.  Anything else -- another buffer, changed bytes, a NAL whose trace is longer than
 * the batch keeps, a read in the other mode -- goes the old way, one call at a time, after the device-side state (the SPS
 * and PPS in force and the tables) has been brought up to the NALs answered so far (hbs_parse_headers_state over them). */
#define WIN_MIN (128u << 10)
#define WIN_MAX (64u << 20)
#define WIN_TRACE_CAP_MAX 1024u
#define WIN_TRACE_CAP_MIN 128u
#define WIN_TRACE_BYTES_MAX (256ull << 20)   /* trace records one batch may hold, on the device and on the host (round 4's advice: 1024 records
                                                 for every NAL of a 64 MiB window of 300-byte NALs were 2.7 GB; the worst case 17 GB) */
static uint32_t g_win_trace_cap = WIN_TRACE_CAP_MAX;     /* trace records a batch keeps per NAL at most (HBS_LEGACY_TRACE_CAP lowers it: a testing aid) */
#define WIN_TRACE_CAP W.trace_cap
typedef struct {
    int valid, trace;
    int truncated;                   /* the trace-memory bound cut the batch short of the NALs the window holds: what lies behind its last NAL is
                                        more NALs, not an unterminated one (round 5's advice) */
    int parsed_ok;                   /* 0: indexed and extracted only -- the parse waits for the first read, which knows the mode (and the caller's sets) */
    uint32_t trace_cap;              /* trace records kept per NAL in THIS batch: g_win_trace_cap, less when n x cap x 12 B would pass WIN_TRACE_BYTES_MAX */
    const uint8_t* base;             /* the caller's buffer the batch was made from, and a copy of those bytes */
    uint64_t len;
    uint8_t* copy; uint64_t copy_cap;
    uint64_t n;                      /* NALs of the batch: those terminated inside the window */
    uint64_t find_next;              /* the NAL the next find_nal_unit is expected to ask for */
    uint64_t served;                 /* NALs read so far, in order */
    uint64_t synced;                 /* the device-side parser state reflects NALs [0, synced) */
    hbs_nal_entry* ent; hbs_parsed_nal* parsed; uint64_t ent_cap;
    uint8_t* structs; uint64_t structs_cap;
    uint8_t* rbsp; uint64_t rbsp_cap;
    hbs_trace_rec* tr; uint32_t* trn; uint64_t tr_cap;
    uint64_t struct_bytes, rbsp_bytes;
    /* device */
    uint8_t* d_stream; uint64_t d_stream_cap;
    uint8_t* d_rbsp; uint64_t d_rbsp_cap;
    uint8_t* d_index; uint64_t d_index_cap;      /* entries, then records */
    uint8_t* d_structs; uint64_t d_structs_cap;
    uint8_t* d_tr; uint64_t d_tr_cap;            /* trace records, then counts */
    uint8_t* d_misc;                             /* summary (64), end-state SPS slot, end-state PPS */
} window_t;
static window_t W;
/* Back-off (round 4's advice).  A batch costs an upload of up to 64 MiB, index, extraction, a parse and the way back; a caller
 * whose reads are not the batch's NALs in order (it parses from a scratch copy, flips between the two readers, edits h->sps)
 * gets nothing for it, and without a brake the next find_nal_unit of a large buffer built the next batch: once per NAL,
 * quadratic in the buffer.  So a batch that is dropped before it paid for itself suppresses the next g_backoff batch builds -- 1,
 * 2, 4 ... 65536, doubling while batches keep failing -- and one that pays resets the count.  "Paid": the calls answered from it
 * (reads, or the find_nal_unit calls of a caller that only walks) at ~150 us apiece, what each costs by itself, against ~300 us
 * + the window's bytes at ~4 GB/s up and down: 3 calls for a 128 KiB buffer, ~115 for a full 64 MiB window. */
static uint32_t g_backoff = 0, g_skip_builds = 0;
static uint64_t g_stat[4];           /* batches built, reads answered from a batch, reads answered one call at a time, builds suppressed */
static int g_no_batch = -1;          /* HBS_LEGACY_NO_BATCH, read once */

static int g_debug = -1;              /* HBS_LEGACY_DEBUG: the batch's life on stderr (a debugging aid) */
#define DBG(...) do { if (g_debug < 0) g_debug = getenv("HBS_LEGACY_DEBUG") ? 1 : 0; if (g_debug) fprintf(stderr, "hbs_legacy: " __VA_ARGS__); } while (0)

static void window_drop(void)
{
    if (!W.valid) return;
    W.valid = 0;
    DBG("drop: served %llu, found %llu of %llu NALs, synced %llu\n", (unsigned long long)W.served, (unsigned long long)W.find_next, (unsigned long long)W.n, (unsigned long long)W.synced);
    if ((W.served > W.find_next ? W.served : W.find_next) * 600000ull >= 1200000ull + W.len) g_backoff = 0;
    else {
        g_backoff = g_backoff ? (g_backoff < 65536u ? g_backoff * 2 : g_backoff) : 1;
        g_skip_builds = g_backoff;
    }
}

static void grow_dev(uint8_t** p, uint64_t* cap, uint64_t want)
{
    int rc;
    if (want <= *cap) return;
    if (*p) hbs_dev_free(g_ctx, *p);
    *cap = want + want / 4 + 4096;
    if ((rc = hbs_dev_alloc(g_ctx, *cap, (void**)p))) die("hbs_dev_alloc", rc);
}
/* host buffers of a batch; 0: no memory -- a batch that does not fit the host is not kept, the calls go one at a time */
static int grow_host_try(void** p, uint64_t* cap, uint64_t want)
{
    void* q;
    if (want <= *cap) return 1;
    q = malloc((size_t)(want + want / 4 + 4096));
    if (!q) return 0;
    free(*p);
    *p = q; *cap = want + want / 4 + 4096;
    return 1;
}
#define W_SUM ((hbs_summary*)W.d_misc)
#define W_END_SPS (W.d_misc + 256)
#define W_END_PPS (W.d_misc + 256 + ((hbs_sps_slot_bytes() + 255) & ~(uint64_t)255))

/* NALs [a, b) of the batch once more, ONE at a time through the sequential parser (what read_nal runs for a single call), only
 * for what they leave behind in the device-side state: the way out when the state behind a partial range cannot be derived in
 * one pass (round 4's advice: this used to abort()) */
static void window_replay(uint64_t a, uint64_t b)
{
    uint64_t k;
    int rc;
    need_block(0);
    for (k = a; k < b; ++k) {
        const int t = W.parsed_ok ? W.parsed[k].nal_unit_type : -1;
        if ((W.ent[k].status & HBS_ST_ERROR)) continue;                      /* read_nal returns before it parses: nothing changes */
        if (W.parsed_ok && !(t == HEVC_NAL_UNIT_TYPE_SPS_NUT || t == HEVC_NAL_UNIT_TYPE_PPS_NUT || (t >= 0 && t <= 9) || (t >= 16 && t <= 21))) continue;
        if ((rc = hbs_parse_headers_ctx(g_ctx, W.d_rbsp, (const hbs_nal_entry*)W.d_index + k, 1, g_dparsed, g_dstruct, sizeof(hevc_vps_t) + 64,
                                        g_dsps_slot, g_dpps, (hbs_summary*)(g_dblock + RES_SUMMARY2)))) die("hbs_parse_headers_ctx", rc);
        if (!W.parsed_ok) {
            hbs_parsed_nal p;
            if ((rc = hbs_copy_to_host(g_ctx, &p, g_dparsed, sizeof(p)))) die("hbs_copy_to_host", rc);
            if (p.struct_off == ~0ull) continue;
            if (p.nal_unit_type == HEVC_NAL_UNIT_TYPE_SPS_NUT && (rc = hbs_copy_device(g_ctx, g_dsps_slot, g_dstruct, sizeof(hevc_sps_t)))) die("hbs_copy_device", rc);
            if (p.nal_unit_type == HEVC_NAL_UNIT_TYPE_PPS_NUT && (rc = hbs_copy_device(g_ctx, g_dpps, g_dstruct, sizeof(hevc_pps_t)))) die("hbs_copy_device", rc);
            continue;
        }
        if (W.parsed[k].struct_off == ~0ull) continue;
        /* an SPS's derived tables went straight into the slot's tables; the struct follows (as in read_nal) */
        if (t == HEVC_NAL_UNIT_TYPE_SPS_NUT && (rc = hbs_copy_device(g_ctx, g_dsps_slot, g_dstruct, sizeof(hevc_sps_t)))) die("hbs_copy_device", rc);
        if (t == HEVC_NAL_UNIT_TYPE_PPS_NUT && (rc = hbs_copy_device(g_ctx, g_dpps, g_dstruct, sizeof(hevc_pps_t)))) die("hbs_copy_device", rc);
    }
}

/* the device-side parser state (g_dsps_slot, g_dpps) up to the NALs answered from the batch so far */
static void window_settle(void)
{
    int rc;
    hbs_summary s;
    if (!W.valid || W.synced >= W.served) return;
    if (W.parsed_ok && W.synced == 0 && W.served == W.n) {     /* everything: the state the batch itself computed */
        if ((rc = hbs_copy_device(g_ctx, g_dsps_slot, W_END_SPS, hbs_sps_slot_bytes()))) die("hbs_copy_device", rc);
        if ((rc = hbs_copy_device(g_ctx, g_dpps, W_END_PPS, sizeof(hevc_pps_t)))) die("hbs_copy_device", rc);
    } else {
        const uint64_t a = W.synced, m = W.served - W.synced;
        /* (the records of this pass go behind the batch's own, which the host has already: the device copy is scratch by now) */
        rc = W.d_structs ? hbs_parse_headers_state(g_ctx, W.d_rbsp, (const hbs_nal_entry*)W.d_index + a, m,
                                          (hbs_parsed_nal*)(W.d_index + W.ent_cap * sizeof(hbs_nal_entry)) + a, W.d_structs, W.d_structs_cap,
                                          g_dsps_slot, g_dpps, NULL, 0, NULL, W_SUM, g_dsps_slot, g_dpps) : HBS_E_ARG;
        if (rc == 0 && (rc = hbs_read_summary(g_ctx, W_SUM, &s))) die("hbs_read_summary", rc);
        /* reserved[1]: a chain of own RPS sets deeper than the exact re-walk follows, somewhere in this range (the batch as a whole
         * had none at its END, or it would not have been kept; an intermediate point can); error: the struct arena of the batch
         * was cut for all of it, a range cannot need more -- either way the single-NAL sequential path gives the same state */
        DBG("settle [%llu, %llu): rc %d deep %llu error %d\n", (unsigned long long)a, (unsigned long long)W.served, rc, rc ? 0ull : (unsigned long long)s.reserved[1], rc ? 0 : s.error);
        if (rc || s.reserved[1] || s.error) window_replay(a, W.served);
    }
    W.synced = W.served;
}

/* parse the NALs of the window in the given mode, from the device-side state, and fetch the answers; W.valid = 0 when the batch
 * cannot be kept (a chain of own sets deeper than the re-walk follows, no memory for it): one call at a time then */
static void window_parse(int trace)
{
    int rc;
    hbs_summary s;
    uint64_t n = W.n;
    hbs_nal_entry* d_ent = (hbs_nal_entry*)W.d_index;
    hbs_parsed_nal* d_par = (hbs_parsed_nal*)(W.d_index + W.ent_cap * sizeof(hbs_nal_entry));
    W.trace_cap = g_win_trace_cap;
    if (trace) {
        /* bounded trace memory: fewer records per NAL first (a NAL with more goes one call at a time, the batch continues behind
         * it), then fewer NALs (the next find_nal_unit behind the batch's last NAL builds the next batch from there) */
        while ((uint64_t)W.trace_cap > WIN_TRACE_CAP_MIN && n * (uint64_t)W.trace_cap * sizeof(hbs_trace_rec) > WIN_TRACE_BYTES_MAX) W.trace_cap /= 2;
        if (n * (uint64_t)W.trace_cap * sizeof(hbs_trace_rec) > WIN_TRACE_BYTES_MAX) {
            n = WIN_TRACE_BYTES_MAX / ((uint64_t)W.trace_cap * sizeof(hbs_trace_rec));
            W.n = n;
            W.truncated = 1;
            if (W.find_next > W.n) W.find_next = W.n;                /* (finds that ran ahead of the reads: the batch answers for its own NALs only) */
            W.rbsp_bytes = W.ent[n - 1].rbsp_off + W.ent[n - 1].rbsp_len;
        }
    }
    /* how large is the struct arena? (a plan-only pass: sizes and offsets, nothing parsed) */
    if ((rc = hbs_parse_headers_ctx(g_ctx, W.d_rbsp, d_ent, n, d_par, NULL, 0, g_dsps_slot, g_dpps, W_SUM))) die("hbs_parse_headers_ctx", rc);
    if ((rc = hbs_read_summary(g_ctx, W_SUM, &s))) die("hbs_read_summary", rc);
    W.struct_bytes = s.reserved[0];
    grow_dev(&W.d_structs, &W.d_structs_cap, W.struct_bytes + 64);
    if (trace) grow_dev(&W.d_tr, &W.d_tr_cap, n * ((uint64_t)WIN_TRACE_CAP * sizeof(hbs_trace_rec) + 4) + 64);
    {
        hbs_trace_rec* d_tr = trace ? (hbs_trace_rec*)W.d_tr : NULL;
        uint32_t* d_trn = trace ? (uint32_t*)(W.d_tr + n * (uint64_t)WIN_TRACE_CAP * sizeof(hbs_trace_rec)) : NULL;
        if ((rc = hbs_parse_headers_state(g_ctx, W.d_rbsp, d_ent, n, d_par, W.d_structs, W.d_structs_cap, g_dsps_slot, g_dpps,
                                          d_tr, trace ? WIN_TRACE_CAP : 0, d_trn, W_SUM, W_END_SPS, W_END_PPS))) die("hbs_parse_headers_state", rc);
    }
    if ((rc = hbs_read_summary(g_ctx, W_SUM, &s))) die("hbs_read_summary", rc);
    if (s.error || s.reserved[1]) { W.valid = 0; return; }      /* (a chain of own sets deeper than the re-walk follows: one call at a time) */
    if (!grow_host_try((void**)&W.structs, &W.structs_cap, W.struct_bytes + 64)) { W.valid = 0; return; }
    if ((rc = hbs_copy_to_host(g_ctx, W.parsed, d_par, n * sizeof(hbs_parsed_nal)))) die("hbs_copy_to_host", rc);
    if (W.struct_bytes && (rc = hbs_copy_to_host(g_ctx, W.structs, W.d_structs, W.struct_bytes))) die("hbs_copy_to_host", rc);
    if (trace) {
        free(W.trn); free(W.tr);
        W.trn = (uint32_t*)malloc((size_t)(n * 4 + 64));
        W.tr = (hbs_trace_rec*)malloc((size_t)(n * (uint64_t)WIN_TRACE_CAP * sizeof(hbs_trace_rec) + 64));
        if (!W.trn || !W.tr) { free(W.trn); free(W.tr); W.trn = NULL; W.tr = NULL; W.valid = 0; return; }
        if ((rc = hbs_copy_to_host(g_ctx, W.trn, W.d_tr + n * (uint64_t)WIN_TRACE_CAP * sizeof(hbs_trace_rec), n * 4))) die("hbs_copy_to_host", rc);
        /* (the records of a NAL sit WIN_TRACE_CAP apart: one copy of the block) */
        if ((rc = hbs_copy_to_host(g_ctx, W.tr, W.d_tr, n * (uint64_t)WIN_TRACE_CAP * sizeof(hbs_trace_rec)))) die("hbs_copy_to_host", rc);
    }
    W.trace = trace;
    W.parsed_ok = 1;
}

/* index + extract + parse the first min(size, WIN_MAX) bytes of buf as one batch; 0: no batch (too many NALs, nothing terminated) */
static int window_build(uint8_t* buf, int size)
{
    int rc;
    hbs_summary s;
    uint64_t len = (uint64_t)size < WIN_MAX ? (uint64_t)size : WIN_MAX, cap, n;
    static int env_read = 0;
    if (!env_read) {
        const char* tc = getenv("HBS_LEGACY_TRACE_CAP");
        if (tc && atoi(tc) > 0 && (uint32_t)atoi(tc) < WIN_TRACE_CAP_MAX) g_win_trace_cap = (uint32_t)atoi(tc);
        env_read = 1;
    }
    window_settle();
    window_drop();
    g_stat[0] += 1;
    cap = len / 48 + 64;                                        /* NALs of 48 bytes and less: not what this is for */
    grow_dev(&W.d_stream, &W.d_stream_cap, len + 64);
    grow_dev(&W.d_rbsp, &W.d_rbsp_cap, len + 64);
    if (cap > W.ent_cap) {
        W.ent_cap = cap + cap / 4;
        grow_dev(&W.d_index, &W.d_index_cap, W.ent_cap * (sizeof(hbs_nal_entry) + sizeof(hbs_parsed_nal)) + 64);
        free(W.ent); free(W.parsed);
        W.ent = (hbs_nal_entry*)malloc((size_t)(W.ent_cap * sizeof(hbs_nal_entry)));
        W.parsed = (hbs_parsed_nal*)malloc((size_t)(W.ent_cap * sizeof(hbs_parsed_nal)));
        if (!W.ent || !W.parsed) { free(W.ent); free(W.parsed); W.ent = NULL; W.parsed = NULL; W.ent_cap = 0; return 0; }   /* no batch: one call at a time */
    }
    if (!W.d_misc && (rc = hbs_dev_alloc(g_ctx, 256 + 2 * ((hbs_sps_slot_bytes() + 255) & ~(uint64_t)255) + sizeof(hevc_pps_t), (void**)&W.d_misc))) die("hbs_dev_alloc", rc);
    /* straight from the caller's buffer (a pageable copy that waits: page-locking a staging buffer of this size costs more than it saves) */
    if ((rc = hbs_copy_to_device(g_ctx, W.d_stream, buf, len))) die("hbs_copy_to_device", rc);
    if ((rc = hbs_index_extract(g_ctx, W.d_stream, len, (hbs_nal_entry*)W.d_index, W.ent_cap, W.d_rbsp, W.d_rbsp_cap, W_SUM))) die("hbs_index_extract", rc);
    if ((rc = hbs_read_summary(g_ctx, W_SUM, &s))) die("hbs_read_summary", rc);
    if (s.error || s.nal_count == 0) return 0;
    n = s.nal_count;
    if ((rc = hbs_copy_to_host(g_ctx, W.ent, W.d_index, n * sizeof(hbs_nal_entry)))) die("hbs_copy_to_host", rc);
    /* only NALs that end inside the window, with what terminates them (as find_nal_unit needs to see it) */
    while (n && ((W.ent[n - 1].status & HBS_ST_UNTERMINATED) || W.ent[n - 1].end + 4 > len)) --n;
    if (n < 2) return 0;
    W.n = n; W.len = len; W.base = buf;
    W.rbsp_bytes = W.ent[n - 1].rbsp_off + W.ent[n - 1].rbsp_len;
    if (!grow_host_try((void**)&W.copy, &W.copy_cap, len)) return 0;
    memcpy(W.copy, buf, (size_t)len);
    if (!grow_host_try((void**)&W.rbsp, &W.rbsp_cap, W.rbsp_bytes + 64)) return 0;
    if (W.rbsp_bytes && (rc = hbs_copy_to_host(g_ctx, W.rbsp, W.d_rbsp, W.rbsp_bytes))) die("hbs_copy_to_host", rc);
    /* The parse waits for the first read: only that call knows the mode (plain or trace -- round 4 parsed every first batch in
     * trace mode and a plain reader paid for a second pass) and the parameter sets the caller's object holds. */
    W.valid = 1; W.parsed_ok = 0; W.truncated = 0; W.find_next = 0; W.served = 0; W.synced = 0;
    DBG("build: %llu NALs in %llu bytes\n", (unsigned long long)n, (unsigned long long)len);
    return 1;
}

/* ---- byte layer ------------------------------------------------------------------------- */

static int find_nal_unit_unlocked(uint8_t* buf, int size, int* nal_start, int* nal_end)
{
    uint64_t len;
    *nal_start = 0;
    *nal_end = 0;
    if (size <= 0) return 0;
    if (need_ctx()) return 0;
    if (W.valid && W.find_next < W.n) {
        /* the next NAL of the batch, asked for where the previous one ended, over the same bytes */
        const uint64_t from = W.find_next ? W.ent[W.find_next - 1].end : 0;
        const hbs_nal_entry* cur = &W.ent[W.find_next];
        const uint64_t need = cur->end - from + 4;
        if (buf == W.base + from && (uint64_t)size >= need && memcmp(buf, W.copy + from, (size_t)need) == 0) {
            *nal_start = (int)(cur->start - from);
            *nal_end = (int)(cur->end - from);
            W.find_next++;
            return *nal_end - *nal_start;
        }
    }
    if (g_no_batch < 0) g_no_batch = getenv("HBS_LEGACY_NO_BATCH") ? 1 : 0;
    if ((uint64_t)size >= WIN_MIN && !g_no_batch) {
        /* a buffer worth a batch (not one this call was already answered from: that is the case above) */
        const int rest_of_old = W.valid && !W.truncated && W.find_next >= W.n && buf == W.base + W.ent[W.n - 1].end &&
                                (uint64_t)size <= W.len - W.ent[W.n - 1].end;     /* what the batch left: its unterminated last NAL */
        if (!rest_of_old && g_skip_builds > 0) {                                   /* the last batches did not pay: not this time */
            g_skip_builds -= 1;
            g_stat[3] += 1;
        } else if (!rest_of_old && window_build(buf, size)) {
            const hbs_nal_entry* cur = &W.ent[0];
            *nal_start = (int)cur->start;
            *nal_end = (int)cur->end;
            W.find_next = 1;
            return *nal_end - *nal_start;
        }
    }
    if (g_fc_next >= 1 && g_fc_next < g_fc_n) {
        const hbs_nal_entry* prev = &g_fc_ent[g_fc_next - 1];
        const hbs_nal_entry* cur = &g_fc_ent[g_fc_next];
        const uint64_t need = cur->end - prev->end + 4;        /* up to and including what terminates it */
        if (buf == g_fc_base + prev->end && !(cur->status & HBS_ST_UNTERMINATED) && cur->end + 4 <= g_fc_len &&
            (uint64_t)size >= need && memcmp(buf, g_fc_copy + prev->end, need) == 0) {
            *nal_start = (int)(cur->start - prev->end);
            *nal_end = (int)(cur->end - prev->end);
            g_fc_next++;
            return *nal_end - *nal_start;
        }
    }
    g_fc_n = g_fc_next = 0;
    /* The answer only depends on the bytes up to the first NAL's end (+3), so scan a growing
     * prefix: a NAL that is terminated inside the prefix is terminated the same way in the
     * whole buffer (its terminator starts at least 4 bytes before the prefix end). */
    for (len = 65536; ; len *= 4) {
        hbs_summary s;
        if (len > (uint64_t)size) len = (uint64_t)size;
        need_bufs(len, 0);
        upload_input(NULL, 0, buf, len);
        {
            int rc = hbs_index_extract(g_ctx, g_dbuf, len, (hbs_nal_entry*)(g_dfind + 64), FIND_INDEX_CAP, NULL, 0, (hbs_summary*)g_dfind);
            if (rc) die("hbs_index_extract", rc);
            if ((rc = hbs_copy_to_host(g_ctx, g_hfind, g_dfind, RES_FIND_BYTES))) die("hbs_copy_to_host", rc);
            memcpy(&s, g_hfind, sizeof(s));
            memcpy(g_fc_ent, g_hfind + 64, sizeof(g_fc_ent));
        }
        if (s.nal_found >= 1 && !(g_fc_ent[0].status & HBS_ST_UNTERMINATED)) {
            /* first NAL terminated inside the prefix (possibly empty: the loop of the callers stops) */
            *nal_start = (int)g_fc_ent[0].start;
            *nal_end = (int)g_fc_ent[0].end;
            if (len <= FIND_CACHE_MAX && s.nal_count > 1) {     /* keep the rest of the answer for the calls to come */
                if (!g_fc_copy) g_fc_copy = (uint8_t*)malloc(FIND_CACHE_MAX);
                memcpy(g_fc_copy, buf, (size_t)len);
                g_fc_base = buf; g_fc_len = len;
                g_fc_n = s.nal_count < FIND_INDEX_CAP ? s.nal_count : FIND_INDEX_CAP;
                g_fc_next = 1;
            }
            return *nal_end - *nal_start;
        }
        if (len == (uint64_t)size) {
            if (s.nal_found >= 1) {                     /* start found, end not: h264_nal.c:71 */
                *nal_start = (int)g_fc_ent[0].start;
                *nal_end = size;
                return -1;
            }
            return 0;                                   /* no start code: h264_nal.c:52 */
        }
    }
}

static int nal_to_rbsp_unlocked(const uint8_t* nal_buf, int* nal_size, uint8_t* rbsp_buf, int* rbsp_size)
{
    static const uint8_t sc[3] = {0, 0, 1};
    const int n = *nal_size;
    hbs_summary s;
    hbs_nal_entry e[LEGACY_INDEX_CAP];
    if (n < 0) return -1;
    if (need_ctx()) return -1;
    need_bufs((uint64_t)n + 16, 0);
    need_block((uint64_t)n + 16);
    /* the kernel works on Annex-B: put a start code in front of the NAL */
    upload_input(sc, 3, nal_buf, (uint64_t)n);
    launch_index((uint64_t)n + 3, 1);
    fetch_results(RES_RBSP + (uint64_t)n + 16, &s, e);             /* the RBSP comes along */
    /* a 00 00 00 / 00 00 01 inside the NAL would end it early: nal_to_rbsp rejects those (h264_nal.c:156-159) */
    if (s.nal_found < 1 || e[0].start != 3 || e[0].end != (uint64_t)n + 3 || (e[0].status & HBS_ST_ERROR)) return -1;
    if ((int)e[0].rbsp_len > *rbsp_size) return -1;                      /* h264_nal.c:179-183 */
    memcpy(rbsp_buf, g_hres + RES_RBSP + e[0].rbsp_off, e[0].rbsp_len);
    *nal_size = (e[0].status & HBS_ST_TRAILING03) ? n - 1 : n;           /* h264_nal.c:170-173, :197 */
    *rbsp_size = (int)e[0].rbsp_len;
    return (int)e[0].rbsp_len;
}

/* One NAL's RBSP (n bytes at d_rbsp, already on the device or on its way there) -> nal_buf; the index entry travels
 * through the pinned staging buffer (with `payload`, if given, right behind it: then d_rbsp must be g_dbuf + 32), the
 * summary and the bytes come back in one copy of the result block. */
static int emit_one(const uint8_t* payload, const uint8_t* d_rbsp, int n, uint8_t* nal_buf)
{
    hbs_nal_entry e;
    hbs_summary s;
    int rc;
    uint64_t out_bytes;
    const uint64_t bound = hbs_annexb_bound((uint64_t)n, 1);
    need_block(bound);
    memset(&e, 0, sizeof(e));
    e.rbsp_off = 0; e.rbsp_len = (uint32_t)n;
    upload_input((const uint8_t*)&e, sizeof(e), payload, payload ? (uint64_t)n : 0);
    /* gap_mode 1, NAL 0: a 4-byte start code goes in front; it is not part of rbsp_to_nal's output */
    if ((rc = hbs_emit_annexb(g_ctx, d_rbsp, (uint64_t)n, (const hbs_nal_entry*)g_dbuf, 1, 1, g_drbsp, g_res_r, NULL, g_dsummary))) die("hbs_emit_annexb", rc);
    if ((rc = hbs_copy_to_host(g_ctx, g_hres, g_dblock, RES_RBSP + bound))) die("hbs_copy_to_host", rc);
    memcpy(&s, g_hres + RES_SUMMARY, sizeof(s));
    if (s.error) die("hbs_emit_annexb(capacity)", s.error);
    out_bytes = s.stream_bytes - 4;
    memcpy(nal_buf, g_hres + RES_RBSP + 4, out_bytes);
    return (int)out_bytes;
}

static int rbsp_to_nal_unlocked(const uint8_t* rbsp_buf, const int* rbsp_size, uint8_t* nal_buf, int* nal_size)
{
    const int n = *rbsp_size;
    if (n <= 0) { *nal_size = 0; return 0; }
    if (need_ctx()) return -1;
    need_bufs((uint64_t)n + 64, 0);
    *nal_size = emit_one(rbsp_buf, g_dbuf + sizeof(hbs_nal_entry), n, nal_buf);      /* h264_nal.c:130 */
    return *nal_size;
}

int find_nal_unit(uint8_t* buf, int size, int* nal_start, int* nal_end)
{
    int r;
    legacy_lock();
    r = find_nal_unit_unlocked(buf, size, nal_start, nal_end);
    legacy_unlock();
    return r;
}
int nal_to_rbsp(const uint8_t* nal_buf, int* nal_size, uint8_t* rbsp_buf, int* rbsp_size)
{
    int r;
    legacy_lock();
    r = nal_to_rbsp_unlocked(nal_buf, nal_size, rbsp_buf, rbsp_size);
    legacy_unlock();
    return r;
}
int rbsp_to_nal(const uint8_t* rbsp_buf, const int* rbsp_size, uint8_t* nal_buf, int* nal_size)
{
    int r;
    legacy_lock();
    r = rbsp_to_nal_unlocked(rbsp_buf, rbsp_size, nal_buf, nal_size);
    legacy_unlock();
    return r;
}

/* h264_stream.c:117-126 */
void debug_bytes(uint8_t* buf, int len)
{
    FILE* f = h264_dbgfile ? h264_dbgfile : stdout;
    int i;
    for (i = 0; i < len; i++) {
        fprintf(f, "%02X ", buf[i]);
        if ((i + 1) % 16 == 0) fprintf(f, "\n");
    }
    fprintf(f, "\n");
}

/* ---- parser object ------------------------------------------------------------------------ */

hevc_stream_t* hevc_new()
{
    int i;
    hevc_stream_t* h = (hevc_stream_t*)calloc(1, sizeof(hevc_stream_t));
    h->nal = (hevc_nal_t*)calloc(1, sizeof(hevc_nal_t));
    for (i = 0; i < 32; i++) h->sps_table[i] = (hevc_sps_t*)calloc(1, sizeof(hevc_sps_t));
    for (i = 0; i < 256; i++) h->pps_table[i] = (hevc_pps_t*)calloc(1, sizeof(hevc_pps_t));
    h->vps = (hevc_vps_t*)calloc(1, sizeof(hevc_vps_t));
    h->sps = (hevc_sps_t*)calloc(1, sizeof(hevc_sps_t));
    h->pps = (hevc_pps_t*)calloc(1, sizeof(hevc_pps_t));
    h->aud = (hevc_aud_t*)calloc(1, sizeof(hevc_aud_t));
    h->sh = (hevc_slice_header_t*)calloc(1, sizeof(hevc_slice_header_t));
    h->slice_data = (hevc_slice_data_rbsp_t*)calloc(1, sizeof(hevc_slice_data_rbsp_t));
    return h;
}

void hevc_free(hevc_stream_t* h)
{
    int i;
    if (!h) return;
    free(h->nal);
    for (i = 0; i < 32; i++) free(h->sps_table[i]);
    for (i = 0; i < 256; i++) free(h->pps_table[i]);
    if (h->slice_data) free(h->slice_data->rbsp_buf);       /* the reference leaks this buffer */
    free(h->slice_data);
    free(h->sh); free(h->aud); free(h->pps); free(h->sps); free(h->vps);
    free(h);
}

/* hevc_nal.c:97-114: NAL header straight from the first two bytes (no RBSP conversion) */
int peek_hevc_nal_unit(hevc_stream_t* h, uint8_t* buf, int size)
{
    const unsigned b0 = size > 0 ? buf[0] : 0, b1 = size > 1 ? buf[1] : 0;
    h->nal->nal_unit_type = (b0 >> 1) & 0x3F;
    h->nal->nal_layer_id = ((b0 & 1) << 5) | (b1 >> 3);
    h->nal->nal_temporal_id_plus1 = b1 & 7;
    if (h->nal->nal_unit_type <= 0 || h->nal->nal_unit_type > MAX_HEVC_VAL_UNIT_TYPE) return -1;
    return h->nal->nal_unit_type;
}

/* ---- per-field trace (read_debug_hevc_nal_unit) ------------------------------------------- */
#include "hbs_trace_names.h"

#define TRACE_CAP 65536u
static hbs_trace_rec* g_dtrace = NULL;
static uint32_t* g_dtrace_count = NULL;
static hbs_trace_rec* g_htrace = NULL;

static void need_trace(void)
{
    int rc;
    if (g_dtrace) return;
    if ((rc = hbs_dev_alloc(g_ctx, (uint64_t)TRACE_CAP * sizeof(hbs_trace_rec), (void**)&g_dtrace))) die("hbs_dev_alloc", rc);
    if ((rc = hbs_dev_alloc(g_ctx, 16, (void**)&g_dtrace_count))) die("hbs_dev_alloc", rc);
    g_htrace = (hbs_trace_rec*)malloc((size_t)TRACE_CAP * sizeof(hbs_trace_rec));
}

static const char* trace_name(unsigned site)
{
    int lo = 0, hi = (int)(sizeof(hbs_trace_names) / sizeof(hbs_trace_names[0])) - 1;
    while (lo <= hi) {
        const int mid = (lo + hi) / 2;
        if (hbs_trace_names[mid].site == site) return hbs_trace_names[mid].name;
        if (hbs_trace_names[mid].site < site) lo = mid + 1; else hi = mid - 1;
    }
    return NULL;
}

static void print_trace(void)
{
    uint32_t n = 0, i;
    int rc;
    if ((rc = hbs_copy_to_host(g_ctx, &n, g_dtrace_count, sizeof(n)))) die("hbs_copy_to_host", rc);
    if (n > TRACE_CAP) n = TRACE_CAP;
    if (n && (rc = hbs_copy_to_host(g_ctx, g_htrace, g_dtrace, (uint64_t)n * sizeof(hbs_trace_rec)))) die("hbs_copy_to_host", rc);
    for (i = 0; i < n; i++) {
        const char* name = trace_name(g_htrace[i].site);
        printf("%ld.%d: ", (long)(g_htrace[i].pos >> 3), (int)(8 - (g_htrace[i].pos & 7)));     /* b->p - b->start, b->bits_left */
        if (!name) printf("site_%u: %d \n", g_htrace[i].site, g_htrace[i].value);                /* table out of date */
        else if (name[0]) printf("%s: %d \n", name, g_htrace[i].value);
        /* an empty name: the reference prints the cursor and nothing else there (hevc_stream.c:3147) */
    }
}

static int is_slice(int t) { return (t >= 0 && t <= 9) || (t >= 16 && t <= 21); }

static void print_trace_records(const hbs_trace_rec* tr, uint32_t n)
{
    uint32_t i;
    for (i = 0; i < n; i++) {
        const char* name = trace_name(tr[i].site);
        printf("%ld.%d: ", (long)(tr[i].pos >> 3), (int)(8 - (tr[i].pos & 7)));
        if (!name) printf("site_%u: %d \n", tr[i].site, tr[i].value);
        else if (name[0]) printf("%s: %d \n", name, tr[i].value);
    }
}

/* NAL W.served of the batch, checked by the caller to be what is asked for: the answer read_nal would fetch from the GPU,
 * from the host copies the batch left (same order of side effects on *h as below) */
static int serve_from_window(hevc_stream_t* h, uint8_t* buf, int size, int* stripped, int trace)
{
    const uint64_t k = W.served;
    const hbs_nal_entry* e = &W.ent[k];
    const hbs_parsed_nal* p = &W.parsed[k];
    const uint8_t* src;
    int t;
    W.served = k + 1;
    if (e->status & HBS_ST_ERROR) return -1;                             /* hevc_stream.c:167: nothing of *h is touched */
    *stripped = 1;
    if (trace) {
        printf("0.8: forbidden_zero_bit: %d \n", (buf[0] >> 7) & 1);
        printf("0.7: nal->nal_unit_type: %d \n", (buf[0] >> 1) & 0x3F);
        printf("0.1: nal->nal_layer_id: %d \n", ((buf[0] & 1) << 5) | ((size > 1 ? buf[1] : 0) >> 3));
        printf("1.3: nal->nal_temporal_id_plus1: %d \n", (size > 1 ? buf[1] : 0) & 7);
        print_trace_records(W.tr + k * (uint64_t)WIN_TRACE_CAP, W.trn[k] < WIN_TRACE_CAP ? W.trn[k] : WIN_TRACE_CAP);
    }
    t = p->nal_unit_type;
    h->nal->nal_unit_type = t;
    h->nal->nal_layer_id = p->nal_layer_id;
    h->nal->nal_temporal_id_plus1 = p->nal_temporal_id_plus1;
    if (p->struct_off == ~0ull) return -1;                               /* unsupported type: hevc_stream.c:221 */
    src = W.structs + p->struct_off;
    if (t == HEVC_NAL_UNIT_TYPE_VPS_NUT) {
        memcpy(h->vps, src, sizeof(hevc_vps_t));
    } else if (t == HEVC_NAL_UNIT_TYPE_SPS_NUT) {
        memcpy(h->sps, src, sizeof(hevc_sps_t));
        memcpy(g_sps_shadow, src, sizeof(hevc_sps_t));                   /* what the device-side state will hold once it is brought up to here */
        g_sps_shadow_ok = 1;
        if (h->sps->sps_seq_parameter_set_id >= 0 && h->sps->sps_seq_parameter_set_id < 32)
            memcpy(h->sps_table[h->sps->sps_seq_parameter_set_id], h->sps, sizeof(hevc_sps_t));
    } else if (t == HEVC_NAL_UNIT_TYPE_PPS_NUT) {
        memcpy(h->pps, src, sizeof(hevc_pps_t));
        memcpy(g_pps_shadow, src, sizeof(hevc_pps_t));
        g_pps_shadow_ok = 1;
        if (h->pps->pic_parameter_set_id >= 0 && h->pps->pic_parameter_set_id < 256)
            memcpy(h->pps_table[h->pps->pic_parameter_set_id], h->pps, sizeof(hevc_pps_t));
    } else if (is_slice(t)) {
        memcpy(h->sh, src, sizeof(hevc_slice_header_t));
        if (h->slice_data) {
            free(h->slice_data->rbsp_buf);
            h->slice_data->rbsp_buf = NULL;
            h->slice_data->rbsp_size = p->slice_data_size;
            if (p->slice_data_size >= 0) {
                h->slice_data->rbsp_buf = (uint8_t*)malloc((size_t)p->slice_data_size + 1);
                if (p->slice_data_size > 0)
                    memcpy(h->slice_data->rbsp_buf, W.rbsp + e->rbsp_off + p->slice_data_off, (size_t)p->slice_data_size);
            }
        }
    }
    if (W.served == W.n) window_settle();                                /* the batch is used up: the state behind it, for whoever comes next */
    return p->rc;
}

/* *stripped = 0 when nal_to_rbsp already rejected the NAL (nothing of *h is touched then).
 * One upload, the scan and the parse back to back, one download of summary + index + parsed record +
 * slice-sized struct: the host looks at the scan's answer only afterwards (a parse of a NAL the scan
 * rejected writes device scratch nobody reads). */
static int read_nal(hevc_stream_t* h, uint8_t* buf, int size, int* stripped, int trace)
{
    static const uint8_t sc[3] = {0, 0, 1};
    hbs_summary s;
    hbs_nal_entry e[LEGACY_INDEX_CAP];
    hbs_parsed_nal p;
    int rc, t;
    *stripped = 0;
    if (size < 0) return -1;
    if (need_ctx()) return -1;
    if (W.valid && W.served < W.n) {
        uint64_t k = W.served;
        const hbs_nal_entry* we = &W.ent[k];
        int mine;
        if (!W.parsed_ok && buf >= W.base + we->start && buf < W.base + W.len) {
            /* the batch's parse, now that the mode and the caller's parameter sets are known */
            sync_context(h);
            window_parse(trace);
            if (!W.valid) { W.valid = 1; window_drop(); }
        }
        mine = W.valid && W.parsed_ok && buf == W.base + we->start && (uint64_t)size == we->end - we->start;
        if (!mine && W.valid && W.parsed_ok && buf > W.base + we->start && buf < W.base + W.len) {
            /* A LATER NAL of the batch: the caller skipped some (an AUD, SEI or VPS it has no use for).  Skipped NALs that the
             * parser would not have taken anything from or left anything behind -- everything but SPS, PPS and slices -- change
             * nothing for the NALs behind them: the batch goes on at this one, and the state pass over the skipped range
             * (window_settle) passes over them just as well.  A skipped parameter set or slice ends the batch. */
            uint64_t j = k;
            const uint64_t off = (uint64_t)(buf - W.base);
            while (j < W.n && W.ent[j].start < off) {
                const int tj = W.parsed[j].nal_unit_type;
                if ((W.ent[j].status & HBS_ST_ERROR) == 0 && (tj == HEVC_NAL_UNIT_TYPE_SPS_NUT || tj == HEVC_NAL_UNIT_TYPE_PPS_NUT || is_slice(tj))) break;
                ++j;
            }
            if (j < W.n && W.ent[j].start == off && (uint64_t)size == W.ent[j].end - off) { k = j; we = &W.ent[k]; W.served = k; mine = 1; }
        }
        mine = mine && memcmp(buf, W.copy + we->start, (size_t)size) == 0;
        mine = mine && W.valid && g_sps_shadow_ok && g_pps_shadow_ok && memcmp(g_sps_shadow, h->sps, sizeof(hevc_sps_t)) == 0 &&
               memcmp(g_pps_shadow, h->pps, sizeof(hevc_pps_t)) == 0;
        if (mine && trace != W.trace && k == W.synced) {
            /* the batch was parsed for the other reader: parse what is left of it again, in this one */
            window_settle();
            if (k) {                                            /* (from NAL k on: the window shrinks to its rest) */
                window_drop();
            } else window_parse(trace);
        }
        if (mine && W.valid && trace == W.trace && (!trace || W.trn[k] <= WIN_TRACE_CAP)) { g_stat[1] += 1; return serve_from_window(h, buf, size, stripped, trace); }
        /* not the call the batch expected, or one it cannot answer: the old way, from the state behind the NALs answered so far */
        window_settle();
        if (mine && W.valid) { /* this very NAL, one call at a time; the batch goes on behind it */ }
        else window_drop();
    } else if (W.valid) {
        window_settle();                                        /* everything answered: the old way continues from the state behind the batch */
    }
    g_stat[2] += 1;
    need_bufs((uint64_t)size + 16, 0);
    need_block((uint64_t)size + 16);
    upload_input(sc, 3, buf, (uint64_t)size);
    /* the parameter sets in force are whatever the caller's object holds (hevc_stream.c:800-801);
     * the derived RPS tables live on the device next to the SPS */
    sync_context(h);
    launch_index((uint64_t)size + 3, 1);
    if (trace) need_trace();
    if ((rc = hbs_parse_headers_trace(g_ctx, g_drbsp, g_dindex, 1, g_dparsed, g_dstruct, sizeof(hevc_vps_t) + 64,
                                      g_dsps_slot, g_dpps, trace ? g_dtrace : NULL, trace ? TRACE_CAP : 0,
                                      trace ? g_dtrace_count : NULL, (hbs_summary*)(g_dblock + RES_SUMMARY2)))) die("hbs_parse_headers", rc);
    fetch_results(RES_RBSP + g_res_r + SLICE_SLOT, &s, e);          /* results, the RBSP, a slice-sized struct */
    memcpy(&p, g_hres + RES_PARSED, sizeof(p));
    if (s.nal_found < 1 || e[0].start != 3 || e[0].end != (uint64_t)size + 3 || (e[0].status & HBS_ST_ERROR))
        return -1;                                                       /* hevc_stream.c:167 */
    *stripped = 1;
    if (trace) {
        /* hevc_stream.c:2363-2367: the header lines, then one line per syntax element the parser read */
        printf("0.8: forbidden_zero_bit: %d \n", (buf[0] >> 7) & 1);
        printf("0.7: nal->nal_unit_type: %d \n", (buf[0] >> 1) & 0x3F);
        printf("0.1: nal->nal_layer_id: %d \n", ((buf[0] & 1) << 5) | ((size > 1 ? buf[1] : 0) >> 3));
        printf("1.3: nal->nal_temporal_id_plus1: %d \n", (size > 1 ? buf[1] : 0) & 7);
        print_trace();
    }
    t = p.nal_unit_type;
    h->nal->nal_unit_type = t;
    h->nal->nal_layer_id = p.nal_layer_id;
    h->nal->nal_temporal_id_plus1 = p.nal_temporal_id_plus1;
    if (p.struct_off == ~0ull) return -1;                                /* unsupported type: hevc_stream.c:221 */
    if (t == HEVC_NAL_UNIT_TYPE_VPS_NUT) {
        if ((rc = hbs_copy_to_host(g_ctx, h->vps, g_dstruct, sizeof(hevc_vps_t)))) die("hbs_copy_to_host", rc);
    } else if (t == HEVC_NAL_UNIT_TYPE_SPS_NUT) {
        /* keep it for the slices to come (device to device; its derived tables were written straight into the slot's
         * tables, next to the rows earlier parameter sets and slices left there), then fetch the struct */
        if ((rc = hbs_copy_device(g_ctx, g_dsps_slot, g_dstruct, sizeof(hevc_sps_t)))) die("hbs_copy_device", rc);
        if ((rc = hbs_copy_to_host(g_ctx, h->sps, g_dstruct, sizeof(hevc_sps_t)))) die("hbs_copy_to_host", rc);
        memcpy(g_sps_shadow, h->sps, sizeof(hevc_sps_t));
        g_sps_shadow_ok = 1;
        if (h->sps->sps_seq_parameter_set_id >= 0 && h->sps->sps_seq_parameter_set_id < 32)
            memcpy(h->sps_table[h->sps->sps_seq_parameter_set_id], h->sps, sizeof(hevc_sps_t));      /* :399 */
    } else if (t == HEVC_NAL_UNIT_TYPE_PPS_NUT) {
        memcpy(h->pps, g_hres + RES_RBSP + g_res_r, sizeof(hevc_pps_t));  /* smaller than a slice header: already here */
        if (h->pps->pic_parameter_set_id >= 0 && h->pps->pic_parameter_set_id < 256)
            memcpy(h->pps_table[h->pps->pic_parameter_set_id], h->pps, sizeof(hevc_pps_t));          /* :498 */
    } else if (is_slice(t)) {
        memcpy(h->sh, g_hres + RES_RBSP + g_res_r, sizeof(hevc_slice_header_t));
        if (h->slice_data) {                                             /* hevc_stream.c:605-613 */
            free(h->slice_data->rbsp_buf);
            h->slice_data->rbsp_buf = NULL;
            h->slice_data->rbsp_size = p.slice_data_size;
            if (p.slice_data_size >= 0) {                                /* malloc(0) is a pointer too, as in the reference */
                h->slice_data->rbsp_buf = (uint8_t*)malloc((size_t)p.slice_data_size + 1);
                if (p.slice_data_size > 0)
                    memcpy(h->slice_data->rbsp_buf, g_hres + RES_RBSP + e[0].rbsp_off + p.slice_data_off, (size_t)p.slice_data_size);
            }
        }
    }
    return p.rc;
}

/* Test hook, not part of the reference's API: the derived RPS tables are process state there (file-static,
 * zero at program start, hevc_stream.c:26-32) and here (on the device); this puts them back to "program start". */
void hbs_legacy_reset_tables(void)
{
    int rc;
    legacy_lock();
    if (need_ctx() == 0) {
        W.valid = 0;
        g_backoff = g_skip_builds = 0;
        if ((rc = hbs_fill_device(g_ctx, g_dsps_slot, 0, hbs_sps_slot_bytes()))) die("hbs_fill_device", rc);
        /* the slot's SPS is all zero now and the shadow says so: batches stay possible (round 4's advice: with the shadow marked
         * unknown every batch was built and then turned down until a one-call SPS read came by) */
        memset(g_sps_shadow, 0, sizeof(hevc_sps_t));
        g_sps_shadow_ok = 1;
    }
    legacy_unlock();
}

/* Test hook: out[0] batches built, [1] reads answered from a batch, [2] reads answered one call at a time, [3] batch builds
 * suppressed by the back-off. */
void hbs_legacy_batch_stats(uint64_t out[4])
{
    legacy_lock();
    memcpy(out, g_stat, sizeof(g_stat));
    legacy_unlock();
}

/* a read that went the old way although it was the batch's next NAL (its trace was too long for the batch): the batch goes on
 * behind it -- the device-side state now includes it */
static int read_nal_w(hevc_stream_t* h, uint8_t* buf, int size, int* stripped, int trace)
{
    const int was_valid = W.valid;
    const int r = read_nal(h, buf, size, stripped, trace);
    const uint64_t k = W.served;         /* (answered from the batch: already one further, and buf is not that NAL) */
    if (was_valid && W.valid && k < W.n && buf == W.base + W.ent[k].start && (uint64_t)size == W.ent[k].end - W.ent[k].start) {
        W.served = k + 1;
        W.synced = k + 1;
    }
    return r;
}

int read_hevc_nal_unit(hevc_stream_t* h, uint8_t* buf, int size)
{
    int stripped, r;
    legacy_lock();
    r = read_nal_w(h, buf, size, &stripped, 0);
    legacy_unlock();
    return r;
}

/*
 * hevc_stream.c:2343-3434: the same parse, printing "<byte>.<bits left>: <name>: <value>" per syntax
 * element to stdout while it reads.  The GPU parser logs (site, cursor, value) per element
 * (hbs_parse_headers_trace); the names come from hbs_trace_names.h.  As in the reference, the debug
 * reader is not the plain reader: it takes ONE bit for sub_layer_level_idc (hevc_stream.c:2939 against
 * :751), so streams with sub-layer levels parse differently from read_hevc_nal_unit -- kept.
 */
int read_debug_hevc_nal_unit(hevc_stream_t* h, uint8_t* buf, int size)
{
    int stripped, r;
    legacy_lock();
    r = read_nal_w(h, buf, size, &stripped, 1);
    legacy_unlock();
    return r;
}

/*
 * hevc_stream.c:1249-1327: serialise the struct *h holds for h->nal's type into buf.  The syntax
 * writers run on the GPU (hbs_write_headers) into an RBSP buffer of size * 3 / 4 bytes (:1265), then
 * rbsp_to_nal.  As in the reference: slices are written against h->pps / h->sps as they stand and
 * the derived RPS tables of the last SPS read or written; a slice write replaces h->slice_data's
 * payload by the zeros behind the header in the RBSP buffer (:1699-1706).
 */
static int write_hevc_nal_unit_unlocked(hevc_stream_t* h, uint8_t* buf, int size)
{
    hbs_parsed_nal p;
    hbs_written_nal w;
    const void* src = NULL;
    uint64_t src_bytes = 0;
    uint32_t cap;
    int rc, t, rbsp_size, nal_size;
    static hbs_written_nal* d_written = NULL;
    if (size < 0) return -1;
    if (need_ctx()) return -1;
    window_settle();                                                     /* the writers read (and an SPS rewrites) the tables in force */
    window_drop();
    cap = (uint32_t)((long)size * 3 / 4);
    need_bufs(16, (uint64_t)cap + 16);
    need_block(0);
    if (!d_written && (rc = hbs_dev_alloc(g_ctx, sizeof(hbs_written_nal), (void**)&d_written))) die("hbs_dev_alloc", rc);
    t = h->nal->nal_unit_type;
    if (t == HEVC_NAL_UNIT_TYPE_VPS_NUT) { src = h->vps; src_bytes = sizeof(hevc_vps_t); }
    else if (t == HEVC_NAL_UNIT_TYPE_SPS_NUT) { src = h->sps; src_bytes = sizeof(hevc_sps_t); }
    else if (t == HEVC_NAL_UNIT_TYPE_PPS_NUT) { src = h->pps; src_bytes = sizeof(hevc_pps_t); }
    else if (is_slice(t)) { src = h->sh; src_bytes = sizeof(hevc_slice_header_t); }
    else return -1;                                                      /* :1306 */
    /* parameter sets in force: what the object holds (the SPS's derived tables stay on the device) */
    sync_context(h);
    /* the NAL's descriptor and, unless it is an SPS (written in place in its slot, so that the tables it derives replace
     * the ones in force), its struct: through the pinned staging buffer, no wait */
    p.struct_off = 0;
    p.rc = 0; p.nal_unit_type = t; p.nal_layer_id = h->nal->nal_layer_id; p.nal_temporal_id_plus1 = h->nal->nal_temporal_id_plus1;
    p.slice_data_size = 0; p.slice_data_off = 0;
    need_stage(256 + src_bytes);
    memcpy(g_hstage, &p, sizeof(p));
    if ((rc = hbs_copy_to_device_async(g_ctx, g_dparsed, g_hstage, sizeof(p)))) die("hbs_copy_to_device_async", rc);
    if (t != HEVC_NAL_UNIT_TYPE_SPS_NUT) {
        memcpy(g_hstage + 256, src, src_bytes);
        if ((rc = hbs_copy_to_device_async(g_ctx, g_dstruct, g_hstage + 256, src_bytes))) die("hbs_copy_to_device_async", rc);
    }
    if ((rc = hbs_write_headers(g_ctx, g_dparsed, 1, t == HEVC_NAL_UNIT_TYPE_SPS_NUT ? g_dsps_slot : g_dstruct,
                                g_dsps_slot, g_dpps, g_dout, cap, d_written))) die("hbs_write_headers", rc);
    if ((rc = hbs_copy_to_host(g_ctx, &w, d_written, sizeof(w)))) die("hbs_copy_to_host", rc);
    if (is_slice(t) && h->slice_data) {                                  /* :1699-1706 */
        free(h->slice_data->rbsp_buf);
        h->slice_data->rbsp_size = w.slice_data_size;
        h->slice_data->rbsp_buf = w.slice_data_size > 0 ? (uint8_t*)calloc(1, (size_t)w.slice_data_size) : NULL;
    }
    if (w.rc < 0) return -1;                                             /* :1312 */
    rbsp_size = (int)w.rbsp_size;
    if (rbsp_size <= 0) return 0;
    /* rbsp_to_nal (:1319) on the RBSP where the writers left it */
    need_bufs(64, 0);
    nal_size = emit_one(NULL, g_dout, rbsp_size, buf);
    return nal_size;
}

int write_hevc_nal_unit(hevc_stream_t* h, uint8_t* buf, int size)
{
    int r;
    legacy_lock();
    r = write_hevc_nal_unit_unlocked(h, buf, size);
    legacy_unlock();
    return r;
}
