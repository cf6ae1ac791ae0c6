/*
 * hbs_capi.hip -- the extern "C" boundary (include/hevcbitstream_amd.h).
 * Thin: owns the per-GPU context (HIP stream, look-back workspace) and turns
 * each call into kernel launches.  No CPU implementation of any entry point
 * exists here: without a gfx950 device the context cannot be created.
 */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include "hbs_scan.h"
#include "hbs_emit_launch.h"
#include "hbs_emit.h"
#include "hbs_parse_launch.h"
#include "hbs_hdrwin.h"
#include "hbs_parse.h"
#include "hbs_parse_compact.h"

#ifndef HBS_DEFAULT_KERNEL
#define HBS_DEFAULT_KERNEL 0
#define HBS_DEFAULT_SCHED 1
#endif

constexpr int kTimingRing = 64;       /* timed calls whose event pairs are kept (hbs_ctx_kernel_ms_back) */

struct hbs_ctx {
    int device;
    hipStream_t own_stream;
    hipStream_t stream;
    int grid_blocks;
    int blocks_per_cu;
    int grid_blocks4, blocks_per_cu4;   /* event-sparse kernel */
    int grid_blocks6, grid_full6, blocks_per_cu6;   /* ... its 24-row geometry (variant 6) */
    int grid_full, grid_full4;          /* ... what the GPU holds; grid_blocks / grid_blocks4 may be cut (hbs_ctx_reserve_workgroups) */
    int grid_env, spare_wgs;            /* HBS_GRID_BLOCKS (0: unset); workgroup slots left free for other streams' kernels */
    int exclusive;                      /* hbs_ctx_set_device_exclusive: no other persistent kernel shares the device */
    int variant;                  /* 0 = automatic */
    int last_variant;             /* the kernel the last hbs_index_extract ran (automatic mode: once read back) */
    int probe_pending;
    int last_index_only;          /* the last hbs_index_extract had no arena: its sparse kernel is the streaming one (5) */
    int parse_sequential;         /* hbs_ctx_set_sequential_parse */
    int count_ahead;              /* hbs_ctx_set_count_ahead: 0 never, 1 streams from 3 GiB up, 2 always */
    uint64_t ingest_window_max;   /* hbs_ctx_set_ingest_window_max (0: the default) */
    void* attachment;             /* state another translation unit keeps with the context (the windowed ingest's buffers) */
    void (*attachment_free)(void*);
    int emit_blocks, emit_two_pass;   /* K3: resident workgroups of the single-pass kernel; 1 = use the older three-step path */
    int emit_tile_blocks, emit_tiles; /* ... of the arena-tile kernel; 0 never / 1 when eligible / 2 pinned */
    int emit_path_set;                /* hbs_ctx_set_emit_path was called: the environment no longer decides */
    const uint32_t* last_emit_tflag;  /* the last hbs_emit_annexb's verdict words: a COPY in emit_verdict (the words themselves live in the shared
                                         workspace, which the next call of any kind overwrites or reallocates); null: no verdict (small path) */
    uint32_t* emit_verdict;           /* 16 bytes of device memory owned by the context */
    uint32_t emit_calls;              /* hbs_emit_annexb calls so far: stamps the dense tiles counted ahead (never 0) */
    int sched;
    unsigned long long* desc;
    uint64_t desc_tiles;
    hbs::RunHeader* hdr;
    uint8_t* tail;                      /* padded copy of the stream's last tile (event-sparse kernel) */
    /* K3 / generator workspace */
    void* ws; uint64_t ws_bytes;
    void* ahead; uint64_t ahead_tiles;   /* K12's dense tiles counted ahead: a table entry and a byte per 192 KiB tile (streams from 3 GiB up) */
    void* ws2; uint64_t ws2_bytes;   /* hbs_index_parse: header windows and the index that points into them (alive across the parse, which carves ws) */
    uint8_t* zeros;              /* sizeof(hevc_sps_t) zero bytes: the "no parameter set yet" structs */
    /* optional timing of the dominant kernel */
    int timing; hipEvent_t ev0, ev1; int ev_valid;        /* ev0 / ev1: the slot of the ring the last call used */
    hipEvent_t ring0[kTimingRing], ring1[kTimingRing];    /* event pairs of the last kTimingRing timed calls */
    unsigned long long timed_calls;
    char err[256];
};

namespace {

int fail(hbs_ctx* c, hipError_t e, const char* what)
{
    if (c) snprintf(c->err, sizeof(c->err), "%s: %s", what, hipGetErrorString(e));
    return HBS_E_HIP;
}

int ensure_workspace(hbs_ctx* c, uint64_t stream_bytes)
{
    const uint64_t tiles = (stream_bytes + hbs::kTileBytes - 1) / hbs::kTileBytes + 1;
    if (tiles > c->desc_tiles) {
        if (c->desc) { (void)hipFree(c->desc); c->desc = nullptr; c->desc_tiles = 0; }
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->desc), tiles * 16);
        if (e != hipSuccess) return fail(c, e, "hipMalloc(look-back descriptors)");
        c->desc_tiles = tiles;
    }
    return 0;
}

int ensure_ws(hbs_ctx* c, uint64_t bytes)
{
    if (bytes > c->ws_bytes) {
        if (c->ws) { (void)hipStreamSynchronize(c->stream); (void)hipFree(c->ws); c->ws = nullptr; c->ws_bytes = 0; }
        hipError_t e = hipMalloc(&c->ws, bytes);
        if (e != hipSuccess) return fail(c, e, "hipMalloc(workspace)");
        c->ws_bytes = bytes;
    }
    return 0;
}

uint64_t round256(uint64_t v) { return (v + 255) & ~255ull; }

} // namespace

extern "C" {

const char* hbs_version(void)
{
    return "hevcbitstream_amd 0.6 (gfx950 HIP; K12 fused scan/index/extract)";
}

int hbs_ctx_create(hbs_ctx** out, int device)
{
    if (!out) return HBS_E_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return HBS_E_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return HBS_E_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return HBS_E_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        fprintf(stderr, "hevcbitstream_amd: device %d is %s; this library carries gfx950 code only\n", device, prop.gcnArchName);
        return HBS_E_NO_DEVICE;
    }
    hbs_ctx* c = new (std::nothrow) hbs_ctx();
    if (!c) return HBS_E_HIP;
    memset(c, 0, sizeof(*c));
    c->device = device;
    hipError_t e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return HBS_E_HIP; }
    c->stream = c->own_stream;
    e = hipMalloc(reinterpret_cast<void**>(&c->hdr), sizeof(hbs::RunHeader));
    if (e != hipSuccess) { (void)hipStreamDestroy(c->own_stream); delete c; return HBS_E_HIP; }
    c->tail = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&c->tail), (size_t)hbs::scan4_tail_bytes()) != hipSuccess) { (void)hipFree(c->hdr); (void)hipStreamDestroy(c->own_stream); delete c; return HBS_E_HIP; }
    c->grid_blocks = hbs::scan_grid_blocks(device, &c->blocks_per_cu);
    if (c->grid_blocks <= 0) { (void)hipFree(c->hdr); (void)hipStreamDestroy(c->own_stream); delete c; return HBS_E_HIP; }
    c->grid_blocks4 = hbs::scan4_grid_blocks(device, &c->blocks_per_cu4);
    if (c->grid_blocks4 <= 0) { (void)hipFree(c->hdr); (void)hipStreamDestroy(c->own_stream); delete c; return HBS_E_HIP; }
    c->grid_blocks6 = hbs::scan4r24_grid_blocks(device, &c->blocks_per_cu6);
    if (c->grid_blocks6 <= 0) { (void)hipFree(c->hdr); (void)hipStreamDestroy(c->own_stream); delete c; return HBS_E_HIP; }
    c->grid_full = c->grid_blocks; c->grid_full4 = c->grid_blocks4; c->grid_full6 = c->grid_blocks6;
    const char* g = getenv("HBS_GRID_BLOCKS");          /* debugging aid: 1 = fully sequential tiles */
    c->grid_env = (g && atoi(g) > 0) ? atoi(g) : 0;
    if (g && atoi(g) > 0 && atoi(g) < c->grid_blocks) c->grid_blocks = atoi(g);
    if (g && atoi(g) > 0 && atoi(g) < c->grid_blocks4) c->grid_blocks4 = atoi(g);
    if (g && atoi(g) > 0 && atoi(g) < c->grid_blocks6) c->grid_blocks6 = atoi(g);
    const char* ca = getenv("HBS_COUNT_AHEAD");
    c->count_ahead = (ca && ca[0] >= '0' && ca[0] <= '2') ? ca[0] - '0' : 1;
    const char* kv = getenv("HBS_KERNEL");              /* 0 automatic, 2 LDS-image, 4 event-sparse, 5 index-only streaming, 6 event-sparse with 24 rows */
    const char* sv = getenv("HBS_SCHED");
    c->sched = (sv && atoi(sv) >= 0 && atoi(sv) <= 2) ? atoi(sv) : HBS_DEFAULT_SCHED;
    c->variant = (kv && (atoi(kv) == 0 || atoi(kv) == 2 || atoi(kv) == 4 || atoi(kv) == 5 || atoi(kv) == 6)) ? atoi(kv) : HBS_DEFAULT_KERNEL;
    c->last_variant = c->variant ? c->variant : 4;
    *out = c;
    return 0;
}

void hbs_ctx_destroy(hbs_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->attachment && c->attachment_free) c->attachment_free(c->attachment);
    if (c->desc) (void)hipFree(c->desc);
    if (c->hdr) (void)hipFree(c->hdr);
    if (c->tail) (void)hipFree(c->tail);
    if (c->ws) (void)hipFree(c->ws);
    if (c->ahead) (void)hipFree(c->ahead);
    if (c->zeros) (void)hipFree(c->zeros);
    if (c->emit_verdict) (void)hipFree(c->emit_verdict);
    if (c->ws2) (void)hipFree(c->ws2);
    if (c->ring0[0]) for (int i = 0; i < kTimingRing; ++i) { (void)hipEventDestroy(c->ring0[i]); (void)hipEventDestroy(c->ring1[i]); }
    (void)hipStreamDestroy(c->own_stream);
    delete c;
}

/* internal (hbs_place.hip): what hbs_last_error will say; not exported */
__attribute__((visibility("hidden"))) void hbs_ctx_set_error(hbs_ctx* c, const char* what, int hip_error)
{
    if (c) snprintf(c->err, sizeof(c->err), "%s: %s", what, hipGetErrorString((hipError_t)hip_error));
}
__attribute__((visibility("hidden"))) uint64_t hbs_ctx_ingest_window_max(hbs_ctx* c) { return c ? c->ingest_window_max : 0; }
/* internal (hbs_ingest.hip): one object kept alive with the context, freed with it; not exported */
__attribute__((visibility("hidden"))) void* hbs_ctx_attachment(hbs_ctx* c) { return c ? c->attachment : nullptr; }
__attribute__((visibility("hidden"))) void hbs_ctx_attach(hbs_ctx* c, void* p, void (*free_fn)(void*))
{
    if (!c) return;
    if (c->attachment && c->attachment_free && c->attachment != p) c->attachment_free(c->attachment);
    c->attachment = p; c->attachment_free = free_fn;
}

int hbs_ctx_set_stream(hbs_ctx* c, void* s)
{
    if (!c) return HBS_E_ARG;
    c->stream = reinterpret_cast<hipStream_t>(s);     /* NULL = the HIP null stream */
    return 0;
}

int hbs_ctx_enable_timing(hbs_ctx* c, int on)
{
    if (!c) return HBS_E_ARG;
    if (on && !c->ring0[0]) {
        if (hipSetDevice(c->device) != hipSuccess) return HBS_E_NO_DEVICE;
        for (int i = 0; i < kTimingRing; ++i)
            if (hipEventCreate(&c->ring0[i]) != hipSuccess || hipEventCreate(&c->ring1[i]) != hipSuccess) {
                /* all or nothing: ring0[0] != null means "the whole ring exists" everywhere else */
                for (int j = 0; j <= i; ++j) {
                    if (c->ring0[j]) (void)hipEventDestroy(c->ring0[j]);
                    if (c->ring1[j]) (void)hipEventDestroy(c->ring1[j]);
                    c->ring0[j] = nullptr; c->ring1[j] = nullptr;
                }
                c->timing = 0; c->ev_valid = 0;
                return HBS_E_HIP;
            }
    }
    c->timing = on ? 1 : 0;
    c->ev_valid = 0;
    c->timed_calls = 0;
    return 0;
}

/* the timed call `back` calls ago (0 = the last one); the ring keeps kTimingRing of them */
int hbs_ctx_kernel_ms_back(hbs_ctx* c, int back, float* ms)
{
    if (!c || !ms || !c->ev_valid || back < 0 || back >= kTimingRing || (unsigned long long)back >= c->timed_calls) return HBS_E_ARG;
    const int slot = (int)((c->timed_calls - 1 - (unsigned long long)back) % kTimingRing);
    hipError_t e = hipEventSynchronize(c->ring1[slot]);
    if (e != hipSuccess) return fail(c, e, "hipEventSynchronize");
    e = hipEventElapsedTime(ms, c->ring0[slot], c->ring1[slot]);
    return e == hipSuccess ? 0 : fail(c, e, "hipEventElapsedTime");
}

int hbs_ctx_kernel_ms(hbs_ctx* c, float* ms)
{
    if (!c || !ms || !c->ev_valid) return HBS_E_ARG;
    hipError_t e = hipEventSynchronize(c->ev1);
    if (e != hipSuccess) return fail(c, e, "hipEventSynchronize");
    e = hipEventElapsedTime(ms, c->ev0, c->ev1);
    return e == hipSuccess ? 0 : fail(c, e, "hipEventElapsedTime");
}

/* The scan kernels are persistent: their workgroups fill the GPU (the event-sparse kernel's 512 use every register), so a kernel
 * of another stream -- RCCL's, in the index gather of the multi-GPU path -- finds no CU until the scan ends, and a "pipelined"
 * exchange runs in the gaps between scans.  Leaving a few workgroup slots free lets it run beside the scan. */
int hbs_ctx_reserve_workgroups(hbs_ctx* c, int spare)
{
    if (!c || spare < 0) return HBS_E_ARG;
    c->spare_wgs = spare;
    c->grid_blocks = c->grid_full - spare > 1 ? c->grid_full - spare : 1;
    c->grid_blocks4 = c->grid_full4 - spare > 1 ? c->grid_full4 - spare : 1;
    c->grid_blocks6 = c->grid_full6 - spare > 1 ? c->grid_full6 - spare : 1;
    /* the HBS_GRID_BLOCKS debugging cap is a ceiling of its own: reserving workgroups never raises it */
    if (c->grid_env > 0 && c->grid_env < c->grid_blocks) c->grid_blocks = c->grid_env;
    if (c->grid_env > 0 && c->grid_env < c->grid_blocks4) c->grid_blocks4 = c->grid_env;
    if (c->grid_env > 0 && c->grid_env < c->grid_blocks6) c->grid_blocks6 = c->grid_env;
    return 0;
}

/* Default 0: every tile of the persistent kernels (scan + extraction, arena-tile emit) comes by ticket, which needs no assumption
 * about what else runs on the device.  1: the caller says this context's calls are the only persistent kernels on the device while
 * they run; a workgroup's first tile is then its number (no queue of 512 atomics on one address at the start of a call: ~1 % of a
 * 1 GiB call).  Two contexts or processes that scan ONE device at the same time must leave it at 0: with static first tiles each
 * could hold workgroup slots the other's low-numbered workgroups need, and wait until the look-back guard gives up (HBS_E_TIMEOUT). */
int hbs_ctx_set_device_exclusive(hbs_ctx* c, int on)
{
    if (!c || (on != 0 && on != 1)) return HBS_E_ARG;
    c->exclusive = on;
    return 0;
}

int hbs_ctx_grid(hbs_ctx* c, int* blocks, int* blocks_per_cu)
{
    if (!c) return HBS_E_ARG;
    const int v = c->variant ? c->variant : c->last_variant;
    if (blocks) *blocks = (v == 4) ? c->grid_blocks4 : (v == 6) ? c->grid_blocks6 : c->grid_blocks;
    if (blocks_per_cu) *blocks_per_cu = (v == 4) ? c->blocks_per_cu4 : (v == 6) ? c->blocks_per_cu6 : c->blocks_per_cu;
    return 0;
}

int hbs_ctx_set_kernel(hbs_ctx* c, int variant)
{
    if (!c || variant < 0 || variant == 1 || variant == 3 || variant > 6) return HBS_E_ARG;
    c->variant = variant;
    if (variant) c->last_variant = variant;
    return 0;
}

int hbs_ctx_get_kernel(hbs_ctx* c) { return c ? c->variant : HBS_E_ARG; }

int hbs_ctx_set_count_ahead(hbs_ctx* c, int mode)
{
    if (!c || mode < 0 || mode > 2) return HBS_E_ARG;
    c->count_ahead = mode;
    return 0;
}
int hbs_ctx_device(hbs_ctx* c) { return c ? c->device : HBS_E_ARG; }

int hbs_ctx_set_sequential_parse(hbs_ctx* c, int on)
{
    if (!c) return HBS_E_ARG;
    c->parse_sequential = on ? 1 : 0;
    return 0;
}

int hbs_ctx_set_ingest_window_max(hbs_ctx* c, uint64_t max_window_bytes)
{
    if (!c) return HBS_E_ARG;
    c->ingest_window_max = max_window_bytes;
    return 0;
}

int hbs_ctx_set_emit_path(hbs_ctx* c, int path)
{
    if (!c || path < -1 || path > 2) return HBS_E_ARG;
    /* -1: everything picked per call; 0: the item kernel (k3_fused); 1: count / scan / emit; 2: the arena-tile kernel whenever the
     * index allows it (the item kernel when it does not) */
    c->emit_two_pass = path == 2 ? 0 : path;
    c->emit_tiles = path == 2 ? 2 : (path == -1 ? 1 : 0);
    c->emit_path_set = 1;
    return 0;
}

/* 1: the arena-tile kernel did the whole of the last hbs_emit_annexb (eligible index, no tile handed over); 0: another path; waits */
int hbs_ctx_last_emit_by_tiles(hbs_ctx* c)
{
    if (!c) return HBS_E_ARG;
    if (!c->last_emit_tflag) return 0;
    if (hipSetDevice(c->device) != hipSuccess) return HBS_E_NO_DEVICE;
    uint32_t f[4] = {0, 0, 0, 0};
    hipError_t e = hipMemcpyAsync(f, c->last_emit_tflag, sizeof(f), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(c, e, "read-back of the emit verdict");
    return (f[1] == 1u && f[0] == 0u && f[3] == 0u && f[2] == 0u) ? 1 : 0;
}

int hbs_ctx_last_kernel(hbs_ctx* c)
{
    if (!c) return HBS_E_ARG;
    if (c->variant) return (c->variant == 5 && !c->last_index_only) ? 4 : c->variant;
    if (!c->probe_pending) return c->last_variant;
    /* automatic mode: the choice was made on the device; read the probe's counts back */
    if (hipSetDevice(c->device) != hipSuccess) return HBS_E_NO_DEVICE;
    hbs::RunHeader h;
    hipError_t e = hipMemcpyAsync(&h, c->hdr, sizeof(h), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(c, e, "read-back of the density probe");
    uint64_t chunks = 0, flagged = 0;
    for (int i = 0; i < 64; ++i) { chunks += h.probe_slot[i][0]; flagged += h.probe_slot[i][1]; }
    c->last_variant = hbs::probe_variant((uint32_t)chunks, (uint32_t)flagged, c->last_index_only != 0);
    c->probe_pending = 0;
    return c->last_variant;
}

int hbs_ctx_use_own_stream(hbs_ctx* c)
{
    if (!c) return HBS_E_ARG;
    c->stream = c->own_stream;
    return 0;
}

void* hbs_ctx_get_stream(hbs_ctx* c) { return c ? reinterpret_cast<void*>(c->stream) : nullptr; }

int hbs_ctx_synchronize(hbs_ctx* c)
{
    if (!c) return HBS_E_ARG;
    hipError_t e = hipStreamSynchronize(c->stream);
    return e == hipSuccess ? 0 : fail(c, e, "hipStreamSynchronize");
}

const char* hbs_last_error(hbs_ctx* c) { return c ? c->err : "no context"; }

uint64_t hbs_ctx_device_bytes(hbs_ctx* c)
{
    if (!c) return 0;
    const uint64_t zb = c->zeros ? ((sizeof(hevc_sps_t) + 255) & ~(uint64_t)255) : 0;
    return c->desc_tiles * 16 + sizeof(hbs::RunHeader) + hbs::scan4_tail_bytes() + c->ws_bytes + c->ws2_bytes + (c->ahead_tiles ? 64 + c->ahead_tiles * hbs::scan4_ahead_entry_bytes() : 0) + zb;
}

uint64_t hbs_workspace_bytes(uint64_t stream_bytes)
{
    return ((stream_bytes + hbs::kTileBytes - 1) / hbs::kTileBytes + 1) * 16 + sizeof(hbs::RunHeader);
}

int hbs_index_extract(hbs_ctx* c, const uint8_t* d_stream, uint64_t n,
                      hbs_nal_entry* d_index, uint64_t index_cap,
                      uint8_t* d_rbsp, uint64_t rbsp_cap, hbs_summary* d_summary)
{
    if (!c || !d_summary || (n && !d_stream) || (index_cap && !d_index)) return HBS_E_ARG;
    if ((reinterpret_cast<uintptr_t>(d_stream) & 15) || (reinterpret_cast<uintptr_t>(d_rbsp) & 15) ||
        (reinterpret_cast<uintptr_t>(d_index) & 7)) {
        snprintf(c->err, sizeof(c->err), "stream/rbsp pointers must be 16-byte aligned");
        return HBS_E_ARG;
    }
    if (hipSetDevice(c->device) != hipSuccess) return HBS_E_NO_DEVICE;
    int rc = ensure_workspace(c, n);
    if (rc) return rc;
    hbs::ScanArgs a;
    a.stream = d_stream; a.n = n;
    a.index = d_index; a.index_cap = index_cap;
    a.rbsp = d_rbsp; a.rbsp_cap = d_rbsp ? rbsp_cap : 0;
    a.desc = c->desc; a.hdr = c->hdr; a.tail = c->tail; a.summary = d_summary;
    a.ws5 = nullptr;
    if (!d_rbsp && n) {                                       /* the index-only kernels keep tile aggregates and elements between their passes */
        rc = ensure_ws(c, hbs::scan5_workspace_bytes(n));
        if (rc) return rc;
        a.ws5 = c->ws;
    }
    a.ahead_cand = nullptr; a.ahead_tab = nullptr; a.ahead_list = nullptr; a.ahead_ctl = nullptr;
    if (d_rbsp && (c->count_ahead == 2 || (c->count_ahead == 1 && hbs::scan4_counts_ahead(n))) && n > (uint64_t)hbs::scan4_tile_bytes() &&
        (c->variant == 0 || c->variant == 4 || c->variant == 5) /* (6: the 24-row geometry counts nothing ahead) */ && !hbs::scan_takes_small_path(n, index_cap, c->variant)) {
        /* K12's dense tiles counted ahead: [AheadCtl | table | a word per tile | list] */
        const uint64_t tiles = (n + (uint64_t)hbs::scan4_tile_bytes() - 1) / (uint64_t)hbs::scan4_tile_bytes();
        if (tiles > c->ahead_tiles) {
            if (c->ahead) { (void)hipStreamSynchronize(c->stream); (void)hipFree(c->ahead); c->ahead = nullptr; c->ahead_tiles = 0; }
            hipError_t e = hipMalloc(&c->ahead, 64 + tiles * hbs::scan4_ahead_entry_bytes());
            if (e != hipSuccess) return fail(c, e, "hipMalloc(count-ahead table)");
            /* the call number, the list's count and the tiles' words start at zero (table and list are written before they are read) */
            e = hipMemsetAsync(c->ahead, 0, 64 + tiles * 72, c->stream);
            if (e != hipSuccess) return fail(c, e, "hipMemsetAsync(count-ahead words)");
            c->ahead_tiles = tiles;
        }
        uint8_t* const p = static_cast<uint8_t*>(c->ahead);
        a.ahead_ctl = reinterpret_cast<hbs::AheadCtl*>(p);
        a.ahead_tab = p + 64;
        a.ahead_cand = reinterpret_cast<unsigned long long*>(p + 64 + c->ahead_tiles * 64);
        a.ahead_list = reinterpret_cast<uint32_t*>(p + 64 + c->ahead_tiles * 72);
    }
    c->last_index_only = (hbs::scan_uses_index_only(n, c->variant, d_rbsp) && !hbs::scan_takes_small_path(n, index_cap, c->variant)) ? 1 : 0;
    a.variant = c->variant;
    a.sched = c->sched;
    a.grid_blocks = c->grid_blocks; a.grid_blocks4 = c->grid_blocks4; a.grid_blocks4r24 = c->grid_blocks6; a.spare_wgs = c->spare_wgs; a.first_static = c->exclusive;
    c->probe_pending = (c->variant == 0 && n) ? 1 : 0;
    if (hbs::scan_takes_small_path(n, index_cap, c->variant)) { c->probe_pending = 0; c->last_variant = 2; }
    if (c->timing && n) {                                     /* this call's slot of the ring */
        const int slot = (int)(c->timed_calls % kTimingRing);
        c->ev0 = c->ring0[slot]; c->ev1 = c->ring1[slot];
        c->timed_calls += 1;
    }
    a.ev_begin = (c->timing && n) ? c->ev0 : nullptr;
    a.ev_end = (c->timing && n) ? c->ev1 : nullptr;
    c->ev_valid = (c->timing && n) ? 1 : (c->ev_valid && c->timing);
    hipError_t e = hbs::launch_scan_extract(a, c->stream);
    return e == hipSuccess ? 0 : fail(c, e, "launch_scan_extract");
}

int hbs_emit_annexb(hbs_ctx* c, const uint8_t* d_rbsp, uint64_t rbsp_bytes,
                    const hbs_nal_entry* d_index_in, uint64_t n_nals, int gap_mode,
                    uint8_t* d_out, uint64_t out_cap, hbs_nal_entry* d_index_out, hbs_summary* d_summary)
{
    if (!c || !d_summary || (n_nals && (!d_rbsp || !d_index_in || !d_out))) return HBS_E_ARG;
    if (hipSetDevice(c->device) != hipSuccess) return HBS_E_NO_DEVICE;
    const uint64_t b_seg = 8192, b_n = round256((n_nals + 1) * 8);       /* b_seg: the scan's 1024 partial sums */
    /* NALs that do not overlap add up to at most rbsp_bytes; an index whose NALs add up to more gets HBS_E_CAPACITY */
    const uint64_t items_cap = hbs::emit_items_bound(n_nals, out_cap < rbsp_bytes ? out_cap : rbsp_bytes);
    const uint64_t b_items = round256(items_cap * 8), b_desc = round256(hbs::emit_desc_words(items_cap) * 8);
    const uint64_t first_cap = rbsp_bytes / (192u * 1024u) + 4;          /* arena tiles of the tile kernel (hbs_emit.hip: kTTileBytes) */
    const uint64_t b_first = round256(first_cap * 8);
    const uint64_t b_cand = round256(first_cap * 4), b_dz = round256(first_cap * hbs::emit_dz_table_words() * 4);   /* dense tiles counted ahead */
    int rc = ensure_ws(c, b_seg + 2 * b_n + b_items + b_desc + 1024 + b_first + b_cand + b_dz);
    if (rc) return rc;
    if (c->emit_blocks <= 0) {
        c->emit_blocks = hbs::emit_grid_blocks(c->device);
        if (c->emit_blocks <= 0) return fail(c, hipErrorUnknown, "occupancy query of the emit kernel");
        const char* eb = getenv("HBS_EMIT_BLOCKS");         /* debugging aid */
        if (eb && atoi(eb) > 0 && atoi(eb) < c->emit_blocks) c->emit_blocks = atoi(eb);
        const char* tp = getenv("HBS_EMIT_TWO_PASS");        /* 1 / 0 pin a way; default: picked on the device */
        if (!c->emit_path_set) { c->emit_two_pass = !tp ? -1 : (atoi(tp) == 1 ? 1 : 0); c->emit_tiles = !tp ? 1 : 0; }
        const char* et = getenv("HBS_EMIT_TILES");          /* 0 / 1 / 2: never / when eligible / pinned */
        if (!c->emit_path_set && et && atoi(et) >= 0 && atoi(et) <= 2) c->emit_tiles = atoi(et);
        c->emit_tile_blocks = hbs::emit_tile_grid_blocks(c->device);
        if (c->emit_tile_blocks <= 0) return fail(c, hipErrorUnknown, "occupancy query of the arena-tile emit kernel");
    }
    uint8_t* w = static_cast<uint8_t*>(c->ws);
    hbs::EmitArgs a;
    a.rbsp = d_rbsp; a.rbsp_bytes = rbsp_bytes; a.index_in = d_index_in; a.n = n_nals; a.gap_mode = gap_mode;
    a.out = d_out; a.out_cap = out_cap; a.index_out = d_index_out; a.summary = d_summary;
    a.scan_tmp = reinterpret_cast<unsigned long long*>(w);
    a.nal_total = reinterpret_cast<unsigned long long*>(w + b_seg);
    a.out_off = reinterpret_cast<unsigned long long*>(w + b_seg + b_n);
    a.items = reinterpret_cast<unsigned long long*>(w + b_seg + 2 * b_n); a.items_cap = items_cap;
    a.desc = reinterpret_cast<unsigned long long*>(w + b_seg + 2 * b_n + b_items);
    uint8_t* tail = w + b_seg + 2 * b_n + b_items + b_desc;
    a.total = reinterpret_cast<unsigned long long*>(tail);
    a.err = reinterpret_cast<uint32_t*>(tail + 256);
    a.ticket = reinterpret_cast<uint32_t*>(tail + 512);
    a.n_items = reinterpret_cast<unsigned long long*>(tail + 768);
    a.total_dense = reinterpret_cast<unsigned long long*>(tail + 768 + 64);
    a.probe = reinterpret_cast<uint32_t*>(tail + 768 + 128);
    a.tflag = reinterpret_cast<uint32_t*>(tail + 768 + 192);
    a.first_k = reinterpret_cast<unsigned long long*>(tail + 1024); a.first_cap = first_cap;
    a.cand_count = reinterpret_cast<uint32_t*>(tail + 640); a.cand_ticket = reinterpret_cast<uint32_t*>(tail + 704);   /* cleared with the counters */
    a.cand_list = reinterpret_cast<uint32_t*>(tail + 1024 + b_first); a.cand_cap = first_cap;
    a.dz_table = reinterpret_cast<uint32_t*>(tail + 1024 + b_first + b_cand);
    /* dense tiles counted ahead of the tile kernel (k3t_sample + a pass over the tiles it lists): from 3 GiB of arena up, as the scan's
     * (hbs_ctx_set_count_ahead: 0 never, 1 from 3 GiB, 2 always).  Until round 6 every call paid for it -- two launches, 11.5 us of a
     * 1 GiB call's 445 with nothing listed -- where mixed content is as unlikely as in the scan. */
    if (c->count_ahead == 0 || (c->count_ahead == 1 && !hbs::scan4_counts_ahead(rbsp_bytes))) a.dz_table = nullptr;
    c->emit_calls += 1; if (c->emit_calls == 0) c->emit_calls = 1;
    a.call_no = c->emit_calls;
    a.first_static = c->exclusive;
    a.tiles = c->emit_tiles; a.tile_blocks = c->emit_tile_blocks;
    a.clear_bytes = b_desc + 1024;                          /* look-back words and the counters behind them */
    a.grid_blocks = c->emit_blocks; a.two_pass = c->emit_two_pass;
    if (!c->emit_verdict) {
        const hipError_t ea = hipMalloc(reinterpret_cast<void**>(&c->emit_verdict), 16);
        if (ea != hipSuccess) return fail(c, ea, "hipMalloc(emit verdict)");
    }
    /* the verdict words leave the shared workspace with the call's last kernel (round 4's advice: hbs_ctx_last_emit_by_tiles after
     * any other call read bytes that call had overwritten, or a freed workspace); the one-launch small path writes none */
    a.verdict_out = c->emit_verdict;
    c->last_emit_tflag = hbs::emit_takes_small_path(a.n, a.rbsp_bytes, a.two_pass) ? nullptr : c->emit_verdict;
    hipError_t e = hbs::launch_emit_annexb(a, c->stream);
    return e == hipSuccess ? 0 : fail(c, e, "launch_emit_annexb");
}

int hbs_synth_rbsp(hbs_ctx* c, uint64_t seed, uint64_t n_nals, int mode,
                   uint8_t* d_rbsp, uint64_t rbsp_cap, hbs_nal_entry* d_index, hbs_summary* d_summary)
{
    if (!c || !d_summary || (n_nals && (!d_rbsp || !d_index)) || (mode != 0 && mode != 1)) return HBS_E_ARG;
    if (hipSetDevice(c->device) != hipSuccess) return HBS_E_NO_DEVICE;
    const uint64_t b_n = round256((n_nals + 1) * 8);
    int rc = ensure_ws(c, 2 * b_n + 512 + 8192);
    if (rc) return rc;
    uint8_t* w = static_cast<uint8_t*>(c->ws);
    hbs::SynthArgs a;
    a.seed = seed; a.n = n_nals; a.mode = mode; a.rbsp = d_rbsp; a.rbsp_cap = rbsp_cap; a.index = d_index; a.summary = d_summary;
    a.lens = reinterpret_cast<unsigned long long*>(w);
    a.offs = reinterpret_cast<unsigned long long*>(w + b_n);
    a.total = reinterpret_cast<unsigned long long*>(w + 2 * b_n);
    a.err = reinterpret_cast<uint32_t*>(w + 2 * b_n + 256);
    a.scan_tmp = reinterpret_cast<unsigned long long*>(w + 2 * b_n + 512);
    hipError_t e = hbs::launch_synth_rbsp(a, c->stream);
    return e == hipSuccess ? 0 : fail(c, e, "launch_synth_rbsp");
}

uint64_t hbs_sps_slot_bytes(void) { return hbs::slot_bytes_of(HEVC_NAL_UNIT_TYPE_SPS_NUT); }
uint64_t hbs_sps_tables_offset(void) { return hbs::round16(sizeof(hevc_sps_t)); }

uint64_t hbs_synth_rbsp_bound(uint64_t n_nals) { return n_nals * 12288ull + 16; }
uint64_t hbs_annexb_bound(uint64_t rbsp_bytes, uint64_t n_nals) { return rbsp_bytes + rbsp_bytes / 2 + 4 * n_nals + 16; }
/* gap_mode 0: the gaps are whatever the index says (a start code, plus any zero bytes that stood in front of it) */
uint64_t hbs_annexb_bound_gaps(uint64_t rbsp_bytes, uint64_t n_nals, uint64_t gap_bytes) { (void)n_nals; return rbsp_bytes + rbsp_bytes / 2 + gap_bytes + 16; }

int hbs_parse_headers(hbs_ctx* c, const uint8_t* d_rbsp, const hbs_nal_entry* d_index, uint64_t n_nals,
                      hbs_parsed_nal* d_parsed, uint8_t* d_structs, uint64_t structs_cap, hbs_summary* d_summary)
{
    return hbs_parse_headers_ctx(c, d_rbsp, d_index, n_nals, d_parsed, d_structs, structs_cap, nullptr, nullptr, d_summary);
}

int hbs_parse_extended(hbs_ctx* c, const uint8_t* d_rbsp, const hbs_nal_entry* d_index, uint64_t n_nals,
                       hbs_parsed_nal* d_parsed, hbs_ext_nal* d_ext)
{
    if (!c || (n_nals && (!d_rbsp || !d_index || !d_parsed || !d_ext))) return HBS_E_ARG;
    if (hipSetDevice(c->device) != hipSuccess) return HBS_E_NO_DEVICE;
    static_assert(sizeof(hbs_parsed_nal) == sizeof(hbs::ParsedNal), "hbs_parsed_nal is hbs::ParsedNal");
    const hipError_t e = hbs::launch_parse_extended(d_rbsp, d_index, n_nals, reinterpret_cast<hbs::ParsedNal*>(d_parsed), d_ext, c->stream);
    return e == hipSuccess ? 0 : fail(c, e, "hbs_parse_extended");
}

int hbs_parse_headers_ctx(hbs_ctx* c, const uint8_t* d_rbsp, const hbs_nal_entry* d_index, uint64_t n_nals,
                          hbs_parsed_nal* d_parsed, uint8_t* d_structs, uint64_t structs_cap,
                          const uint8_t* d_initial_sps_slot, const uint8_t* d_initial_pps, hbs_summary* d_summary)
{
    return hbs_parse_headers_trace(c, d_rbsp, d_index, n_nals, d_parsed, d_structs, structs_cap, d_initial_sps_slot, d_initial_pps,
                                   nullptr, 0, nullptr, d_summary);
}

int hbs_parse_headers_trace(hbs_ctx* c, const uint8_t* d_rbsp, const hbs_nal_entry* d_index, uint64_t n_nals,
                            hbs_parsed_nal* d_parsed, uint8_t* d_structs, uint64_t structs_cap,
                            const uint8_t* d_initial_sps_slot, const uint8_t* d_initial_pps,
                            hbs_trace_rec* d_trace, uint32_t trace_cap, uint32_t* d_trace_count, hbs_summary* d_summary)
{
    return hbs_parse_headers_state(c, d_rbsp, d_index, n_nals, d_parsed, d_structs, structs_cap, d_initial_sps_slot, d_initial_pps,
                                   d_trace, trace_cap, d_trace_count, d_summary, nullptr, nullptr);
}

static int parse_impl(hbs_ctx* c, const uint8_t* d_rbsp, const hbs_nal_entry* d_index, uint64_t n_nals,
                      hbs_parsed_nal* d_parsed, uint8_t* d_structs, uint64_t structs_cap,
                      const uint8_t* d_initial_sps_slot, const uint8_t* d_initial_pps,
                      hbs_trace_rec* d_trace, uint32_t trace_cap, uint32_t* d_trace_count, hbs_summary* d_summary,
                      uint8_t* d_state_sps_slot, uint8_t* d_state_pps,
                      hbs_slice_compact* d_compact, const uint64_t* d_want, uint64_t n_want);

int hbs_parse_headers_state(hbs_ctx* c, const uint8_t* d_rbsp, const hbs_nal_entry* d_index, uint64_t n_nals,
                            hbs_parsed_nal* d_parsed, uint8_t* d_structs, uint64_t structs_cap,
                            const uint8_t* d_initial_sps_slot, const uint8_t* d_initial_pps,
                            hbs_trace_rec* d_trace, uint32_t trace_cap, uint32_t* d_trace_count, hbs_summary* d_summary,
                            uint8_t* d_state_sps_slot, uint8_t* d_state_pps)
{
    return parse_impl(c, d_rbsp, d_index, n_nals, d_parsed, d_structs, structs_cap, d_initial_sps_slot, d_initial_pps,
                      d_trace, trace_cap, d_trace_count, d_summary, d_state_sps_slot, d_state_pps, nullptr, nullptr, 0);
}

int hbs_parse_headers_compact(hbs_ctx* c, const uint8_t* d_rbsp, const hbs_nal_entry* d_index, uint64_t n_nals,
                              hbs_parsed_nal* d_parsed, hbs_slice_compact* d_compact, uint8_t* d_structs, uint64_t structs_cap,
                              const uint8_t* d_initial_sps_slot, const uint8_t* d_initial_pps, hbs_summary* d_summary)
{
    if (!d_compact && n_nals) return HBS_E_ARG;
    return parse_impl(c, d_rbsp, d_index, n_nals, d_parsed, d_structs, structs_cap, d_initial_sps_slot, d_initial_pps,
                      nullptr, 0, nullptr, d_summary, nullptr, nullptr, d_compact, nullptr, 0);
}

int hbs_parse_materialize(hbs_ctx* c, const uint8_t* d_rbsp, const hbs_nal_entry* d_index, uint64_t n_nals,
                          hbs_parsed_nal* d_parsed, hbs_slice_compact* d_compact, uint8_t* d_structs, uint64_t structs_cap,
                          const uint8_t* d_initial_sps_slot, const uint8_t* d_initial_pps,
                          const uint64_t* d_nal_list, uint64_t n_list, hbs_summary* d_summary)
{
    if ((!d_compact && n_nals) || (n_list && !d_nal_list)) return HBS_E_ARG;
    return parse_impl(c, d_rbsp, d_index, n_nals, d_parsed, d_structs, structs_cap, d_initial_sps_slot, d_initial_pps,
                      nullptr, 0, nullptr, d_summary, nullptr, nullptr, d_compact, d_nal_list, n_list);
}

static int parse_impl(hbs_ctx* c, const uint8_t* d_rbsp, const hbs_nal_entry* d_index, uint64_t n_nals,
                      hbs_parsed_nal* d_parsed, uint8_t* d_structs, uint64_t structs_cap,
                      const uint8_t* d_initial_sps_slot, const uint8_t* d_initial_pps,
                      hbs_trace_rec* d_trace, uint32_t trace_cap, uint32_t* d_trace_count, hbs_summary* d_summary,
                      uint8_t* d_state_sps_slot, uint8_t* d_state_pps,
                      hbs_slice_compact* d_compact, const uint64_t* d_want, uint64_t n_want)
{
    static_assert(sizeof(hbs_slice_compact) == sizeof(hbs::SliceCompact), "public record == kernel record");
    if ((d_state_sps_slot == nullptr) != (d_state_pps == nullptr)) return HBS_E_ARG;
    if (d_state_sps_slot && (!d_structs || !n_nals)) return HBS_E_ARG;
    static_assert(sizeof(hbs_trace_rec) == sizeof(hbs::TraceRec), "public record == kernel record");
    static_assert(sizeof(hbs_parsed_nal) == sizeof(hbs::ParsedNal), "public record == kernel record");
    if (!c || !d_summary || (n_nals && (!d_rbsp || !d_index || !d_parsed))) return HBS_E_ARG;
    if (reinterpret_cast<uintptr_t>(d_structs) & 15) return HBS_E_ARG;
    if (hipSetDevice(c->device) != hipSuccess) return HBS_E_NO_DEVICE;
    if (!c->zeros) {
        const size_t zb = (sizeof(hevc_sps_t) + 255) & ~(size_t)255;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->zeros), zb);
        if (e != hipSuccess) return fail(c, e, "hipMalloc(zero structs)");
        e = hipMemsetAsync(c->zeros, 0, zb, c->stream);
        if (e != hipSuccess) return fail(c, e, "hipMemsetAsync(zero structs)");
    }
    const uint64_t b_n = round256((n_nals + 1) * 8);
    const uint64_t b_rows = round256(hbs::parse_own_rows_bytes(n_nals));
    /* the exact re-walk's records: three words per NAL, a summary word per 256, the temporaries of its lanes */
    const uint64_t b_n4 = round256((n_nals + 1) * 4), b_bsum = round256((n_nals / 256 + 2) * 4), b_fix = round256(hbs::parse_fix_temps_bytes());
    const uint64_t b_fixs = d_compact ? round256(hbs::parse_fix_structs_bytes()) : 0;     /* a compact parse's re-walk has no slots to walk into */
    const uint64_t fix_off = 3 * b_n + 512 + round256(1024 * 24) + b_rows;
    int rc = ensure_ws(c, fix_off + 3 * b_n4 + b_bsum + 256 + b_fix + b_fixs);
    if (rc) return rc;
    uint8_t* w = static_cast<uint8_t*>(c->ws);
    hbs::ParseArgs a;
    a.rbsp = d_rbsp; a.index = d_index; a.n = n_nals;
    a.parsed = reinterpret_cast<hbs::ParsedNal*>(d_parsed);
    a.structs = d_structs; a.structs_cap = d_structs ? structs_cap : 0; a.summary = d_summary;
    a.slot_size = reinterpret_cast<unsigned long long*>(w);
    a.ctx_sps = reinterpret_cast<long long*>(w + b_n);
    a.ctx_pps = reinterpret_cast<long long*>(w + 2 * b_n);
    a.zeros = c->zeros;
    a.initial_sps_slot = d_initial_sps_slot;
    a.initial_pps = d_initial_pps;
    a.total = reinterpret_cast<unsigned long long*>(w + 3 * b_n);
    a.err = reinterpret_cast<uint32_t*>(w + 3 * b_n + 256);
    a.div_flag = reinterpret_cast<uint32_t*>(w + 3 * b_n + 256 + 64);
    a.scan_tmp = w + 3 * b_n + 512;
    a.own_rows = reinterpret_cast<hbs::RpsRow*>(w + 3 * b_n + 512 + round256(1024 * 24));
    a.deps = reinterpret_cast<uint32_t*>(w + fix_off);
    a.wmask = reinterpret_cast<uint32_t*>(w + fix_off + b_n4);
    a.fix_list = reinterpret_cast<uint32_t*>(w + fix_off + 2 * b_n4);
    a.bsum = reinterpret_cast<uint32_t*>(w + fix_off + 3 * b_n4);
    a.fix_count = reinterpret_cast<uint32_t*>(w + fix_off + 3 * b_n4 + b_bsum);
    a.fix_temps = reinterpret_cast<hbs::RpsRow*>(w + fix_off + 3 * b_n4 + b_bsum + 256);
    a.trace = reinterpret_cast<hbs::TraceRec*>(d_trace); a.trace_cap = trace_cap; a.trace_count = d_trace_count;
    a.state_sps_slot_out = d_state_sps_slot; a.state_pps_out = d_state_pps;
    a.compact = reinterpret_cast<hbs::SliceCompact*>(d_compact); a.want_list = d_want; a.want_n = d_compact ? n_want : 0;
    a.fix_structs = d_compact ? w + fix_off + 3 * b_n4 + b_bsum + 256 + b_fix : nullptr;
    a.sequential = (d_state_sps_slot || d_compact) ? 0 : c->parse_sequential;
    hipError_t e = hbs::launch_parse_headers(a, c->stream);
    return e == hipSuccess ? 0 : fail(c, e, "launch_parse_headers");
}

static int index_parse_impl(hbs_ctx* c, const uint8_t* d_stream, uint64_t stream_bytes,
                            hbs_nal_entry* d_index, uint64_t index_cap, uint32_t header_window,
                            hbs_parsed_nal* d_parsed, hbs_slice_compact* d_compact, uint8_t* d_structs, uint64_t structs_cap, uint64_t* d_payload_off,
                            hbs_summary* d_scan_summary, hbs_summary* d_parse_summary, uint64_t* nal_count_out);

int hbs_index_parse(hbs_ctx* c, const uint8_t* d_stream, uint64_t stream_bytes,
                    hbs_nal_entry* d_index, uint64_t index_cap, uint32_t header_window,
                    hbs_parsed_nal* d_parsed, uint8_t* d_structs, uint64_t structs_cap, uint64_t* d_payload_off,
                    hbs_summary* d_scan_summary, hbs_summary* d_parse_summary, uint64_t* nal_count_out)
{
    return index_parse_impl(c, d_stream, stream_bytes, d_index, index_cap, header_window, d_parsed, nullptr, d_structs, structs_cap, d_payload_off,
                            d_scan_summary, d_parse_summary, nal_count_out);
}

int hbs_index_parse_compact(hbs_ctx* c, const uint8_t* d_stream, uint64_t stream_bytes,
                            hbs_nal_entry* d_index, uint64_t index_cap, uint32_t header_window,
                            hbs_parsed_nal* d_parsed, hbs_slice_compact* d_compact, uint8_t* d_structs, uint64_t structs_cap, uint64_t* d_payload_off,
                            hbs_summary* d_scan_summary, hbs_summary* d_parse_summary, uint64_t* nal_count_out)
{
    if (!d_compact) return HBS_E_ARG;
    return index_parse_impl(c, d_stream, stream_bytes, d_index, index_cap, header_window, d_parsed, d_compact, d_structs, structs_cap, d_payload_off,
                            d_scan_summary, d_parse_summary, nal_count_out);
}

static int index_parse_impl(hbs_ctx* c, const uint8_t* d_stream, uint64_t stream_bytes,
                            hbs_nal_entry* d_index, uint64_t index_cap, uint32_t header_window,
                            hbs_parsed_nal* d_parsed, hbs_slice_compact* d_compact, uint8_t* d_structs, uint64_t structs_cap, uint64_t* d_payload_off,
                            hbs_summary* d_scan_summary, hbs_summary* d_parse_summary, uint64_t* nal_count_out)
{
    if (!c || !d_scan_summary || !d_parse_summary || !d_index || !index_cap || !d_parsed) return HBS_E_ARG;
    if (header_window == 0) header_window = 512;
    if (header_window < 64 || header_window > (1u << 16) || (header_window & 15u)) return HBS_E_ARG;
    /* 1. find_nal_unit over the stream: no arena */
    int rc = hbs_index_extract(c, d_stream, stream_bytes, d_index, index_cap, nullptr, 0, d_scan_summary);
    if (rc) return rc;
    /* the parse's launches are sized by the number of NALs: the one wait of the call */
    hbs_summary s;
    rc = hbs_read_summary(c, d_scan_summary, &s);
    if (rc) return rc;
    if (nal_count_out) *nal_count_out = s.nal_count;
    if (s.error) return s.error;
    const uint64_t nals = s.nal_count;
    /* 2. the bytes the parse can look at, stripped into windows */
    hbs::HdrWinArgs a;
    /* everything below is sized by the NALs FOUND (known since the wait above), not by the caller's index capacity: a default
     * capacity of stream_bytes / 64 entries would ask for 137 GB of windows on a 16 GiB stream (round 3's advice) */
    const uint64_t slots = nals ? nals : 1;
    a.stream = d_stream; a.index = d_index; a.nals = nals; a.index_cap = slots; a.window = header_window;
    a.arena_bytes = hbs::hdrwin_arena_bytes(slots, header_window, stream_bytes);
    const uint64_t b_arena = round256(a.arena_bytes + 64), b_idx = round256(slots * sizeof(hbs_nal_entry));
    const uint64_t b_notes = round256(slots * 16);
    if (b_arena + b_idx + b_notes + 256 > c->ws2_bytes) {                 /* grow-only */
        if (c->ws2) { (void)hipStreamSynchronize(c->stream); (void)hipFree(c->ws2); c->ws2 = nullptr; c->ws2_bytes = 0; }
        const hipError_t e = hipMalloc(&c->ws2, b_arena + b_idx + b_notes + 256);
        if (e != hipSuccess) return fail(c, e, "hipMalloc(header windows)");
        c->ws2_bytes = b_arena + b_idx + b_notes + 256;
    }
    uint8_t* w = static_cast<uint8_t*>(c->ws2);
    a.arena = w; a.idx2 = reinterpret_cast<hbs_nal_entry*>(w + b_arena); a.bump = reinterpret_cast<unsigned long long*>(w + b_arena + b_idx);
    a.notes = w + b_arena + b_idx + 256;
    hipError_t e = hbs::launch_hdr_strip(a, c->stream);
    if (e != hipSuccess) return fail(c, e, "launch_hdr_strip");
    /* 3. K4 on the windows, 4. slice_data_size against the real lengths, windows that were too small reported */
    rc = d_compact ? hbs_parse_headers_compact(c, a.arena, a.idx2, nals, d_parsed, d_compact, d_structs, structs_cap, nullptr, nullptr, d_parse_summary)
                   : hbs_parse_headers(c, a.arena, a.idx2, nals, d_parsed, d_structs, structs_cap, d_parse_summary);
    if (rc) return rc;
    e = hbs::launch_hdr_fix(a, d_parsed, d_parse_summary, reinterpret_cast<unsigned long long*>(d_payload_off), c->stream, d_compact ? 1 : 0);
    return e == hipSuccess ? 0 : fail(c, e, "launch_hdr_fix");
}

int hbs_write_headers(hbs_ctx* c, const hbs_parsed_nal* d_parsed, uint64_t n_nals, uint8_t* d_structs,
                      const uint8_t* d_initial_sps_slot, const uint8_t* d_initial_pps,
                      uint8_t* d_rbsp_out, uint32_t rbsp_cap, hbs_written_nal* d_written)
{
    static_assert(sizeof(hbs_written_nal) == sizeof(hbs::WrittenNal), "public record == kernel record");
    if (!c || (n_nals && (!d_parsed || !d_structs || !d_rbsp_out || !d_written))) return HBS_E_ARG;
    if (hipSetDevice(c->device) != hipSuccess) return HBS_E_NO_DEVICE;
    if (!c->zeros) {
        const size_t zb = (sizeof(hevc_sps_t) + 255) & ~(size_t)255;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->zeros), zb);
        if (e != hipSuccess) return fail(c, e, "hipMalloc(zero structs)");
        e = hipMemsetAsync(c->zeros, 0, zb, c->stream);
        if (e != hipSuccess) return fail(c, e, "hipMemsetAsync(zero structs)");
    }
    const uint64_t b_n = round256((n_nals + 1) * 8);
    int rc = ensure_ws(c, 3 * b_n + 512 + round256(1024 * 24) + round256(hbs::parse_own_rows_bytes(n_nals)));
    if (rc) return rc;
    uint8_t* w = static_cast<uint8_t*>(c->ws);
    hbs::WriteArgs a;
    a.parsed = reinterpret_cast<const hbs::ParsedNal*>(d_parsed); a.n = n_nals; a.structs = d_structs;
    a.rbsp_out = d_rbsp_out; a.rbsp_cap = rbsp_cap; a.written = reinterpret_cast<hbs::WrittenNal*>(d_written);
    a.slot_size = reinterpret_cast<unsigned long long*>(w);
    a.ctx_sps = reinterpret_cast<long long*>(w + b_n);
    a.ctx_pps = reinterpret_cast<long long*>(w + 2 * b_n);
    a.zeros = c->zeros; a.initial_sps_slot = d_initial_sps_slot; a.initial_pps = d_initial_pps;
    a.total = reinterpret_cast<unsigned long long*>(w + 3 * b_n);
    a.scan_tmp = w + 3 * b_n + 512;
    a.own_rows = reinterpret_cast<hbs::RpsRow*>(w + 3 * b_n + 512 + round256(1024 * 24));
    hipError_t e = hbs::launch_write_headers(a, c->stream);
    return e == hipSuccess ? 0 : fail(c, e, "launch_write_headers");
}

/* plain device-memory helpers so that C callers (hbs_legacy.c) need no HIP headers */
int hbs_dev_alloc(hbs_ctx* c, uint64_t bytes, void** out)
{
    if (!c || !out) return HBS_E_ARG;
    if (hipSetDevice(c->device) != hipSuccess) return HBS_E_NO_DEVICE;
    hipError_t e = hipMalloc(out, bytes ? bytes : 16);
    return e == hipSuccess ? 0 : fail(c, e, "hipMalloc");
}

int hbs_dev_free(hbs_ctx* c, void* p)
{
    if (!c) return HBS_E_ARG;
    (void)hipStreamSynchronize(c->stream);
    hipError_t e = hipFree(p);
    return e == hipSuccess ? 0 : fail(c, e, "hipFree");
}

int hbs_copy_to_device(hbs_ctx* c, void* d_dst, const void* h_src, uint64_t bytes)
{
    if (!c) return HBS_E_ARG;
    if (!bytes) return 0;
    hipError_t e = hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);      /* the host buffer may be reused at once */
    return e == hipSuccess ? 0 : fail(c, e, "hipMemcpy(H2D)");
}

int hbs_copy_to_host(hbs_ctx* c, void* h_dst, const void* d_src, uint64_t bytes)
{
    if (!c) return HBS_E_ARG;
    if (!bytes) return 0;
    hipError_t e = hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    return e == hipSuccess ? 0 : fail(c, e, "hipMemcpy(D2H)");
}

/* the same without the wait: h_src must be page-locked (hbs_host_alloc) and stay untouched until the stream
 * has passed the copy; the legacy wrappers use it to put a whole call behind ONE synchronisation */
int hbs_copy_to_device_async(hbs_ctx* c, void* d_dst, const void* h_src, uint64_t bytes)
{
    if (!c) return HBS_E_ARG;
    if (!bytes) return 0;
    hipError_t e = hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c->stream);
    return e == hipSuccess ? 0 : fail(c, e, "hipMemcpyAsync(H2D)");
}

int hbs_copy_device(hbs_ctx* c, void* d_dst, const void* d_src, uint64_t bytes)
{
    if (!c) return HBS_E_ARG;
    if (!bytes) return 0;
    hipError_t e = hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, c->stream);
    return e == hipSuccess ? 0 : fail(c, e, "hipMemcpyAsync(D2D)");
}

int hbs_host_alloc(hbs_ctx* c, uint64_t bytes, void** out)
{
    if (!c || !out) return HBS_E_ARG;
    if (hipSetDevice(c->device) != hipSuccess) return HBS_E_NO_DEVICE;
    hipError_t e = hipHostMalloc(out, bytes ? bytes : 16, hipHostMallocDefault);
    return e == hipSuccess ? 0 : fail(c, e, "hipHostMalloc");
}

int hbs_host_free(hbs_ctx* c, void* p)
{
    if (!c) return HBS_E_ARG;
    if (!p) return 0;
    (void)hipStreamSynchronize(c->stream);
    hipError_t e = hipHostFree(p);
    return e == hipSuccess ? 0 : fail(c, e, "hipHostFree");
}

int hbs_fill_device(hbs_ctx* c, void* d_dst, int value, uint64_t bytes)
{
    if (!c) return HBS_E_ARG;
    if (!bytes) return 0;
    hipError_t e = hipMemsetAsync(d_dst, value, bytes, c->stream);
    return e == hipSuccess ? 0 : fail(c, e, "hipMemsetAsync");
}

int hbs_read_summary(hbs_ctx* c, const hbs_summary* d_summary, hbs_summary* h_summary)
{
    if (!c || !d_summary || !h_summary) return HBS_E_ARG;
    hipError_t e = hipMemcpyAsync(h_summary, d_summary, sizeof(hbs_summary), hipMemcpyDeviceToHost, c->stream);
    if (e != hipSuccess) return fail(c, e, "hipMemcpyAsync(summary)");
    e = hipStreamSynchronize(c->stream);
    return e == hipSuccess ? 0 : fail(c, e, "hipStreamSynchronize");
}

} // extern "C"
