/*
 * hbs_ingest.hip -- HIP backend of the windowed ingest (hbs_ingest.h) and its C entry point
 * hbs_index_extract_host: host stream in, host index + RBSP out, any length.
 *
 * Two device window buffers and two HIP streams: while the compute stream scans window w
 * (hbs_index_extract) and downloads its index and RBSP, the copy stream uploads the fresh bytes
 * of window w+1; the bytes window w+1 scans again are moved device-to-device behind the scan.
 * Host buffers that are page-locked (hipHostMalloc / hipHostRegister / torch pin_memory) make the
 * transfers truly asynchronous; pageable ones work, staged by the driver.
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <new>
#include "hbs_ingest.h"

extern "C" {
int hbs_ctx_set_stream(hbs_ctx* ctx, void* hip_stream);
void* hbs_ctx_get_stream(hbs_ctx* ctx);
/* internal: an object that lives and dies with the context (hbs_capi.hip) */
__attribute__((visibility("hidden"))) void* hbs_ctx_attachment(hbs_ctx* ctx);
__attribute__((visibility("hidden"))) void hbs_ctx_attach(hbs_ctx* ctx, void* p, void (*free_fn)(void*));
__attribute__((visibility("hidden"))) uint64_t hbs_ctx_ingest_window_max(hbs_ctx* ctx);
}

namespace {

struct HipBackend {
    hbs_ctx* ctx;
    const uint8_t* h_stream;
    uint8_t* h_rbsp;                   /* nullable */
    bool want_rbsp = false;
    uint64_t lead, window;
    uint64_t idx_cap;
    uint8_t* d_buf[2] = {nullptr, nullptr};
    uint64_t fresh[2] = {0, 0};
    uint8_t* d_rbsp = nullptr;
    hbs_nal_entry* d_index = nullptr;
    hbs_summary* d_summary = nullptr;
    hbs_nal_entry* h_stage = nullptr;  /* page-locked: a window's entries usually come down in one copy */
    static constexpr uint64_t kStageEntries = 65536;
    hipStream_t s_in = nullptr, s_cmp = nullptr;
    hipEvent_t up_done[2] = {nullptr, nullptr}, buf_free[2] = {nullptr, nullptr};
    void* saved_stream = nullptr;
    hipError_t err = hipSuccess;

    uint64_t lead_capacity() const { return lead; }
    uint64_t index_capacity() const { return idx_cap; }
    uint64_t fresh_len(int b) const { return fresh[b]; }
    hbs_nal_entry* staging() const { return h_stage; }
    uint64_t staging_entries() const { return kStageEntries; }

    int ok(hipError_t e) { if (e != hipSuccess) { err = e; return HBS_E_HIP; } return 0; }

    /* device buffers, streams and events: made once per (window size, RBSP wanted) and kept with the context --
     * a hipMalloc of gigabytes costs more than the transfer of a window */
    bool allocated = false;
    int allocate()
    {
        const uint64_t cap = lead + window + 256;
        for (int b = 0; b < 2; ++b) {
            if (int rc = ok(hipMalloc(reinterpret_cast<void**>(&d_buf[b]), cap))) return rc;
            if (int rc = ok(hipEventCreateWithFlags(&up_done[b], hipEventDisableTiming))) return rc;
            if (int rc = ok(hipEventCreateWithFlags(&buf_free[b], hipEventDisableTiming))) return rc;
        }
        if (want_rbsp) if (int rc = ok(hipMalloc(reinterpret_cast<void**>(&d_rbsp), cap))) return rc;
        if (int rc = ok(hipMalloc(reinterpret_cast<void**>(&d_index), (idx_cap ? idx_cap : 1) * sizeof(hbs_nal_entry)))) return rc;
        if (int rc = ok(hipMalloc(reinterpret_cast<void**>(&d_summary), sizeof(hbs_summary)))) return rc;
        if (int rc = ok(hipHostMalloc(reinterpret_cast<void**>(&h_stage), kStageEntries * sizeof(hbs_nal_entry), hipHostMallocDefault))) return rc;
        if (int rc = ok(hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking))) return rc;
        if (int rc = ok(hipStreamCreateWithFlags(&s_cmp, hipStreamNonBlocking))) return rc;
        allocated = true;
        return 0;
    }
    void release()
    {
        if (s_cmp) (void)hipStreamSynchronize(s_cmp);
        if (s_in) (void)hipStreamSynchronize(s_in);
        for (int b = 0; b < 2; ++b) {
            if (d_buf[b]) (void)hipFree(d_buf[b]);
            if (up_done[b]) (void)hipEventDestroy(up_done[b]);
            if (buf_free[b]) (void)hipEventDestroy(buf_free[b]);
        }
        if (d_rbsp) (void)hipFree(d_rbsp);
        if (d_index) (void)hipFree(d_index);
        if (d_summary) (void)hipFree(d_summary);
        if (h_stage) (void)hipHostFree(h_stage);
        if (s_in) (void)hipStreamDestroy(s_in);
        if (s_cmp) (void)hipStreamDestroy(s_cmp);
    }
    int begin(uint64_t)
    {
        if (!allocated) if (int rc = allocate()) return rc;
        fresh[0] = fresh[1] = 0;
        /* a fresh run: both window buffers are free */
        for (int b = 0; b < 2; ++b) if (int rc = ok(hipEventRecord(buf_free[b], s_cmp))) return rc;
        saved_stream = hbs_ctx_get_stream(ctx);
        return hbs_ctx_set_stream(ctx, s_cmp);
    }
    void end()
    {
        if (s_cmp) { (void)hipStreamSynchronize(s_cmp); (void)hbs_ctx_set_stream(ctx, saved_stream); }
        if (s_in) (void)hipStreamSynchronize(s_in);
    }
    int upload(int b, uint64_t dst_off, uint64_t src_lo, uint64_t len)
    {
        fresh[b] = len;
        /* the buffer's previous window must have been scanned and carried out of */
        if (int rc = ok(hipStreamWaitEvent(s_in, buf_free[b], 0))) return rc;
        if (len) if (int rc = ok(hipMemcpyAsync(d_buf[b] + dst_off, h_stream + src_lo, len, hipMemcpyHostToDevice, s_in))) return rc;
        return ok(hipEventRecord(up_done[b], s_in));
    }
    int carry(int from, uint64_t from_off, int to, uint64_t to_off, uint64_t len)
    {
        if (len) if (int rc = ok(hipMemcpyAsync(d_buf[to] + to_off, d_buf[from] + from_off, len, hipMemcpyDeviceToDevice, s_cmp))) return rc;
        return ok(hipEventRecord(buf_free[from], s_cmp));
    }
    int scan(int b, uint64_t off, uint64_t len, hbs_summary* out)
    {
        if (int rc = ok(hipStreamWaitEvent(s_cmp, up_done[b], 0))) return rc;
        int rc = hbs_index_extract(ctx, d_buf[b] + off, len, d_index, idx_cap, h_rbsp ? d_rbsp : nullptr, h_rbsp ? lead + window + 256 : 0, d_summary);
        if (rc) return rc;
        return hbs_read_summary(ctx, d_summary, out);
    }
    int fetch_index(uint64_t first, uint64_t count, hbs_nal_entry* dst)
    {
        if (int rc = ok(hipMemcpyAsync(dst, d_index + first, count * sizeof(hbs_nal_entry), hipMemcpyDeviceToHost, s_cmp))) return rc;
        return ok(hipStreamSynchronize(s_cmp));
    }
    int fetch_rbsp(uint64_t off, uint64_t len, uint64_t dst_off)
    {
        if (!len) return 0;
        return ok(hipMemcpyAsync(h_rbsp + dst_off, d_rbsp + off, len, hipMemcpyDeviceToHost, s_cmp));
    }
};

} // namespace

extern "C" int hbs_index_extract_host(hbs_ctx* ctx, const uint8_t* h_stream, uint64_t stream_bytes, uint64_t window_bytes,
                                      hbs_nal_entry* h_index, uint64_t index_cap,
                                      uint8_t* h_rbsp, uint64_t rbsp_cap, hbs_summary* h_summary)
{
    if (!ctx || !h_summary || (stream_bytes && !h_stream) || (index_cap && !h_index)) return HBS_E_ARG;
    window_bytes &= ~15ull;
    if (window_bytes < 4096) return HBS_E_ARG;
    /* A NAL longer than the window used to end the call with HBS_E_CAPACITY (the reference's 32 MiB reader at least parses such a
     * NAL cut short, hevc_analyze.c:126,190-209).  Since round 5 the window GROWS: the walk has delivered everything up to the
     * end of the last complete NAL (find_nal_unit is stateless: any NAL end is a place to resume, hevc_analyze.c:176), so the
     * call goes on from there with device windows of twice the size -- again and again up to hbs_ctx_set_ingest_window_max
     * (default 1 GiB; a ceiling at or below window_bytes keeps round 4's behaviour).  Later windows keep the larger size. */
    uint64_t ceiling = hbs_ctx_ingest_window_max(ctx);
    if (ceiling == 0) ceiling = 1ull << 30;
    hbs_summary total;
    total.nal_count = total.nal_found = total.rbsp_bytes = 0; total.stream_bytes = stream_bytes;
    total.stop_reason = 0; total.error = 0; total.reserved[0] = total.reserved[1] = total.reserved[2] = 0;
    uint64_t at = 0;                       /* stream offset this run starts at: 0, or the NAL end the previous run reached */
    uint64_t window = window_bytes;
    int rc = 0;
    for (;;) {
        /* entries one window can produce: a window (with what it scans again) of 2 x window, one NAL per 32 bytes */
        const uint64_t per_window = (2 * window) / 32 + 64;
        HipBackend* be = static_cast<HipBackend*>(hbs_ctx_attachment(ctx));
        if (be && (be->window != window || (h_rbsp != nullptr && !be->want_rbsp))) {
            hbs_ctx_attach(ctx, nullptr, nullptr);               /* frees it */
            be = nullptr;
        }
        if (!be) {
            be = new (std::nothrow) HipBackend();
            if (!be) return HBS_E_HIP;
            be->ctx = ctx; be->window = window; be->lead = window; be->idx_cap = per_window;
            be->want_rbsp = h_rbsp != nullptr;
            hbs_ctx_attach(ctx, be, [](void* p) { HipBackend* b = static_cast<HipBackend*>(p); b->release(); delete b; });
        }
        be->h_stream = h_stream + at; be->h_rbsp = h_rbsp ? h_rbsp + total.rbsp_bytes : nullptr;
        hbs_summary s;
        const uint64_t had = total.nal_count < index_cap ? total.nal_count : index_cap;
        rc = hbs::ingest_windowed(*be, stream_bytes - at, window, h_index + had, index_cap - had, h_rbsp != nullptr,
                                  rbsp_cap > total.rbsp_bytes ? rbsp_cap - total.rbsp_bytes : 0, &s);
        be->end();
        if (!be->allocated) hbs_ctx_attach(ctx, nullptr, nullptr);   /* an allocation failed: do not keep the pieces */
        if (rc) {
            /* a hard error in a later run: the runs before it have delivered entries into h_index / h_rbsp -- say how many (round
             * 5's advice: the summary was left unwritten) */
            total.error = rc;
            *h_summary = total;
            return rc;
        }
        /* this run's entries are relative to where it started */
        if (at || total.rbsp_bytes)
            for (uint64_t k = 0; k < s.nal_count && had + k < index_cap; ++k) {
                h_index[had + k].start += at; h_index[had + k].end += at; h_index[had + k].rbsp_off += total.rbsp_bytes;
            }
        total.nal_count += s.nal_count;
        total.rbsp_bytes += s.rbsp_bytes;
        total.stop_reason = s.stop_reason;
        total.nal_found = total.nal_count + (s.nal_found - s.nal_count);
        const bool window_too_small = s.error == HBS_E_CAPACITY && s.reserved[2] == 1;
        if (window_too_small && 2 * window <= ceiling) {
            at += s.reserved[1];
            window *= 2;
            total.reserved[0] = window;                        /* the window the call ended with (0: never grown) */
            continue;
        }
        if (!total.error) total.error = s.error;                /* (the first error of the call stays) */
        if (window_too_small) { total.reserved[1] = at + s.reserved[1]; total.reserved[2] = 1; }     /* the ceiling was reached: where, in the stream */
        break;
    }
    /* (a run that ran out of index or arena room and THEN out of window reported only the latter) */
    if (total.nal_count > index_cap) { total.nal_count = index_cap; if (!total.error) total.error = HBS_E_CAPACITY; }
    if (h_rbsp && total.rbsp_bytes > rbsp_cap && !total.error) total.error = HBS_E_CAPACITY;
    *h_summary = total;
    return 0;
}
