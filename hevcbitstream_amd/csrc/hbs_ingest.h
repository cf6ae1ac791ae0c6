/*
 * hbs_ingest.h -- windowed ingest: scan + index + RBSP extraction of a stream that
 * lives in HOST memory and may be larger than device memory, with the same result as
 * hbs_index_extract over the whole stream.
 *
 * Replaces the windowed reader of the reference's CLI (hevc_analyze.c:124-210: 32 MiB
 * reads, find_nal_unit loop, carry-over of the unfinished tail), whose walk it
 * reproduces exactly: find_nal_unit is stateless, every call resumes at the end of the
 * previous NAL (hevc_analyze.c:176), so a window may start at ANY previous NAL end and
 * finds the same NALs from there on.  Per window:
 *
 *   - scan stream[a, hi) where a = the end of the last complete NAL rounded down to 16
 *     (device loads are 16-byte aligned; the bytes in front of that end are the tail of
 *     a finished NAL and cannot hold a pattern), hi = a multiple of 16 or the stream end;
 *   - unless hi is the stream end, a last NAL that the window leaves unterminated is not
 *     trusted (its end, or its very existence, may be an artefact of the end-of-buffer
 *     rules h264_nal.c:52,71): it is dropped and scanned again with the next window,
 *     which therefore starts at the end of the last NAL that was kept (the reference's own
 *     carry-over mishandles a NAL that straddles two reads);
 *   - NALs found again because of the round-down are skipped (starts are monotonic).
 *
 * The driver below is a template over a backend so that the same logic runs against the
 * HIP backend (hbs_ingest.hip: two streams, the upload of the next window overlaps the
 * scan and the download of the current one) and against the CPU single-stepper in
 * tests/sim.  Host code; plain C++.
 */
#ifndef HBS_INGEST_H
#define HBS_INGEST_H

#include <stdint.h>
#include "../../include/hevcbitstream_amd.h"

namespace hbs {

/*
 * Backend concept:
 *   uint64_t lead_capacity()                      bytes a window buffer holds in front of the fresh upload
 *   uint64_t fresh_len(buf)                       length of the last upload into `buf`
 *   int begin(n)                                  once
 *   int upload(buf, dst_off, src_lo, len)         stream[src_lo, +len) -> window buffer `buf` at dst_off (may be asynchronous)
 *   int carry(from_buf, from_off, to_buf, to_off, len)   window-buffer to window-buffer
 *   int scan(buf, off, len, hbs_summary* out)     run K12 on buffer bytes [off, off+len); synchronous result
 *   int fetch_index(first, count, hbs_nal_entry* dst)    entries [first, first+count) of the last scan (window-relative)
 *   hbs_nal_entry* staging(); uint64_t staging_entries() where fetch_index may put them (page-locked on the GPU)
 *   int fetch_rbsp(off, len, uint64_t dst_off)           RBSP bytes of the last scan -> caller's arena at dst_off
 *   uint64_t index_capacity()                     entries one scan can return
 */
template <class Backend>
int ingest_windowed(Backend& be, uint64_t n, uint64_t window_bytes,
                    hbs_nal_entry* h_index, uint64_t index_cap, bool want_rbsp, uint64_t rbsp_cap, hbs_summary* out)
{
    hbs_summary res;
    res.nal_count = res.nal_found = res.rbsp_bytes = 0;
    res.stream_bytes = n;
    res.stop_reason = 0; res.error = 0;
    res.reserved[0] = res.reserved[1] = res.reserved[2] = 0;
    window_bytes &= ~15ull;
    if (window_bytes < 64) return HBS_E_ARG;
    const uint64_t lead = be.lead_capacity();     /* room in front of the fresh bytes for what is scanned again */
    int rc = be.begin(n);
    if (rc) return rc;

    hbs_nal_entry* const stage = be.staging();         /* entries on their way to the caller's index */
    uint64_t lo = 0;                   /* the walk resumes here (a NAL end, or 0) */
    uint64_t hi = 0;                   /* stream bytes uploaded so far             */
    int buf = 0;
    bool have_last = false;
    uint64_t last_start = 0;           /* start of the last entry written          */
    /* window 0: fresh bytes sit at offset `lead` of their buffer */
    {
        const uint64_t len = n < window_bytes ? n : window_bytes;
        rc = be.upload(buf, lead, 0, len);
        if (rc) return rc;
        hi = len;
    }
    for (;;) {
        const bool final_window = hi == n;
        const uint64_t a = lo & ~15ull;
        /* the window occupies [lead - (fresh_lo - a), lead + fresh_len) of buffer `buf`, where
         * fresh_lo = the first byte uploaded for it; by construction that prefix is in place */
        const uint64_t win_len = hi - a;
        const uint64_t fresh_lo = hi - be.fresh_len(buf);
        const uint64_t win_off = lead - (fresh_lo - a);
        /* start the next window's upload now: it overlaps this window's scan */
        uint64_t next_len = 0;
        if (!final_window) {
            next_len = (n - hi) < window_bytes ? (n - hi) : window_bytes;
            rc = be.upload(buf ^ 1, lead, hi, next_len);
            if (rc) return rc;
        }
        hbs_summary s;
        rc = be.scan(buf, win_off, win_len, &s);
        if (rc) return rc;
        if (s.error == HBS_E_TIMEOUT) { res.error = s.error; break; }
        if (s.error == HBS_E_CAPACITY && s.nal_count >= be.index_capacity()) {
            /* more NALs in one window than the device index holds: the caller must use smaller windows */
            res.error = HBS_E_CAPACITY;
            break;
        }
        uint64_t count = s.nal_count;
        bool stop = final_window;
        uint64_t keep = count;
        if (!final_window) {
            if (s.stop_reason == 1) {
                stop = true;                                  /* an empty NAL ends the reference's walk for good */
            } else if (s.stop_reason == -1 && count > 0) {
                keep = count - 1;                             /* the unterminated last NAL is scanned again */
            }
        }
        /* entries: skip the ones found again, rebase, append */
        uint64_t new_lo = lo;
        uint64_t first_kept = keep, kept = 0, rbsp_first = 0, rbsp_end = 0;
        const uint64_t chunk = be.staging_entries();
        for (uint64_t i = 0; i < keep; i += chunk) {
            const uint64_t m = (keep - i) < chunk ? (keep - i) : chunk;
            rc = be.fetch_index(i, m, stage);
            if (rc) return rc;
            for (uint64_t j = 0; j < m; ++j) {
                hbs_nal_entry e = stage[j];
                const uint64_t gstart = a + e.start, gend = a + e.end;
                if (have_last && gstart <= last_start) continue;            /* found again after the round-down */
                if (first_kept == keep) { first_kept = i + j; rbsp_first = e.rbsp_off; }
                rbsp_end = e.rbsp_off + e.rbsp_len;
                if (res.nal_count < index_cap) {
                    e.start = gstart; e.end = gend;
                    e.rbsp_off = res.rbsp_bytes + (e.rbsp_off - rbsp_first);
                    h_index[res.nal_count] = e;
                } else {
                    res.error = HBS_E_CAPACITY;
                }
                ++res.nal_count; ++kept;
                have_last = true; last_start = gstart;
                new_lo = gend;
            }
        }
        if (kept && want_rbsp) {
            const uint64_t bytes = rbsp_end - rbsp_first;
            if (res.rbsp_bytes + bytes <= rbsp_cap) {
                rc = be.fetch_rbsp(rbsp_first, bytes, res.rbsp_bytes);
                if (rc) return rc;
            } else {
                res.error = HBS_E_CAPACITY;
            }
        }
        if (kept) res.rbsp_bytes += rbsp_end - rbsp_first;
        if (stop) {
            res.stop_reason = s.stop_reason;
            res.nal_found = res.nal_count + (s.nal_found - s.nal_count);
            break;
        }
        /* where the next window resumes */
        if (kept == 0 && count == 0) {
            /* no start code at all in this window: only its last 4 bytes can still begin one (h264_nal.c:52) */
            const uint64_t tail = hi >= 4 ? hi - 4 : 0;
            new_lo = tail > lo ? tail : lo;
        }
        lo = new_lo;
        const uint64_t na = lo & ~15ull;
        const uint64_t again = hi - na;                       /* bytes of this window that the next one scans again */
        if (again > lead) {
            /* a NAL (or a gap) longer than the window: everything up to `lo` is delivered; the caller may go on from there with
             * a larger window (reserved[2] = 1, reserved[1] = lo: hbs_index_extract_host does, up to its ceiling) */
            res.error = HBS_E_CAPACITY; res.reserved[1] = lo; res.reserved[2] = 1;
            break;
        }
        rc = be.carry(buf, win_off + (na - a), buf ^ 1, lead - again, again);
        if (rc) return rc;
        hi += next_len;
        buf ^= 1;
    }
    if (res.nal_count > index_cap) res.nal_count = index_cap;
    *out = res;
    return 0;
}

} // namespace hbs
#endif
