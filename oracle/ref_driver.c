/*
 * ref_driver.c -- TEST / BASELINE INFRASTRUCTURE, not product code.
 *
 * The loop a caller of the reference runs over a buffer (hevc_analyze.c:135-177: find_nal_unit,
 * nal_to_rbsp per NAL, resume at the NAL's end), written against the REAL reference library
 * (oracle/_ref/libhevcref.so, compiled from /root/reference by oracle/Makefile).  bench.py times
 * it as cpu_baseline.kind = "reference" when the prebuilt files are present.  Only the two
 * prototypes below are declared here; nothing of the reference's source is copied.
 */
#include <stdint.h>

int find_nal_unit(uint8_t* buf, int size, int* nal_start, int* nal_end);                       /* h264_nal.c:38 */
int nal_to_rbsp(const uint8_t* nal_buf, int* nal_size, uint8_t* rbsp_buf, int* rbsp_size);    /* h264_nal.c:147 */

/* walks buf[0, size): returns the number of NALs found; *rbsp_bytes = bytes nal_to_rbsp produced, packed
 * back to back into arena; starts[k] (optional, cap entries) = offset of NAL k */
int64_t ref_walk(uint8_t* buf, int64_t size, uint8_t* arena, int64_t arena_cap, int64_t* rbsp_bytes,
                 uint64_t* starts, int64_t cap)
{
    int64_t base = 0, n = 0, out = 0;
    for (;;) {
        const int64_t left = size - base;
        const int win = left > 0x7fff0000 ? 0x7fff0000 : (int)left;      /* the reference takes an int */
        int s = 0, e = 0, r;
        if (win <= 0) break;
        r = find_nal_unit(buf + base, win, &s, &e);
        if (r == 0) break;
        {
            int nal_size = e - s;
            int rbsp_size = (arena_cap - out) > 0x7fff0000 ? 0x7fff0000 : (int)(arena_cap - out);
            const int rc = nal_to_rbsp(buf + base + s, &nal_size, arena + out, &rbsp_size);
            if (starts && n < cap) starts[n] = (uint64_t)(base + s);
            if (rc >= 0) out += rbsp_size;
            n++;
        }
        if (r < 0) break;                                                   /* the last NAL runs to the end of the buffer */
        base += e;
    }
    *rbsp_bytes = out;
    return n;
}
