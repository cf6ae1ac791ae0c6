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
#include <string.h>

int find_nal_unit(uint8_t* buf, int size, int* nal_start, int* nal_end);                       /* h264_nal.c:38 */
int nal_to_rbsp(const uint8_t* nal_buf, int* nal_size, uint8_t* rbsp_buf, int* rbsp_size);    /* h264_nal.c:147 */
int rbsp_to_nal(const uint8_t* rbsp_buf, const int* rbsp_size, uint8_t* nal_buf, int* nal_size);  /* h264_nal.c:92 */

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

/* The same walk with everything it learns per NAL written down (tests/test_gpu_fullsize.py compares a multi-GiB
 * GPU index with it entry by entry): start / end in the buffer, where the NAL's RBSP went in the arena, how long it
 * is, nal_to_rbsp's return value, find_nal_unit's return value. */
typedef struct { uint64_t start, end, rbsp_off; int32_t rbsp_len, rc_rbsp, rc_find, pad; } ref_entry;

int64_t ref_walk_index(uint8_t* buf, int64_t size, uint8_t* arena, int64_t arena_cap, ref_entry* ent, int64_t cap, int64_t* rbsp_bytes)
{
    int64_t base = 0, n = 0, out = 0;
    for (;;) {
        const int64_t left = size - base;
        const int win = left > 0x7fff0000 ? 0x7fff0000 : (int)left;
        int s = 0, e = 0, r;
        if (win <= 0) break;
        r = find_nal_unit(buf + base, win, &s, &e);
        if (r == 0) break;
        {
            int nal_size = e - s;
            int rbsp_size = (arena_cap - out) > 0x7fff0000 ? 0x7fff0000 : (int)(arena_cap - out);
            const int rc = nal_to_rbsp(buf + base + s, &nal_size, arena + out, &rbsp_size);
            if (n < cap) {
                ent[n].start = (uint64_t)(base + s); ent[n].end = (uint64_t)(base + e); ent[n].rbsp_off = (uint64_t)out;
                ent[n].rbsp_len = rc >= 0 ? rbsp_size : -1; ent[n].rc_rbsp = rc; ent[n].rc_find = r; ent[n].pad = 0;
            }
            if (rc >= 0) out += rbsp_size;
            n++;
        }
        if (r < 0) break;
        base += e;
    }
    *rbsp_bytes = out;
    return n;
}

/* The way back with the reference's rbsp_to_nal: NAL k = start code (00 00 00 01 when k % 4 == 0, else 00 00 01: the
 * rule of the synthetic stream) + rbsp_to_nal of its RBSP.  Returns the bytes written, -1 when out is too small. */
int64_t ref_emit_synthetic(const uint8_t* arena, const uint64_t* rbsp_off, const int32_t* rbsp_len, int64_t n, uint8_t* out, int64_t out_cap)
{
    int64_t k, o = 0;
    for (k = 0; k < n; k++) {
        int rbsp_size = rbsp_len[k], nal_size = 0;
        const int sc = (k % 4 == 0) ? 4 : 3;
        if (o + sc + (int64_t)rbsp_size * 3 / 2 + 4 > out_cap) return -1;
        memset(out + o, 0, (size_t)(sc - 1));
        out[o + sc - 1] = 1;
        o += sc;
        nal_size = (int)(out_cap - o > 0x7fff0000 ? 0x7fff0000 : out_cap - o);
        rbsp_to_nal(arena + rbsp_off[k], &rbsp_size, out + o, &nal_size);
        o += nal_size;
    }
    return o;
}
