/*
 * hbs_oracle_nal.c -- TEST INFRASTRUCTURE ONLY (see hbs_oracle.h).
 *
 * Restatement of the reference's Annex-B byte layer.  Loop shapes follow the
 * reference byte-at-a-time code so that timing this file is a fair stand-in
 * for the reference CPU path (bench.py cpu_baseline, kind "port").
 */
#include "hbs_oracle.h"
#include <string.h>

/* byte i of a buffer of `size` bytes; past-the-end reads are 0xFF (header). */
static inline unsigned peek(const uint8_t* b, int64_t size, int64_t i)
{
    return i < size ? b[i] : 0xFFu;
}

/* 00 00 01 at i? */
static inline int is_sc3(const uint8_t* b, int64_t n, int64_t i)
{
    return peek(b, n, i) == 0 && peek(b, n, i + 1) == 0 && peek(b, n, i + 2) == 1;
}
/* 00 00 00 at i? */
static inline int is_z3(const uint8_t* b, int64_t n, int64_t i)
{
    return peek(b, n, i) == 0 && peek(b, n, i + 1) == 0 && peek(b, n, i + 2) == 0;
}
/* 00 00 00 01 at i? */
static inline int is_sc4(const uint8_t* b, int64_t n, int64_t i)
{
    return is_z3(b, n, i) && peek(b, n, i + 3) == 1;
}

/*
 * reference: h264_nal.c:38-76 (find_nal_unit).
 *   :42-43  outputs zeroed
 *   :46-53  start search; first candidate unchecked, then `i+4 >= size` -> 0
 *   :55-58  skip the leading zero of a 4-byte code
 *   :60-62  nal_start = i + 3
 *   :64-72  end search; first candidate unchecked, then `i+3 >= size` ->
 *           nal_end = size, return -1
 *   :74-75  nal_end = i, return length
 */
int64_t orc_find_nal_unit64(const uint8_t* buf, int64_t size, int64_t* nal_start, int64_t* nal_end)
{
    int64_t i = 0;
    *nal_start = 0;
    *nal_end = 0;

    while (!is_sc3(buf, size, i) && !is_sc4(buf, size, i)) {
        i++;
        if (i + 4 >= size) return 0;
    }
    if (!is_sc3(buf, size, i)) i++;
    if (!is_sc3(buf, size, i)) return 0;   /* :60, unreachable */
    i += 3;
    *nal_start = i;

    while (!is_z3(buf, size, i) && !is_sc3(buf, size, i)) {
        i++;
        if (i + 3 >= size) { *nal_end = size; return -1; }
    }
    *nal_end = i;
    return *nal_end - *nal_start;
}

int orc_find_nal_unit(const uint8_t* buf, int size, int* nal_start, int* nal_end)
{
    int64_t s, e;
    int64_t r = orc_find_nal_unit64(buf, size, &s, &e);
    *nal_start = (int)s;
    *nal_end = (int)e;
    return (int)r;
}

/*
 * reference: h264_nal.c:147-200 (nal_to_rbsp).
 *   :156-159 two zeros then a byte < 3            -> -1
 *   :161-177 two zeros then 03: next byte > 3 (if any) -> -1; 03 is the last
 *            byte -> stop (dropped); else skip it and restart the zero count
 *   :179-183 output full -> -1
 *   :185-194 copy, count zeros
 *   :197-199 *nal_size = consumed, *rbsp_size = produced (success only)
 */
int orc_nal_to_rbsp(const uint8_t* nal_buf, int* nal_size, uint8_t* rbsp_buf, int* rbsp_size)
{
    int in = 0, out = 0, zeros = 0;
    const int n = *nal_size;

    for (in = 0; in < n; in++) {
        if (zeros == 2 && nal_buf[in] < 3) return -1;
        if (zeros == 2 && nal_buf[in] == 3) {
            if (in < n - 1 && nal_buf[in + 1] > 3) return -1;
            if (in == n - 1) break;
            in++;
            zeros = 0;
        }
        if (out >= *rbsp_size) return -1;
        rbsp_buf[out++] = nal_buf[in];
        zeros = (nal_buf[in] == 0) ? zeros + 1 : 0;
    }
    *nal_size = in;
    *rbsp_size = out;
    return out;
}

/*
 * reference: h264_nal.c:92-132 (rbsp_to_nal).
 *   :110-116 two zeros pending and next byte <= 3 -> emit 03, restart count,
 *            re-examine the same input byte
 *   :117-127 copy, count zeros
 *   :130-131 *nal_size = produced (input value ignored; no bounds check)
 */
int orc_rbsp_to_nal(const uint8_t* rbsp_buf, const int* rbsp_size, uint8_t* nal_buf, int* nal_size)
{
    int in = 0, out = 0, zeros = 0;
    const int n = *rbsp_size;

    while (in < n) {
        if (zeros == 2 && (rbsp_buf[in] & 0xFC) == 0) {
            nal_buf[out++] = 3;
            zeros = 0;
            continue;
        }
        nal_buf[out++] = rbsp_buf[in];
        zeros = (rbsp_buf[in] == 0) ? zeros + 1 : 0;
        in++;
    }
    *nal_size = out;
    return out;
}

/*
 * reference: hevc_analyze.c:135-205 -- the NAL loop over one window, here
 * with the whole stream as the window and 64-bit offsets:
 *   :135      while (find_nal_unit(p, sz, ...) > 0)
 *   :148,175  p += nal_end (start of the next search is the previous end)
 *   :176      sz -= nal_end
 *   :190-205  after the loop the "last NAL" [nal_start, nal_end) of the failed
 *             call is parsed too; it is a real NAL only on the -1 path.
 */
int64_t orc_index_stream(const uint8_t* buf, int64_t size, orc_nal_entry* out, int64_t cap, int* stop_reason)
{
    int64_t base = 0, n = 0, s, e, r;

    *stop_reason = 0;
    for (;;) {
        r = orc_find_nal_unit64(buf + base, size - base, &s, &e);
        if (r <= 0) break;
        if (n < cap) {
            out[n].start = (uint64_t)(base + s);
            out[n].end = (uint64_t)(base + e);
            out[n].rbsp_off = 0; out[n].rbsp_len = 0; out[n].status = 0;
        }
        n++;
        base += e;
    }
    if (r == -1) {
        if (n < cap) {
            out[n].start = (uint64_t)(base + s);
            out[n].end = (uint64_t)(base + e);
            out[n].rbsp_off = 0; out[n].rbsp_len = 0; out[n].status = ORC_ST_UNTERMINATED;
        }
        n++;
        *stop_reason = -1;
    } else if (s != 0 || e != 0) {
        *stop_reason = 1;   /* start code found, zero-length NAL: the loop stops here */
    }
    return n;
}

/* number of 00 00 03 windows fully inside n[0..len) -- App. B of SURVEY.md:
 * for every NAL nal_to_rbsp accepts this is exactly the bytes it removes. */
static int64_t count_epb(const uint8_t* n, int64_t len)
{
    int64_t i, c = 0;
    for (i = 2; i < len; i++)
        if (n[i] == 3 && n[i - 1] == 0 && n[i - 2] == 0) c++;
    return c;
}

int64_t orc_extract_rbsp(const uint8_t* buf, orc_nal_entry* idx, int64_t n, uint8_t* arena, int64_t arena_cap)
{
    int64_t k, off = 0;

    for (k = 0; k < n; k++) {
        const uint8_t* nal = buf + idx[k].start;
        int64_t len = (int64_t)(idx[k].end - idx[k].start);
        int64_t keep = len - count_epb(nal, len);
        int nal_size = (int)len, rbsp_size = (int)len, rc;

        idx[k].rbsp_off = (uint64_t)off;
        idx[k].rbsp_len = (uint32_t)keep;
        idx[k].status &= ORC_ST_UNTERMINATED;
        if (off + keep > arena_cap) return -1;

        rc = orc_nal_to_rbsp(nal, &nal_size, arena + off, &rbsp_size);
        if (rc < 0) {
            /* reference leaves the output unspecified; this build defines it
             * as "every byte except 00 00 03 emulation bytes" so that arenas
             * can be compared whole. */
            int64_t i, j = 0;
            for (i = 0; i < len; i++) {
                if (i >= 2 && nal[i] == 3 && nal[i - 1] == 0 && nal[i - 2] == 0) continue;
                arena[off + j++] = nal[i];
            }
            idx[k].status |= ORC_ST_ERROR;
        } else {
            if (nal_size == len - 1) idx[k].status |= ORC_ST_TRAILING03;
        }
        off += keep;
    }
    return off;
}

int64_t orc_emit_annexb(const uint8_t* arena, const orc_nal_entry* idx, int64_t n, uint8_t* out, int64_t out_cap)
{
    int64_t k, o = 0, prev_end = 0;

    for (k = 0; k < n; k++) {
        int64_t gap = (int64_t)idx[k].start - prev_end;   /* zeros then 01 */
        int rbsp_size = (int)idx[k].rbsp_len, nal_size = 0;
        if (gap < 3) return -1;
        if (o + gap + (int64_t)rbsp_size * 3 / 2 + 4 > out_cap) return -2;
        memset(out + o, 0, (size_t)(gap - 1));
        out[o + gap - 1] = 1;
        o += gap;
        orc_rbsp_to_nal(arena + idx[k].rbsp_off, &rbsp_size, out + o, &nal_size);
        o += nal_size;
        prev_end = (int64_t)idx[k].end;
    }
    return o;
}
