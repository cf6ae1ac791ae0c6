/*
 * hbs_oracle_gen.c -- TEST INFRASTRUCTURE ONLY (see hbs_oracle.h).
 *
 * Host restatement of the synthetic Annex-B stream S(seed, n_nals, mode) that
 * BASELINE.json's configs are quoted on.  The product's
 * device generator (hbs_synth_*) must produce the same bytes; tests compare.
 * There is no reference counterpart: the reference ships no streams.
 *
 *   NAL k (k = 0..n-1):
 *     key_k   = seed ^ ((k+1) * GOLDEN)
 *     L_k     = 8192 + mix(key_k) % 4097             RBSP bytes (8..12 KiB)
 *     word w  = mix((key_k ^ SALT) + (w+1) * GOLDEN)  little-endian 8 bytes
 *     byte j  = word[j/8] >> 8*(j%8)
 *     mode 1 ("zero-heavy"): b < 26 -> 00, 26..38 -> 01 + (b-26)%3
 *     bytes 0,1 = hdr0,hdr1 (02 01 = TRAIL_R), byte L-1 = 80
 *     start code 00 00 00 01 when k % 4 == 0, else 00 00 01
 *     NAL bytes = rbsp_to_nal(RBSP)   (h264_nal.c:92-132 semantics)
 */
#include "hbs_oracle.h"

#define ORC_GOLDEN 0x9E3779B97F4A7C15ull
#define ORC_SALT   0xD1B54A32D192ED03ull

uint64_t orc_mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

uint32_t orc_gen_rbsp_len(uint64_t seed, uint64_t k)
{
    uint64_t key = seed ^ ((k + 1) * ORC_GOLDEN);
    return 8192u + (uint32_t)(orc_mix64(key) % 4097u);
}

void orc_gen_rbsp(uint64_t seed, uint64_t k, int mode, uint8_t* out /* L_k bytes */)
{
    uint64_t key = seed ^ ((k + 1) * ORC_GOLDEN);
    uint32_t len = orc_gen_rbsp_len(seed, k), j;
    for (j = 0; j < len; j++) {
        uint64_t w = orc_mix64((key ^ ORC_SALT) + (uint64_t)(j / 8 + 1) * ORC_GOLDEN);
        unsigned b = (unsigned)(w >> (8 * (j % 8))) & 0xFFu;
        if (mode == 1) {
            if (b < 26) b = 0;
            else if (b < 39) b = 1 + (b - 26) % 3;
        }
        out[j] = (uint8_t)b;
    }
    out[0] = 0x02;
    out[1] = 0x01;
    out[len - 1] = 0x80;
}

/* Writes the stream; fills idx[k] with start/end/rbsp_off/rbsp_len (status 0;
 * the last NAL gets ORC_ST_UNTERMINATED as orc_index_stream would report) and,
 * if arena != NULL, the packed RBSP arena.  Returns stream bytes, or -1 when
 * out_cap is too small. */
int64_t orc_gen_stream(uint64_t seed, int64_t n_nals, int mode, uint8_t* out, int64_t out_cap,
                       orc_nal_entry* idx, uint8_t* arena)
{
    static uint8_t tmp[8192 + 4097];
    int64_t k, o = 0, a = 0;

    for (k = 0; k < n_nals; k++) {
        int rbsp_size = (int)orc_gen_rbsp_len(seed, (uint64_t)k), nal_size = 0;
        int sc = (k % 4 == 0) ? 4 : 3;
        if (o + sc + (int64_t)rbsp_size * 3 / 2 + 2 > out_cap) return -1;
        orc_gen_rbsp(seed, (uint64_t)k, mode, tmp);
        if (sc == 4) out[o++] = 0;
        out[o++] = 0; out[o++] = 0; out[o++] = 1;
        orc_rbsp_to_nal(tmp, &rbsp_size, out + o, &nal_size);
        if (idx) {
            idx[k].start = (uint64_t)o;
            idx[k].end = (uint64_t)(o + nal_size);
            idx[k].rbsp_off = (uint64_t)a;
            idx[k].rbsp_len = (uint32_t)rbsp_size;
            idx[k].status = (k == n_nals - 1) ? ORC_ST_UNTERMINATED : 0;
        }
        if (arena) {
            int j;
            for (j = 0; j < rbsp_size; j++) arena[a + j] = tmp[j];
        }
        a += rbsp_size;
        o += nal_size;
    }
    return o;
}
