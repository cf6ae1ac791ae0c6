// Dev aid, round 4: where K12's per-process spread (5.88 ... 6.23 ms on one box) comes from.  No torch: the stream is generated
// through the C ABI, and the buffers of hbs_index_extract are placed in the ways under test.  Results: profiles/r04/placement_*.txt.
//
//   placement MODE [NALS] [REPS]
//     sep        stream, arena, index: three hipMalloc calls (what torch's allocator amounts to for buffers this large)
//     slab       ONE hipMalloc; stream, arena and index are 2 MiB-aligned pieces of it
//     skew       slab, then the arena moved up by 0, 256 B, 4 KiB, 64 KiB ... inside it (time against the skew)
//     realloc    sep, with the arena + index freed and allocated again 8 times (round 3's placement_probe, without torch)
//     vmm        stream / arena / index in one virtual range backed by hipMemCreate chunks of the granularity asked for (env VMM_CHUNK_MB)
//     idxskew    slab; the INDEX at 12 distances from the arena
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "hevcbitstream_amd.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define HB(x) do { int r_ = (x); if (r_) { fprintf(stderr, "%s:%d %s -> %d (%s)\n", __FILE__, __LINE__, #x, r_, hbs_last_error(ctx)); exit(3); } } while (0)

static hbs_ctx* ctx;
static uint64_t up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

struct Stream { uint8_t* bytes; uint64_t n, nals, rbsp_bytes; };

/* S(seed, nals, uniform) written to `dst` (cap bytes); the generator's own arena and index are freed again */
static Stream make_stream(uint8_t* dst, uint64_t cap, uint64_t nals)
{
    const uint64_t rcap = hbs_synth_rbsp_bound(nals);
    uint8_t* rbsp; hbs_nal_entry* idx; hbs_summary* dsum;
    CK(hipMalloc((void**)&rbsp, rcap)); CK(hipMalloc((void**)&idx, nals * 32 + 64)); CK(hipMalloc((void**)&dsum, sizeof(hbs_summary)));
    hbs_summary s;
    HB(hbs_synth_rbsp(ctx, 0x1234, nals, 0, rbsp, rcap, idx, dsum));
    HB(hbs_read_summary(ctx, dsum, &s));
    const uint64_t rb = s.stream_bytes;
    if (hbs_annexb_bound(rb, nals) > cap) { fprintf(stderr, "stream buffer too small\n"); exit(4); }
    HB(hbs_emit_annexb(ctx, rbsp, rb, idx, nals, 1, dst, cap, idx, dsum));
    HB(hbs_read_summary(ctx, dsum, &s));
    if (s.error) { fprintf(stderr, "emit error %d\n", s.error); exit(4); }
    CK(hipFree(rbsp)); CK(hipFree(idx)); CK(hipFree(dsum));
    return Stream{dst, s.stream_bytes, nals, rb};
}

static double run(const Stream& st, uint8_t* arena, uint64_t arena_cap, hbs_nal_entry* index, hbs_summary* dsum, int reps, double* lo = nullptr, double* hi = nullptr)
{
    std::vector<float> ms;
    for (int i = 0; i < reps + 1; ++i) {
        HB(hbs_index_extract(ctx, st.bytes, st.n, index, st.nals + 8, arena, arena_cap, dsum));
        float t = 0; HB(hbs_ctx_kernel_ms(ctx, &t));
        if (i) ms.push_back(t);
    }
    hbs_summary s; HB(hbs_read_summary(ctx, dsum, &s));
    if (s.error || s.nal_count != st.nals || s.rbsp_bytes != st.rbsp_bytes) { fprintf(stderr, "WRONG RESULT: error %d nals %llu rbsp %llu\n", s.error, (unsigned long long)s.nal_count, (unsigned long long)s.rbsp_bytes); exit(5); }
    std::sort(ms.begin(), ms.end());
    if (lo) *lo = ms.front();
    if (hi) *hi = ms.back();
    return ms[ms.size() / 2];
}

int main(int argc, char** argv)
{
    const char* mode = argc > 1 ? argv[1] : "sep";
    const uint64_t nals = argc > 2 ? strtoull(argv[2], nullptr, 10) : 1677000ull;
    const int reps = argc > 3 ? atoi(argv[3]) : 5;
    if (hbs_ctx_create(&ctx, 0)) { fprintf(stderr, "no GPU\n"); return 1; }
    HB(hbs_ctx_enable_timing(ctx, 1));
    const uint64_t M2 = 2ull << 20;
    const uint64_t scap = up(hbs_annexb_bound(hbs_synth_rbsp_bound(nals), nals) + 4096, M2);
    const uint64_t acap = scap, icap = up((nals + 8) * 32, M2);
    hbs_summary* dsum; CK(hipMalloc((void**)&dsum, 256));
    double lo, hi;
    if (!strcmp(mode, "sep") || !strcmp(mode, "realloc")) {
        uint8_t *sb, *ar; hbs_nal_entry* ix;
        CK(hipMalloc((void**)&sb, scap));
        Stream st = make_stream(sb, scap, nals);
        const int rounds = !strcmp(mode, "realloc") ? 8 : 1;
        for (int r = 0; r < rounds; ++r) {
            CK(hipMalloc((void**)&ar, acap)); CK(hipMalloc((void**)&ix, icap));
            const double m = run(st, ar, acap, ix, dsum, reps, &lo, &hi);
            printf("%s stream %p arena %p index %p : %.3f ms (%.3f .. %.3f)  frac %.4f\n", mode, sb, ar, ix, m, lo, hi, (st.n + st.rbsp_bytes + 32.0 * nals) / (m * 1e-3) / 8e12);
            CK(hipFree(ar)); CK(hipFree(ix));
            if (r + 1 < rounds) { void* junk; CK(hipMalloc(&junk, (size_t)(r + 1) * (37ull << 20))); CK(hipFree(junk)); }
        }
    } else if (!strcmp(mode, "slab") || !strcmp(mode, "skew") || !strcmp(mode, "idxskew")) {
        uint8_t* slab; const uint64_t extra = 1ull << 30;
        CK(hipMalloc((void**)&slab, scap + acap + icap + extra));
        Stream st = make_stream(slab, scap, nals);
        if (!strcmp(mode, "slab")) {
            const double m = run(st, slab + scap, acap, (hbs_nal_entry*)(slab + scap + acap), dsum, reps, &lo, &hi);
            printf("slab %p (+%llu MiB arena, +%llu MiB index): %.3f ms (%.3f .. %.3f)  frac %.4f\n", slab, (unsigned long long)(scap >> 20), (unsigned long long)((scap + acap) >> 20), m, lo, hi,
                   (st.n + st.rbsp_bytes + 32.0 * nals) / (m * 1e-3) / 8e12);
        } else if (!strcmp(mode, "skew")) {
            const uint64_t skews[] = {0, 256, 1024, 4096, 16384, 65536, 262144, 1ull << 20, 3ull << 20, 16ull << 20, 100ull << 20, 512ull << 20, (512ull << 20) + 4096 + 256};
            for (uint64_t sk : skews) {
                const double m = run(st, slab + scap + sk, acap, (hbs_nal_entry*)(slab + scap + acap + extra - icap), dsum, reps, &lo, &hi);
                printf("skew %10llu : %.3f ms (%.3f .. %.3f)\n", (unsigned long long)sk, m, lo, hi);
            }
        } else {
            for (int k = 0; k < 12; ++k) {
                const uint64_t off = scap + acap + (uint64_t)k * (icap + 4096 * (uint64_t)k);
                if (off + icap > scap + acap + icap + extra) break;
                const double m = run(st, slab + scap, acap, (hbs_nal_entry*)(slab + off), dsum, reps, &lo, &hi);
                printf("idxskew +%llu KiB: %.3f ms (%.3f .. %.3f)\n", (unsigned long long)((off - scap - acap) >> 10), m, lo, hi);
            }
        }
    } else if (!strcmp(mode, "vmm")) {
        const char* e = getenv("VMM_CHUNK_MB");
        hipMemAllocationProp prop; memset(&prop, 0, sizeof(prop));
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        size_t gran_min = 0, gran_rec = 0;
        CK(hipMemGetAllocationGranularity(&gran_min, &prop, hipMemAllocationGranularityMinimum));
        CK(hipMemGetAllocationGranularity(&gran_rec, &prop, hipMemAllocationGranularityRecommended));
        uint64_t chunk = e ? (uint64_t)atoll(e) << 20 : 1ull << 30;
        chunk = up(chunk, gran_rec);
        const uint64_t total = up(scap + acap + icap, chunk);
        void* va = nullptr;
        CK(hipMemAddressReserve(&va, total, chunk > (1ull << 30) ? (1ull << 30) : chunk, nullptr, 0));
        std::vector<hipMemGenericAllocationHandle_t> hs;
        for (uint64_t o = 0; o < total; o += chunk) {
            hipMemGenericAllocationHandle_t h;
            CK(hipMemCreate(&h, chunk, &prop, 0));
            CK(hipMemMap((uint8_t*)va + o, chunk, 0, h, 0));
            hs.push_back(h);
        }
        hipMemAccessDesc acc; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(va, total, &acc, 1));
        uint8_t* slab = (uint8_t*)va;
        Stream st = make_stream(slab, scap, nals);
        const double m = run(st, slab + scap, acap, (hbs_nal_entry*)(slab + scap + acap), dsum, reps, &lo, &hi);
        printf("vmm va %p granularity min %zu rec %zu chunk %llu MiB x %zu: %.3f ms (%.3f .. %.3f)  frac %.4f\n", va, gran_min, gran_rec, (unsigned long long)(chunk >> 20), hs.size(), m, lo, hi,
               (st.n + st.rbsp_bytes + 32.0 * nals) / (m * 1e-3) / 8e12);
    } else { fprintf(stderr, "unknown mode %s\n", mode); return 1; }
    hbs_ctx_destroy(ctx);
    return 0;
}
