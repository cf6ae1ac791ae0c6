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

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__global__ void k_copy(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}
__global__ void k_read(const u32x4* __restrict__ src, size_t n, uint32_t* sink)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { u32x4 v = __builtin_nontemporal_load(src + i); if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345u) *sink = 1; }
}
__global__ void k_write(u32x4* __restrict__ dst, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { u32x4 v = {(uint32_t)i, 1u, 2u, 3u}; __builtin_nontemporal_store(v, dst + i); }
}
/* K12's geometry without its logic: persistent workgroups of 4 wavefronts, a wavefront holds ROWS rows of 1 KiB in registers, all
 * loaded (DEPTH at a time in flight), then all stored at dst + shift */
template <int ROWS, int DEPTH>
__global__ __launch_bounds__(256, 2) void k_tilecopy(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint64_t ntiles, int shift, unsigned* ticket)
{
    __shared__ unsigned tk;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    typedef u32x4 u32x4_u1 __attribute__((aligned(1)));
    for (;;) {
        if (threadIdx.x == 0) tk = atomicAdd(ticket, 1u);
        __syncthreads();
        const uint64_t t = tk;
        __syncthreads();
        if (t >= ntiles) break;
        const uint64_t off = (t * 4 + wv) * (uint64_t)(ROWS * 1024) + 16 * lane;
        u32x4 r[ROWS];
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            r[i] = __builtin_nontemporal_load((const u32x4*)(src + off + 1024 * i));
            if (DEPTH > 0 && i >= DEPTH) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DEPTH > 0 ? DEPTH : 0) : "memory");
        }
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            __builtin_nontemporal_store(r[i], (u32x4_u1*)(dst + off + 1024 * i + shift));
            if (DEPTH > 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DEPTH > 0 ? DEPTH : 0) : "memory");
        }
    }
}
/* fine-grained mix: every thread alternates a load from src and a store to dst, many independent pairs in flight */
__global__ void k_copy_mis(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t n, int shift)
{
    typedef u32x4 u32x4_u1 __attribute__((aligned(1)));
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(__builtin_nontemporal_load((const u32x4*)src + i), (u32x4_u1*)(dst + 16 * i + shift));
}
template <class F> static double time_ms(F f, int reps = 5)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    std::vector<float> ms;
    for (int i = 0; i < reps + 1; ++i) { (void)hipEventRecord(a, 0); f(); (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b); float t; (void)hipEventElapsedTime(&t, a, b); if (i) ms.push_back(t); }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

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
    } else if (!strcmp(mode, "probe")) {
        /* one process = one placement: K12, a plain copy / read / write of the same buffers, and K12 on 1 GiB pieces */
        uint8_t *sb, *ar; hbs_nal_entry* ix; uint32_t* sink;
        CK(hipMalloc((void**)&sb, scap));
        Stream st = make_stream(sb, scap, nals);
        CK(hipMalloc((void**)&ar, acap)); CK(hipMalloc((void**)&ix, icap)); CK(hipMalloc((void**)&sink, 64));
        const double m = run(st, ar, acap, ix, dsum, reps, &lo, &hi);
        const size_t n16 = st.n / 16;
        const unsigned blocks = (unsigned)((n16 + 255) / 256);
        CK(hipDeviceSynchronize());
        const double tc = time_ms([&] { k_copy<<<blocks, 256>>>((const u32x4*)sb, (u32x4*)ar, n16); });
        const double tr = time_ms([&] { k_read<<<blocks, 256>>>((const u32x4*)sb, n16, sink); });
        const double tw = time_ms([&] { k_write<<<blocks, 256>>>((u32x4*)ar, n16); });
        const double tra = time_ms([&] { k_read<<<blocks, 256>>>((const u32x4*)ar, n16, sink); });
        const double tws = 0;
        printf("probe stream %p arena %p index %p : K12 %.3f ms | copy s->a %.3f (%.0f GB/s) read s %.3f (%.0f) write a %.3f (%.0f) read a %.3f (%.0f)\n", sb, ar, ix, m,
               tc, 2.0 * st.n / tc / 1e6, tr, st.n / tr / 1e6, tw, st.n / tw / 1e6, tra, st.n / tra / 1e6);
        (void)tws;
        /* K12 on pieces: the same bytes, the arena piece at the same distance */
        const uint64_t piece = 1ull << 30;
        printf("pieces (ms per GiB):");
        for (uint64_t o = 0; o + piece <= st.n; o += piece) {
            Stream sub{sb + o, piece, 0, 0};
            std::vector<float> ms;
            for (int i = 0; i < 4; ++i) {
                HB(hbs_index_extract(ctx, sub.bytes, sub.n, ix, nals + 8, ar + o, piece + 4096, dsum));
                float t = 0; HB(hbs_ctx_kernel_ms(ctx, &t)); if (i) ms.push_back(t);
            }
            std::sort(ms.begin(), ms.end());
            printf(" %.3f", ms[1]);
        }
        printf("\n");
        /* the plain copy on the same pieces */
        printf("copy pieces (ms per GiB):");
        for (uint64_t o = 0; o + piece <= st.n; o += piece) {
            const size_t pn = piece / 16;
            const double t = time_ms([&] { k_copy<<<(unsigned)(pn / 256), 256>>>((const u32x4*)(sb + o), (u32x4*)(ar + o), pn); }, 3);
            printf(" %.3f", t);
        }
        printf("\n");
    } else if (!strcmp(mode, "matrix")) {
        /* one stream, two arenas: K12 with each; then K12 and the plain copy from 1 GiB piece i of the stream to piece j of an arena */
        uint8_t *sb, *ar[2]; hbs_nal_entry* ix;
        CK(hipMalloc((void**)&sb, scap));
        Stream st = make_stream(sb, scap, nals);
        CK(hipMalloc((void**)&ar[0], acap)); CK(hipMalloc((void**)&ix, icap)); CK(hipMalloc((void**)&ar[1], acap));
        for (int a = 0; a < 2; ++a) {
            const double m = run(st, ar[a], acap, ix, dsum, reps, &lo, &hi);
            printf("matrix stream %p arena%d %p : K12 %.3f ms (%.3f .. %.3f)\n", sb, a, ar[a], m, lo, hi);
        }
        const uint64_t piece = 1ull << 30;
        const int np = (int)(st.n / piece);
        for (int a = 0; a < 2; ++a)
            for (int i = 0; i < np; i += 7) {
                printf("K12  s[%2d] -> arena%d[j]:", i, a);
                for (int j = 0; j < np; ++j) {
                    std::vector<float> ms;
                    for (int r = 0; r < 4; ++r) {
                        HB(hbs_index_extract(ctx, sb + (uint64_t)i * piece, piece, ix, nals + 8, ar[a] + (uint64_t)j * piece, piece + 4096, dsum));
                        float t = 0; HB(hbs_ctx_kernel_ms(ctx, &t)); if (r) ms.push_back(t);
                    }
                    std::sort(ms.begin(), ms.end());
                    printf(" %.3f", ms[1]);
                }
                printf("\n");
                CK(hipDeviceSynchronize());
                printf("copy s[%2d] -> arena%d[j]:", i, a);
                for (int j = 0; j < np; ++j) {
                    const size_t pn = piece / 16;
                    const double t = time_ms([&] { k_copy<<<(unsigned)(pn / 256), 256>>>((const u32x4*)(sb + (uint64_t)i * piece), (u32x4*)(ar[a] + (uint64_t)j * piece), pn); }, 3);
                    printf(" %.3f", t);
                }
                printf("\n");
            }
        /* inside one allocation: stream piece i -> stream piece j would destroy the stream; arena0[i] -> arena1[j] and arena0[i] -> arena0[j] */
        for (int i = 0; i < np; i += 7) {
            printf("copy arena0[%2d] -> arena1[j]:", i);
            for (int j = 0; j < np; ++j) {
                const size_t pn = piece / 16;
                const double t = time_ms([&] { k_copy<<<(unsigned)(pn / 256), 256>>>((const u32x4*)(ar[0] + (uint64_t)i * piece), (u32x4*)(ar[1] + (uint64_t)j * piece), pn); }, 3);
                printf(" %.3f", t);
            }
            printf("\n");
            printf("copy arena0[%2d] -> arena0[j]:", i);
            for (int j = 0; j < np; ++j) {
                if (j == i) { printf("   -  "); continue; }
                const size_t pn = piece / 16;
                const double t = time_ms([&] { k_copy<<<(unsigned)(pn / 256), 256>>>((const u32x4*)(ar[0] + (uint64_t)i * piece), (u32x4*)(ar[0] + (uint64_t)j * piece), pn); }, 3);
                printf(" %.3f", t);
            }
            printf("\n");
        }
    } else if (!strcmp(mode, "delta") || !strcmp(mode, "chunks")) {
        uint8_t* sb; hbs_nal_entry* ix;
        CK(hipMalloc((void**)&sb, scap));
        Stream st = make_stream(sb, scap, nals);
        CK(hipMalloc((void**)&ix, icap));
        const uint64_t piece = 1ull << 30;
        auto k12_piece = [&](const uint8_t* src, uint8_t* dst) {
            std::vector<float> ms;
            for (int r = 0; r < 6; ++r) {
                HB(hbs_index_extract(ctx, src, piece, ix, nals + 8, dst, piece + 4096, dsum));
                float t = 0; HB(hbs_ctx_kernel_ms(ctx, &t)); if (r) ms.push_back(t);
            }
            std::sort(ms.begin(), ms.end());
            return ms[2];
        };
        if (!strcmp(mode, "delta")) {
            /* ONE allocation: 1 GiB of stream at its start, the arena piece at 1 GiB + skew behind it */
            uint8_t* b; CK(hipMalloc((void**)&b, 5ull << 30));
            CK(hipMemcpy(b, sb, piece, hipMemcpyDeviceToDevice));
            printf("delta: buffer %p\n16 MiB steps:", b);
            for (int k = 0; k < 64; ++k) printf(" %.3f", k12_piece(b, b + piece + (uint64_t)k * (16ull << 20)));
            printf("\n1 GiB steps:");
            for (int k = 0; k < 3; ++k) printf(" %.3f", k12_piece(b, b + piece + (uint64_t)k * (1ull << 30)));
            printf("\n256 KiB steps:");
            for (int k = 0; k < 64; ++k) printf(" %.3f", k12_piece(b, b + piece + (uint64_t)k * (256ull << 10)));
            printf("\n4 KiB steps:");
            for (int k = 0; k < 32; ++k) printf(" %.3f", k12_piece(b, b + piece + (uint64_t)k * 4096ull));
            printf("\nsource moved, 16 MiB steps (arena at 3 GiB):");
            CK(hipMemcpy(b, sb, 2 * piece, hipMemcpyDeviceToDevice));
            for (int k = 0; k < 64; ++k) printf(" %.3f", k12_piece(b + (uint64_t)k * (16ull << 20), b + 3 * piece));
            printf("\n");
        } else {
            /* arena pieces that are allocations of their own: 12 x hipMalloc(1 GiB + 2 MiB), 12 x hipMemCreate(1 GiB) */
            printf("chunks: hipMalloc pieces:");
            std::vector<uint8_t*> ps;
            for (int k = 0; k < 12; ++k) { uint8_t* q; CK(hipMalloc((void**)&q, piece + (2ull << 20))); ps.push_back(q); }
            for (int k = 0; k < 12; ++k) printf(" %.3f", k12_piece(sb + 3 * piece, ps[k]));
            printf("\n  again with stream piece 9:");
            for (int k = 0; k < 12; ++k) printf(" %.3f", k12_piece(sb + 9 * piece, ps[k]));
            printf("\n");
            for (auto q : ps) CK(hipFree(q));
            hipMemAllocationProp prop; memset(&prop, 0, sizeof(prop));
            prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
            const uint64_t csz = piece + (2ull << 20);
            void* va = nullptr; CK(hipMemAddressReserve(&va, 12 * csz, 2ull << 20, nullptr, 0));
            for (int k = 0; k < 12; ++k) { hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, csz, &prop, 0)); CK(hipMemMap((uint8_t*)va + k * csz, csz, 0, h, 0)); }
            hipMemAccessDesc acc; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
            CK(hipMemSetAccess(va, 12 * csz, &acc, 1));
            printf("chunks: hipMemCreate pieces:");
            for (int k = 0; k < 12; ++k) printf(" %.3f", k12_piece(sb + 3 * piece, (uint8_t*)va + k * csz));
            printf("\n  again with stream piece 9:");
            for (int k = 0; k < 12; ++k) printf(" %.3f", k12_piece(sb + 9 * piece, (uint8_t*)va + k * csz));
            printf("\n");
        }
    } else if (!strcmp(mode, "pairs")) {
        /* N physical chunks of 1 GiB (+ 2 MiB): K12 from chunk i to chunk j for every pair -- is "slow" a property of the pair, and do
         * the chunks fall into classes (slow inside a class, fast across)? */
        uint8_t* sb; hbs_nal_entry* ix;
        CK(hipMalloc((void**)&sb, scap));
        Stream st = make_stream(sb, scap, nals);
        CK(hipMalloc((void**)&ix, icap));
        const uint64_t piece = 1ull << 30, csz = piece + (2ull << 20);
        const int N = getenv("PAIRS_N") ? atoi(getenv("PAIRS_N")) : 16;
        hipMemAllocationProp prop; memset(&prop, 0, sizeof(prop));
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        void* va = nullptr; CK(hipMemAddressReserve(&va, (uint64_t)N * csz, 2ull << 20, nullptr, 0));
        for (int k = 0; k < N; ++k) { hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, csz, &prop, 0)); CK(hipMemMap((uint8_t*)va + k * csz, csz, 0, h, 0)); }
        hipMemAccessDesc acc; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(va, (uint64_t)N * csz, &acc, 1));
        auto chunk = [&](int k) { return (uint8_t*)va + (uint64_t)k * csz; };
        std::vector<std::vector<float>> T(N, std::vector<float>(N, 0.f));
        for (int i = 0; i < N; ++i) {
            for (int j = 0; j < N; ++j) {
                if (i == j) continue;
                CK(hipMemcpyAsync(chunk(i), sb + 2 * piece, piece, hipMemcpyDeviceToDevice, (hipStream_t)hbs_ctx_get_stream(ctx)));
                std::vector<float> ms;
                for (int r = 0; r < 5; ++r) {
                    HB(hbs_index_extract(ctx, chunk(i), piece, ix, nals + 8, chunk(j), piece + 4096, dsum));
                    float t = 0; HB(hbs_ctx_kernel_ms(ctx, &t)); if (r) ms.push_back(t);
                }
                std::sort(ms.begin(), ms.end());
                T[i][j] = ms[1];
            }
        }
        printf("pairs: K12 ms per GiB, row = source chunk, column = destination chunk\n");
        for (int i = 0; i < N; ++i) { for (int j = 0; j < N; ++j) { if (i == j) printf("   -  "); else printf(" %.3f", T[i][j]); } printf("\n"); }
        printf("as classes (S = slower than the midpoint of the matrix' range):\n");
        float mn = 1e9f, mx = 0; for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) if (i != j) { mn = std::min(mn, T[i][j]); mx = std::max(mx, T[i][j]); }
        for (int i = 0; i < N; ++i) { for (int j = 0; j < N; ++j) printf("%c", i == j ? '-' : (T[i][j] > 0.5f * (mn + mx) ? 'S' : '.')); printf("\n"); }
        /* the stream allocation against every chunk, both directions */
        printf("stream piece 2 -> chunk j:");
        for (int j = 0; j < N; ++j) {
            std::vector<float> ms;
            for (int r = 0; r < 5; ++r) { HB(hbs_index_extract(ctx, sb + 2 * piece, piece, ix, nals + 8, chunk(j), piece + 4096, dsum)); float t = 0; HB(hbs_ctx_kernel_ms(ctx, &t)); if (r) ms.push_back(t); }
            std::sort(ms.begin(), ms.end()); printf(" %.3f", ms[1]);
        }
        printf("\n");
    } else if (!strcmp(mode, "probekernels")) {
        /* which simple kernel tells the two classes of physical memory apart as well as K12 does? */
        uint8_t* sb; hbs_nal_entry* ix; unsigned* ticket;
        CK(hipMalloc((void**)&sb, scap));
        Stream st = make_stream(sb, scap, nals);
        CK(hipMalloc((void**)&ix, icap)); CK(hipMalloc((void**)&ticket, 64));
        const uint64_t piece = 1ull << 30, csz = piece + (2ull << 20);
        const int N = 10;
        hipMemAllocationProp prop; memset(&prop, 0, sizeof(prop));
        prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
        void* va = nullptr; CK(hipMemAddressReserve(&va, (uint64_t)N * csz, 2ull << 20, nullptr, 0));
        for (int k = 0; k < N; ++k) { hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, csz, &prop, 0)); CK(hipMemMap((uint8_t*)va + k * csz, csz, 0, h, 0)); }
        hipMemAccessDesc acc; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(va, (uint64_t)N * csz, &acc, 1));
        auto chunk = [&](int k) { return (uint8_t*)va + (uint64_t)k * csz; };
        auto k12 = [&](const uint8_t* src, uint8_t* dst, uint64_t len) {
            std::vector<float> ms;
            for (int r = 0; r < 5; ++r) { HB(hbs_index_extract(ctx, src, len, ix, nals + 8, dst, len + 4096, dsum)); float t = 0; HB(hbs_ctx_kernel_ms(ctx, &t)); if (r) ms.push_back(t); }
            std::sort(ms.begin(), ms.end()); return (double)ms[1];
        };
        CK(hipMemcpy(chunk(0), sb + 2 * piece, piece, hipMemcpyDeviceToDevice));
        int same = -1, cross = -1; double ts = 0, tc = 0;
        printf("K12 chunk 0 -> j:");
        for (int j = 1; j < N; ++j) { const double t = k12(chunk(0), chunk(j), piece); printf(" %.3f", t); if (t > 0.416 && same < 0) { same = j; ts = t; } if (t < 0.412 && cross < 0) { cross = j; tc = t; } }
        printf("\n");
        if (same < 0 || cross < 0) { printf("only one class among the chunks: run again\n"); return 0; }
        printf("same class: chunk %d (%.3f), other class: chunk %d (%.3f): K12 ratio %.4f\n", same, ts, cross, tc, ts / tc);
        const size_t pn = piece / 16;
        auto report = [&](const char* name, auto launch) {
            CK(hipDeviceSynchronize());
            const double a = time_ms([&] { launch(chunk(same)); }, 7), b = time_ms([&] { launch(chunk(cross)); }, 7);
            const double a2 = time_ms([&] { launch(chunk(same)); }, 7), b2 = time_ms([&] { launch(chunk(cross)); }, 7);
            printf("%-34s same %.4f %.4f  cross %.4f %.4f  ratio %.4f\n", name, a, a2, b, b2, (a + a2) / (b + b2));
        };
        report("copy aligned", [&](uint8_t* d) { k_copy<<<(unsigned)(pn / 256), 256>>>((const u32x4*)chunk(0), (u32x4*)d, pn); });
        report("copy dst+7", [&](uint8_t* d) { k_copy_mis<<<(unsigned)(pn / 256), 256>>>(chunk(0), d, pn, 7); });
        report("copy dst+32", [&](uint8_t* d) { k_copy_mis<<<(unsigned)(pn / 256), 256>>>(chunk(0), d, pn, 32); });
        report("tile 48 rows, burst, aligned", [&](uint8_t* d) { CK(hipMemsetAsync(ticket, 0, 4, 0)); k_tilecopy<48, 0><<<512, 256>>>(chunk(0), d, piece / (192 << 10), 0, ticket); });
        report("tile 48 rows, burst, +7", [&](uint8_t* d) { CK(hipMemsetAsync(ticket, 0, 4, 0)); k_tilecopy<48, 0><<<512, 256>>>(chunk(0), d, piece / (192 << 10), 7, ticket); });
        report("tile 48 rows, depth 3, +7", [&](uint8_t* d) { CK(hipMemsetAsync(ticket, 0, 4, 0)); k_tilecopy<48, 3><<<512, 256>>>(chunk(0), d, piece / (192 << 10), 7, ticket); });
        report("tile 48 rows, depth 3, aligned", [&](uint8_t* d) { CK(hipMemsetAsync(ticket, 0, 4, 0)); k_tilecopy<48, 3><<<512, 256>>>(chunk(0), d, piece / (192 << 10), 0, ticket); });
        report("tile 16 rows, burst, +7", [&](uint8_t* d) { CK(hipMemsetAsync(ticket, 0, 4, 0)); k_tilecopy<16, 0><<<1024, 256>>>(chunk(0), d, piece / (64 << 10), 7, ticket); });
        /* smaller probes: how short can a decision be? */
        for (uint64_t len : {64ull << 20, 256ull << 20}) {
            const double a = k12(chunk(0), chunk(same), len), b = k12(chunk(0), chunk(cross), len);
            printf("K12 on %llu MiB: same %.4f cross %.4f ratio %.4f\n", (unsigned long long)(len >> 20), a, b, a / b);
        }
        /* the way back (K3): RBSP in chunk 0, stream written to chunk j */
        {
            hbs_nal_entry* ix2; CK(hipMalloc((void**)&ix2, icap));
            uint8_t* ar; CK(hipMalloc((void**)&ar, piece + 4096));
            HB(hbs_index_extract(ctx, chunk(0), piece, ix, nals + 8, ar, piece + 4096, dsum));
            hbs_summary s; HB(hbs_read_summary(ctx, dsum, &s));
            const uint64_t m = s.nal_count, rb = s.rbsp_bytes;
            CK(hipMemcpy(chunk(0), ar, rb, hipMemcpyDeviceToDevice));
            for (int which = 0; which < 2; ++which) {
                uint8_t* d = chunk(which ? cross : same);
                std::vector<double> ts2;
                for (int r = 0; r < 5; ++r) {
                    ts2.push_back(time_ms([&] { HB(hbs_emit_annexb(ctx, chunk(0), rb, ix, m, 1, d, piece + (1 << 20), ix2, dsum)); }, 1));
                }
                std::sort(ts2.begin(), ts2.end());
                printf("K3 emit (call) -> %s chunk: %.4f ms (%.4f .. %.4f)\n", which ? "other-class" : "same-class", ts2[2], ts2[0], ts2[4]);
            }
        }
    } else { fprintf(stderr, "unknown mode %s\n", mode); return 1; }
    hbs_ctx_destroy(ctx);
    return 0;
}
