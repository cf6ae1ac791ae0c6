// Microbenchmark, round 3: the memory side of K12 (hbs_scan4.hip) with the compute replaced by timed waits, to find
// out which change of STRUCTURE moves the ceiling before porting it into the real kernel.  Dev aid; results in
// profiles/r03/ceiling3.txt.  16 GiB each way (the bench stream's size), non-temporal loads and stores as in K12.
//
//   simple        one 16-byte chunk per thread, huge grid (the guide's float4 copy), aligned / shifted by 7
//   tile          K12's geometry: ticketed 4 x ROWS KiB tiles, all rows in registers between load and store
//     ILV = 1     rows dealt to the four wavefronts round-robin (wavefront w owns rows w, w+4, ...): one 4 KiB front
//     (round 3 also tried taking the next ticket before the copy and loading row r of the next tile right behind the
//      store of row r: 4-10 % slower in every setting, profiles/r03/ceiling3_c1.txt; removed)
//     THR/THRS    at most that many loads / stores of a wavefront in flight (s_waitcnt vmcnt behind each)
//     PROG        the flag pass under the fetch, four rows at a time
//   skel          the same plus K12's serial chain: every wavefront waits `dflag` cycles behind its loads (flag pass),
//                 wavefront 0 `dserA` more (elements), publishes an aggregate, resolves a REAL decoupled look-back over
//                 64 predecessors per step (or none: fake), waits `dserB` (index entries), then everybody stores at the
//                 offset the look-back gave (byte-misaligned, different per tile)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include "hbs_elems.h"        /* K12's own decoupled look-back (lb 3) */
using hbs::u32x4;
typedef u32x4 u32x4_u1 __attribute__((aligned(1)));
typedef const __attribute__((address_space(1))) u32x4* gptr;

__device__ __forceinline__ u32x4 ld_nt(const uint8_t* p) { return __builtin_nontemporal_load((gptr)(uintptr_t)p); }
__device__ __forceinline__ void st_nt(uint8_t* p, u32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<u32x4_u1*>(p)); }

__global__ void k_fill(uint64_t* p, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) {
        uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        p[i] = z ^ (z >> 31);
    }
}

__global__ void k_copy_simple(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t nchunks, int shift)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nchunks) return;
    st_nt(dst + shift + 16 * i, ld_nt(src + 16 * i));
}

__device__ __forceinline__ void busy(uint32_t cycles, uint32_t& sink)
{
    if (cycles == 0) return;
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < cycles) { sink = sink * 1664525u + 1013904223u; sink ^= sink >> 7; sink += 3; sink ^= sink << 3; }
}

__device__ __forceinline__ uint64_t ld_desc(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_desc(unsigned long long* p, uint64_t v) { __hip_atomic_store(p, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <class T> __device__ __forceinline__ T* uni(T* p)
{
    const uint64_t u = (uint64_t)p;
    return (T*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(u >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)u));
}

/* THR > 0: at most THR vector-memory operations of a wavefront in flight while it issues a tile's loads or stores, so that
 * the CU's memory queue stays short for the other workgroup's look-back polls */
template <int THR> __device__ __forceinline__ void throttle()
{
    if (THR > 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(THR) : "memory");
}

struct SkelArgs {
    const uint8_t* src; uint8_t* dst; size_t ntiles; unsigned* ticket; unsigned long long* desc;
    uint32_t dflag, dserA, dserB; int lb;      /* lb: 0 none (offset = tile start), 1 simple look-back (8-byte words, 64 per step),
                                                  2 fake (no wait, computed offset), 3 K12's look-back (hbs_elems.h: 16-byte descriptors, 256 per step) */
    hbs::RunHeader* hdr;
    unsigned long long* stats;                 /* [0] look-back polls, [1] tiles */
};

// offsets inside a tile: row r of wavefront w
template <int WAVES, int ROWS, int ILV> __device__ __forceinline__ size_t row_off(int w, int r)
{
    return ILV ? (size_t)(WAVES * r + w) * 1024 : (size_t)(w * ROWS + r) * 1024;
}

/* THR / THRS: throttle() depth behind every load / store of a tile.  PROG = 1: the flag pass runs under the fetch, four rows
 * at a time: while group g's rows are awaited and flagged (dflag / groups cycles), group g+1's loads are in flight. */
template <int WAVES, int ROWS, int ILV, int PROG, int MINW, int THR = 0, int THRS = THR>
__global__ __launch_bounds__(64 * WAVES, MINW)
void k_skel(SkelArgs a)
{
    __shared__ unsigned tk;
    __shared__ unsigned long long ex_sh;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr size_t tile_bytes = (size_t)WAVES * ROWS * 1024;
    uint32_t sink = threadIdx.x;
    unsigned long long polls = 0, tiles = 0;
    for (;;) {
        if (threadIdx.x == 0) tk = atomicAdd(a.ticket, 1u);
        __syncthreads();
        const size_t t = tk;
        if (t >= a.ntiles) break;
        u32x4 v[ROWS];
        const uint8_t* sp = a.src + t * tile_bytes + 16 * lane;
        if (PROG) {
            static_assert(ROWS % 4 == 0, "groups of four rows");
#pragma unroll
            for (int g = 0; g < ROWS / 4; ++g) {
#pragma unroll
                for (int r = 4 * g; r < 4 * g + 4; ++r) v[r] = ld_nt(sp + row_off<WAVES, ROWS, ILV>(wave, r));
                if (g > 0) { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); busy(a.dflag * 4 / ROWS, sink); }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            busy(a.dflag * 4 / ROWS, sink);
        } else {
#pragma unroll
            for (int r = 0; r < ROWS; ++r) { v[r] = ld_nt(sp + row_off<WAVES, ROWS, ILV>(wave, r)); throttle<THR>(); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            busy(a.dflag, sink);
        }
        __syncthreads();
        if (wave == 0) {
            busy(a.dserA, sink);
            const uint64_t agg = tile_bytes - ((t * 2654435761ull >> 7) & 63ull);
            uint64_t excl = 0;
            if (a.lb == 3) {
                hbs::TileAgg ta; ta.cnt = 16; ta.known = (uint32_t)agg; ta.sig = 0; ta.last = hbs::kKindStart;
                hbs::Prefix ex; uint32_t it, stl;
                hbs::look_back4(a.desc, t, ta, a.hdr, lane, ex, it, stl);
                excl = ex.kept; polls += it;
            } else if (a.lb == 2) {
                excl = t * (tile_bytes - 31);
            } else {
                excl = t * tile_bytes;
            }
            busy(a.dserB, sink);
            if (lane == 0) ex_sh = excl;
        }
        __syncthreads();
        uint8_t* d = a.dst + ex_sh + 16 * lane;
        ++tiles;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) { st_nt(d + row_off<WAVES, ROWS, ILV>(wave, r), v[r]); throttle<THRS>(); }
    }
    if (sink == 0x12345678u) a.dst[0] = 1;
    if (lane == 0 && wave == 0 && a.stats) { atomicAdd(&a.stats[0], polls); atomicAdd(&a.stats[1], tiles); }
}

// ---- reads (index-only scan, hbs_scan5.hip): who reads what, with non-temporal loads -------------------------------
// tile:  a wavefront (alone in its workgroup) takes a TILE_KB tile by ticket and streams it SPAN KiB at a time, the next span in flight
// coop:  a workgroup of WAVES wavefronts takes a tile by ticket; its spans are dealt to the wavefronts round-robin
// sweep: no tiles: spans dealt grid-wide round-robin (the fastest read pattern of profiles/r02/ceiling2.txt)
template <int SPAN, int TILE_KB, int WAVES, int DEPTH2>
__global__ __launch_bounds__(64 * WAVES)
void k_read_coop(const uint8_t* __restrict__ src, uint32_t* __restrict__ out, size_t ntiles, unsigned* ticket)
{
    __shared__ unsigned tk;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t acc = 0;
    constexpr int spans = TILE_KB / SPAN, mine = spans / WAVES;
    static_assert(spans % WAVES == 0, "whole spans per wavefront");
    for (;;) {
        if (WAVES > 1) __syncthreads();
        if (threadIdx.x == 0) tk = atomicAdd(ticket, 1u);
        if (WAVES > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();
        const size_t t = (unsigned)__builtin_amdgcn_readfirstlane((int)tk);
        if (t >= ntiles) break;
        const uint8_t* base = src + t * (size_t)TILE_KB * 1024 + 16 * lane;
        u32x4 cur[SPAN], nxt[SPAN];
#pragma unroll
        for (int r = 0; r < SPAN; ++r) nxt[r] = ld_nt(base + ((size_t)wave * SPAN + r) * 1024);
#pragma unroll 1
        for (int i = 0; i < mine; ++i) {
#pragma unroll
            for (int r = 0; r < SPAN; ++r) cur[r] = nxt[r];
            if (i + 1 < mine) {
#pragma unroll
                for (int r = 0; r < SPAN; ++r) nxt[r] = ld_nt(base + ((size_t)((i + 1) * WAVES + wave) * SPAN + r) * 1024);
            }
#pragma unroll
            for (int r = 0; r < SPAN; ++r) acc += cur[r].x ^ cur[r].y ^ cur[r].z ^ cur[r].w;
        }
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}
template <int SPAN>
__global__ __launch_bounds__(256)
void k_read_sweep(const uint8_t* __restrict__ src, uint32_t* __restrict__ out, size_t nspans)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t acc = 0;
    const size_t stride = (size_t)gridDim.x * 4;
    size_t s = (size_t)blockIdx.x * 4 + wave;
    u32x4 cur[SPAN], nxt[SPAN];
    if (s < nspans) {
#pragma unroll
        for (int r = 0; r < SPAN; ++r) nxt[r] = ld_nt(src + (s * SPAN + r) * 1024 + 16 * lane);
    }
    for (; s < nspans; s += stride) {
#pragma unroll
        for (int r = 0; r < SPAN; ++r) cur[r] = nxt[r];
        if (s + stride < nspans) {
#pragma unroll
            for (int r = 0; r < SPAN; ++r) nxt[r] = ld_nt(src + ((s + stride) * SPAN + r) * 1024 + 16 * lane);
        }
#pragma unroll
        for (int r = 0; r < SPAN; ++r) acc += cur[r].x ^ cur[r].y ^ cur[r].z ^ cur[r].w;
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

template <class F> float time_ms(F f, int reps = 4)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) f();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms / reps;
}

int main(int argc, char** argv)
{
    const size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 16;
    const size_t n = gib << 30;
    uint8_t *src, *dst; unsigned* ticket; unsigned long long *desc, *stats;
    hipMalloc(&src, n + 4096); hipMalloc(&dst, n + 4096); hipMalloc(&ticket, 256);
    hipMalloc(&desc, 16 * (n / 32768 + 64)); hipMalloc(&stats, 64);
    k_fill<<<4096, 256>>>((uint64_t*)src, (n + 4096) / 8);
    hipMemset(dst, 0, n + 4096);
    hipDeviceSynchronize();
    const size_t nc = n / 16;
    auto rw = [&](const char* name, float ms, const char* extra = "") {
        printf("copy %-58s %7.0f GB/s r+w  (%.3f ms) %s\n", name, 2.0 * n / ms / 1e6, ms, extra); fflush(stdout);
    };
    for (int shift : {0, 7}) {
        char nm[96];
        snprintf(nm, 96, "simple nt-both shift %d", shift);
        rw(nm, time_ms([&] { k_copy_simple<<<(unsigned)(nc / 256), 256>>>(src, dst, nc, shift); }));
    }
    hbs::RunHeader* hdr; hipMalloc(&hdr, sizeof(hbs::RunHeader)); hipMemset(hdr, 0, sizeof(hbs::RunHeader));
    auto run = [&](const char* what, void (*k)(SkelArgs), int waves, int rows, int per_cu, uint32_t dflag, uint32_t dA, uint32_t dB, int lb) {
        SkelArgs a;
        a.src = src; a.dst = dst; a.ntiles = n / ((size_t)waves * rows * 1024); a.ticket = ticket; a.desc = desc; a.hdr = hdr;
        a.dflag = dflag; a.dserA = dA; a.dserB = dB; a.lb = lb; a.stats = stats;
        hipMemset(stats, 0, 16);
        int occ = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, 64 * waves, 0);
        const int grid = 256 * (occ < per_cu ? occ : per_cu);
        const float ms = time_ms([&] {
            hipMemsetAsync(ticket, 0, 4);
            if (lb == 1 || lb == 3) hipMemsetAsync(desc, 0, 16 * (a.ntiles + 1));
            k<<<grid, 64 * waves>>>(a);
        }, 4);
        unsigned long long st[2];
        hipMemcpy(st, stats, 16, hipMemcpyDeviceToHost);
        char nm[128], ex[64];
        snprintf(nm, 128, "%s %dw x %dr x %d/CU flag %u serA %u serB %u lb %d", what, waves, rows, grid / 256, dflag, dA, dB, lb);
        snprintf(ex, 64, "polls/tile %.2f", st[1] ? (double)st[0] / (double)st[1] : 0.0);
        rw(nm, ms, ex);
    };
    if (argc > 2 && !strcmp(argv[2], "shift")) {
        // what alignment of the destination costs a copy: bytes, 16-byte granules, 64-byte sectors, 128-byte lines
        for (int rep = 0; rep < 2; ++rep)
            for (int shift : {0, 1, 7, 16, 32, 48, 64, 96, 128, 144, 256, 1024, 1031}) {
                char nm[96];
                snprintf(nm, 96, "simple nt-both shift %d", shift);
                rw(nm, time_ms([&] { k_copy_simple<<<(unsigned)(nc / 256), 256>>>(src, dst, nc, shift); }));
            }
        hipDeviceSynchronize();
        return 0;
    }
    if (argc > 2 && !strcmp(argv[2], "read")) {
        uint32_t* outp; hipMalloc(&outp, 4096);
        auto ro = [&](const char* name, float ms) { printf("read %-44s %7.0f GB/s  (%.3f ms)\n", name, 1.0 * n / ms / 1e6, ms); fflush(stdout); };
        for (int rep = 0; rep < 2; ++rep) {
            ro("tile 1 MiB span16, 1 wave, x5120", time_ms([&] { hipMemsetAsync(ticket, 0, 4); k_read_coop<16, 1024, 1, 0><<<5120, 64>>>(src, outp, n >> 20, ticket); }));
            ro("tile 256 KiB span16, 1 wave, x5120", time_ms([&] { hipMemsetAsync(ticket, 0, 4); k_read_coop<16, 256, 1, 0><<<5120, 64>>>(src, outp, n >> 18, ticket); }));
            ro("tile 1 MiB span8, 1 wave, x8192", time_ms([&] { hipMemsetAsync(ticket, 0, 4); k_read_coop<8, 1024, 1, 0><<<8192, 64>>>(src, outp, n >> 20, ticket); }));
            ro("tile 1 MiB span16, 4 waves coop, x1280", time_ms([&] { hipMemsetAsync(ticket, 0, 4); k_read_coop<16, 1024, 4, 0><<<1280, 256>>>(src, outp, n >> 20, ticket); }));
            ro("tile 1 MiB span16, 16 waves coop, x256", time_ms([&] { hipMemsetAsync(ticket, 0, 4); k_read_coop<16, 1024, 16, 0><<<256, 1024>>>(src, outp, n >> 20, ticket); }));
            ro("tile 1 MiB span16, 8 waves coop, x512", time_ms([&] { hipMemsetAsync(ticket, 0, 4); k_read_coop<16, 1024, 8, 0><<<512, 512>>>(src, outp, n >> 20, ticket); }));
            ro("tile 4 MiB span16, 16 waves coop, x256", time_ms([&] { hipMemsetAsync(ticket, 0, 4); k_read_coop<16, 4096, 16, 0><<<256, 1024>>>(src, outp, n >> 22, ticket); }));
            ro("sweep span16 grid 1024", time_ms([&] { k_read_sweep<16><<<1024, 256>>>(src, outp, n >> 14); }));
            ro("sweep span16 grid 1280", time_ms([&] { k_read_sweep<16><<<1280, 256>>>(src, outp, n >> 14); }));
            ro("sweep span16 grid 2048", time_ms([&] { k_read_sweep<16><<<2048, 256>>>(src, outp, n >> 14); }));
            ro("sweep span8 grid 2048", time_ms([&] { k_read_sweep<8><<<2048, 256>>>(src, outp, n >> 13); }));
            ro("sweep span4 grid 2048", time_ms([&] { k_read_sweep<4><<<2048, 256>>>(src, outp, n >> 12); }));
        }
        hipDeviceSynchronize();
        return 0;
    }
#define GEO(W, R, PC, MINW, PROG, TL, TS, ...) run("prog " #PROG " thr ld " #TL " st " #TS, k_skel<W, R, 0, PROG, MINW, TL, TS>, W, R, PC, __VA_ARGS__)
    for (int rep = 0; rep < 2; ++rep) {
        // store depth 3, load depth swept
        GEO(4, 48, 2, 2, 0, 0, 0, 2000, 4000, 7000, 3);
        GEO(4, 48, 2, 2, 0, 3, 3, 2000, 4000, 7000, 3);
        GEO(4, 48, 2, 2, 0, 6, 3, 2000, 4000, 7000, 3);
        GEO(4, 48, 2, 2, 0, 8, 3, 2000, 4000, 7000, 3);
        GEO(4, 48, 2, 2, 0, 12, 3, 2000, 4000, 7000, 3);
        GEO(4, 48, 2, 2, 0, 16, 3, 2000, 4000, 7000, 3);
        GEO(4, 48, 2, 2, 0, 0, 3, 2000, 4000, 7000, 3);
        GEO(4, 48, 2, 2, 0, 6, 2, 2000, 4000, 7000, 3);
        GEO(4, 48, 2, 2, 0, 6, 4, 2000, 4000, 7000, 3);
        // the flag pass under the fetch (4 loads issued, wait for the previous 4, 1000 cycles of flags), all 12000 cycles of it
        GEO(4, 48, 2, 2, 1, 0, 0, 12000, 4000, 7000, 3);
        GEO(4, 48, 2, 2, 1, 0, 3, 12000, 4000, 7000, 3);
        GEO(4, 48, 2, 2, 1, 0, 4, 12000, 4000, 7000, 3);
        GEO(4, 48, 2, 2, 1, 0, 3, 12000, 4000, 0, 3);
        GEO(4, 48, 2, 2, 1, 0, 3, 8000, 4000, 7000, 3);
        GEO(4, 48, 2, 2, 1, 0, 3, 12000, 4000, 7000, 2);
        GEO(4, 32, 3, 3, 1, 0, 2, 8000, 3000, 5000, 3);
        GEO(4, 32, 3, 3, 1, 0, 3, 8000, 3000, 5000, 3);
        GEO(4, 32, 2, 2, 1, 0, 3, 8000, 3000, 5000, 3);
        GEO(4, 40, 2, 2, 1, 0, 3, 10000, 3500, 6000, 3);
    }
    hipError_t e = hipDeviceSynchronize();
    printf("status: %s\n", hipGetErrorString(e));
    return e == hipSuccess ? 0 : 1;
}
