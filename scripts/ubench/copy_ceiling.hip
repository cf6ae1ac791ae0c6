// Microbenchmark: what a plain HBM->HBM copy reaches on this GPU with several loads in flight per
// lane, with/without nontemporal hints, with a byte-misaligned destination (dev aid; calibrates the
// practical ceiling for K12, which reads the stream once and writes ~all of it once).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(1))) U16 { u32x4 v; };

template <int U, bool NT>
__global__ void k_copy(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t nchunks, int shift)
{
    // each workgroup takes contiguous spans of U KiB per wavefront; grid-stride over spans
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const size_t span = (size_t)U * 64;                      // chunks per wave-span
    size_t s = ((size_t)blockIdx.x * wpb + wave) * span;
    const size_t stride = (size_t)gridDim.x * wpb * span;
    for (; s + span <= nchunks; s += stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const u32x4* p = reinterpret_cast<const u32x4*>(src + 16 * (s + (size_t)u * 64 + lane));
            v[u] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            U16* q = reinterpret_cast<U16*>(dst + shift + 16 * (s + (size_t)u * 64 + lane));
            if (NT && shift == 0) __builtin_nontemporal_store(v[u], reinterpret_cast<u32x4*>(q)); else q->v = v[u];
        }
    }
}
template <int U>
__global__ void k_read(const uint8_t* __restrict__ src, uint32_t* __restrict__ out, size_t nchunks)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const size_t span = (size_t)U * 64;
    size_t s = ((size_t)blockIdx.x * wpb + wave) * span;
    const size_t stride = (size_t)gridDim.x * wpb * span;
    uint32_t acc = 0;
    for (; s + span <= nchunks; s += stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const u32x4*>(src + 16 * (s + (size_t)u * 64 + lane));
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}
template <class F> float time_ms(F f)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) f();
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}
int main()
{
    const size_t n = 4ull << 30;
    uint8_t *src, *dst; uint32_t* out;
    hipMalloc(&src, n + 64); hipMalloc(&dst, n + 64); hipMalloc(&out, 4096);
    hipMemset(src, 1, n + 64); hipMemset(dst, 0, n + 64);
    const size_t nc = n / 16;
    for (int blocks : {512, 1024, 2048, 4096}) {
        for (int shift : {0, 7}) {
            float a = time_ms([&] { k_copy<8, false><<<blocks, 256>>>(src, dst, nc, shift); });
            float b = time_ms([&] { k_copy<16, false><<<blocks, 256>>>(src, dst, nc, shift); });
            float c = time_ms([&] { k_copy<16, true><<<blocks, 256>>>(src, dst, nc, shift); });
            float d = time_ms([&] { k_copy<32, false><<<blocks, 256>>>(src, dst, nc, shift); });
            printf("copy blocks %4d shift %d: U8 %.0f  U16 %.0f  U16nt %.0f  U32 %.0f GB/s (read+write)\n", blocks, shift,
                   2.0 * n / a / 1e6, 2.0 * n / b / 1e6, 2.0 * n / c / 1e6, 2.0 * n / d / 1e6);
        }
        float r = time_ms([&] { k_read<16><<<blocks, 256>>>(src, out, nc); });
        printf("read blocks %4d: U16 %.0f GB/s\n", blocks, 1.0 * n / r / 1e6);
    }
    return 0;
}
