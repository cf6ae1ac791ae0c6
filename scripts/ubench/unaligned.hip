// How much do byte-misaligned 16-byte loads / stores cost on MI355X?  (dev aid)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/unaligned scripts/ubench/unaligned.hip && /tmp/unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(1))) U16 { u32x4 v; };

template <int U>
__global__ void k_copy(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t nchunks, int sshift, int dshift)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const size_t span = (size_t)U * 64;
    size_t s = ((size_t)blockIdx.x * wpb + wave) * span;
    const size_t stride = (size_t)gridDim.x * wpb * span;
    for (; s + span <= nchunks; s += stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = reinterpret_cast<const U16*>(src + sshift + 16 * (s + (size_t)u * 64 + lane))->v;
#pragma unroll
        for (int u = 0; u < U; ++u) reinterpret_cast<U16*>(dst + dshift + 16 * (s + (size_t)u * 64 + lane))->v = v[u];
    }
}
template <int U>
__global__ void k_read(const uint8_t* __restrict__ src, uint32_t* __restrict__ out, size_t nchunks, int sshift)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const size_t span = (size_t)U * 64;
    size_t s = ((size_t)blockIdx.x * wpb + wave) * span;
    const size_t stride = (size_t)gridDim.x * wpb * span;
    uint32_t acc = 0;
    for (; s + span <= nchunks; s += stride) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = reinterpret_cast<const U16*>(src + sshift + 16 * (s + (size_t)u * 64 + lane))->v;
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}
template <int U>
__global__ void k_write(uint8_t* __restrict__ dst, size_t nchunks, int dshift)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const size_t span = (size_t)U * 64;
    size_t s = ((size_t)blockIdx.x * wpb + wave) * span;
    const size_t stride = (size_t)gridDim.x * wpb * span;
    u32x4 v; v.x = lane; v.y = 2; v.z = 3; v.w = 4;
    for (; s + span <= nchunks; s += stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) reinterpret_cast<U16*>(dst + dshift + 16 * (s + (size_t)u * 64 + lane))->v = v;
    }
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
    const int blocks = 512;
    for (int sh : {0, 4, 7, 8, 13}) {
        float r = time_ms([&] { k_read<32><<<blocks, 256>>>(src, out, nc, sh); });
        float w = time_ms([&] { k_write<32><<<blocks, 256>>>(dst, nc, sh); });
        float c0 = time_ms([&] { k_copy<32><<<blocks, 256>>>(src, dst, nc, sh, 0); });
        float c1 = time_ms([&] { k_copy<32><<<blocks, 256>>>(src, dst, nc, 0, sh); });
        float c2 = time_ms([&] { k_copy<32><<<blocks, 256>>>(src, dst, nc, sh, (sh * 5) & 15); });
        printf("shift %2d: read %.0f  write %.0f  copy(src shifted) %.0f  copy(dst shifted) %.0f  copy(both) %.0f GB/s\n", sh,
               n / r / 1e6, n / w / 1e6, 2.0 * n / c0 / 1e6, 2.0 * n / c1 / 1e6, 2.0 * n / c2 / 1e6);
    }
    return 0;
}
