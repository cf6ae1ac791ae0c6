// Microbenchmark: bandwidth of 16-byte-per-lane global stores whose destination is
// misaligned by `shift` bytes, vs aligned.  (dev aid)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(1))) U16 { u32x4 v; };

__global__ void k_copy(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t nchunks, int shift)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < nchunks; i += stride) {
        u32x4 v = *reinterpret_cast<const u32x4*>(src + 16 * i);
        reinterpret_cast<U16*>(dst + shift + 16 * i)->v = v;
    }
}
__global__ void k_copy_ld(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t nchunks, int shift)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < nchunks; i += stride) {
        u32x4 v = reinterpret_cast<const U16*>(src + shift + 16 * i)->v;
        *reinterpret_cast<u32x4*>(dst + 16 * i) = v;
    }
}
int main()
{
    const size_t n = 1ull << 30;
    uint8_t *src, *dst;
    hipMalloc(&src, n + 64); hipMalloc(&dst, n + 64);
    hipMemset(src, 1, n + 64); hipMemset(dst, 0, n + 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
    for (int shift : {0, 1, 2, 4, 7, 8, 13}) {
        for (int it = 0; it < 2; ++it) {
            if (mode == 0) k_copy<<<256 * 8, 256>>>(src, dst, n / 16, shift); else k_copy_ld<<<256 * 8, 256>>>(src, dst, n / 16, shift);
        }
        hipEventRecord(e0);
        for (int it = 0; it < 5; ++it) {
            if (mode == 0) k_copy<<<256 * 8, 256>>>(src, dst, n / 16, shift); else k_copy_ld<<<256 * 8, 256>>>(src, dst, n / 16, shift);
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%s shift %2d: %.3f ms  %.1f GB/s (read+write)\n", mode == 0 ? "unaligned STORE" : "unaligned LOAD ", shift, ms, 2.0 * n / ms / 1e6);
    }
    // verify shift=7 store correctness
    k_copy<<<256 * 8, 256>>>(src, dst, n / 16, 7);
    std::vector<uint8_t> h(64); hipMemcpy(h.data(), dst, 64, hipMemcpyDeviceToHost);
    printf("dst[0..15] after shift 7:"); for (int i = 0; i < 16; ++i) printf(" %d", h[i]); printf("\n");
    return 0;
}
