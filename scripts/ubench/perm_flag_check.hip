// Dev check (round 3): chunk_pattern_any_dev (v_perm / v_min3 form, hbs_wave.h) against chunk_pattern_any (hbs_chunk.h) on
// random chunks dense in 00..03 bytes.   hipcc -O3 --offload-arch=gfx950 -Iinclude -Ihevcbitstream_amd/csrc -o build/ubench/perm_flag_check scripts/ubench/perm_flag_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "hbs_wave.h"
#include "hbs_chunk.h"
using namespace hbs;
__global__ void k(const uint32_t* in, int n, unsigned* bad, unsigned* hits)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* q = in + 6 * (size_t)i;
    const bool a = chunk_pattern_any(q[0], q[1], q[2], q[3], q[4], q[5]);
    const bool b = chunk_pattern_any_dev(q[0], q[1], q[2], q[3], q[4], q[5]);
    if (a != b) atomicAdd(bad, 1u);
    if (a) atomicAdd(hits, 1u);
}
int main()
{
    const int n = 1 << 22;
    std::vector<uint8_t> h((size_t)n * 24);
    srand(7);
    for (size_t i = 0; i < h.size(); ++i) {
        const int r = rand() & (i < h.size() / 2 ? 15 : 63);        /* first half: 5 in 16 bytes zero; second: 5 in 64 */
        h[i] = r < 5 ? 0 : r < 9 ? (uint8_t)(r - 4) : (uint8_t)rand();
    }
    uint32_t* d; unsigned* c;
    hipMalloc(&d, h.size()); hipMalloc(&c, 8); hipMemset(c, 0, 8);
    hipMemcpy(d, h.data(), h.size(), hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(d, n, c, c + 1);
    unsigned out[2]; hipMemcpy(out, c, 8, hipMemcpyDeviceToHost);
    printf("chunks %d, with a pattern %u, disagreements %u\n", n, out[1], out[0]);
    return out[0] != 0;
}
