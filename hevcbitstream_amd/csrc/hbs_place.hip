/*
 * hbs_place.hip -- output buffers placed against the input they will be written from (hbs_pair_alloc / hbs_pair_free).
 *
 * Why this exists (round 4, scripts/ubench/placement.cpp, profiles/r04/placement_*.txt): on MI355X a kernel that reads one
 * buffer and writes another in long bursts -- K12 reads a stream and writes its RBSP arena, K3 the other way round -- runs in
 * one of two modes, decided by WHICH PHYSICAL MEMORY the two buffers got: 16 GiB in 5.90 ms or in 6.20 ms (0.729 or 0.692 of
 * the HBM peak), the same in every 1 GiB piece of the pair, for any offset inside either allocation, and unchanged by
 * anything the kernel does.  Physical memory falls into two classes (20 chunks of 1 GiB: every pair inside a class is
 * slow, every pair across classes fast, no exception in 2 x 380 pairs); pure reads and pure writes do not care, a
 * one-chunk-per-thread copy cares by 1.7 %, a copy with K12's geometry (48 KiB per wavefront loaded, then stored) by 4 %.
 * It looks like the two ranks behind every HBM channel: reads and writes that alternate on ONE rank pay its write-to-read
 * turnaround, on two ranks they do not.  HIP does not tell physical addresses, so the class of a chunk relative to an input
 * is MEASURED: a content-free copy with K12's geometry from the input piece into the chunk, against the same copy inside
 * the chunk (same chunk = same class = the slow case by construction).  A chunk whose pairing with its input piece is not
 * faster than its pairing with itself is put aside and another one is asked for; the pile is given back at the end.
 *
 * No reference counterpart (the reference allocates with malloc, h264_nal.c / hevc_nal.c); buffers from hipMalloc or
 * torch work with every call exactly as before -- they just land in the slow mode about every other time.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <vector>
#include "hevcbitstream_amd.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr uint64_t kChunk = 1ull << 30;           /* physical chunks: the classes were seen at this grain, and a probe needs ~0.5 GiB to tell them apart */
constexpr uint64_t kGran = 2ull << 20;
constexpr int kRows = 48;                          /* K12's wavefront: 48 rows of 1 KiB in registers */
constexpr uint64_t kTile = 4ull * kRows * 1024;    /* 192 KiB per workgroup */
constexpr uint64_t kProbeMin = 384ull << 20;       /* below this a probe does not separate the classes (256 MiB: 2.5 %, 64 MiB: nothing) */

/* K12's memory side without its logic: a persistent workgroup takes 192 KiB tiles by ticket, every wavefront loads its 48 rows,
 * then stores them.  Content-free, so the same bytes need not be on both sides of a comparison. */
__global__ __launch_bounds__(256, 2)
void k_probe_copy(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint64_t ntiles, unsigned* __restrict__ ticket)
{
    __shared__ unsigned tk;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (;;) {
        if (threadIdx.x == 0) tk = atomicAdd(ticket, 1u);
        __syncthreads();
        const uint64_t t = tk;
        __syncthreads();
        if (t >= ntiles) break;
        const uint64_t off = (t * 4 + (uint64_t)wv) * (uint64_t)(kRows * 1024) + 16u * (uint64_t)lane;
        u32x4 r[kRows];
#pragma unroll
        for (int i = 0; i < kRows; ++i) r[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src + off + 1024 * i));
#pragma unroll
        for (int i = 0; i < kRows; ++i) __builtin_nontemporal_store(r[i], reinterpret_cast<u32x4*>(dst + off + 1024 * i));
    }
}

struct Pair {
    void* va = nullptr;
    uint64_t va_bytes = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    std::vector<uint64_t> offs, sizes;
    int device = 0;
};

std::mutex g_mu;
std::map<void*, Pair*> g_pairs;

uint64_t up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

struct Prober {
    hipStream_t st;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    unsigned* ticket = nullptr;
    int grid = 512;
    bool ok = false;
    explicit Prober(hipStream_t s, int device) : st(s)
    {
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
        grid = 2 * cus;
        ok = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess && hipMalloc(reinterpret_cast<void**>(&ticket), 256) == hipSuccess;
    }
    ~Prober()
    {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (ticket) (void)hipFree(ticket);
    }
    /* best of three (after one warm-up) of a probe copy of `bytes` (a multiple of the tile); < 0 on a HIP error */
    double copy_ms(const uint8_t* src, uint8_t* dst, uint64_t bytes)
    {
        double best = -1.0;
        for (int r = 0; r < 4; ++r) {
            if (hipMemsetAsync(ticket, 0, 4, st) != hipSuccess) return -1.0;
            if (hipEventRecord(e0, st) != hipSuccess) return -1.0;
            k_probe_copy<<<dim3((unsigned)grid), dim3(256), 0, st>>>(src, dst, bytes / kTile, ticket);
            if (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) return -1.0;
            float ms = 0;
            if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return -1.0;
            if (r && (best < 0 || ms < best)) best = ms;
        }
        return best;
    }
};

void release_pair(Pair* p)
{
    for (size_t i = 0; i < p->handles.size(); ++i) {
        (void)hipMemUnmap(static_cast<uint8_t*>(p->va) + p->offs[i], p->sizes[i]);
        (void)hipMemRelease(p->handles[i]);
    }
    if (p->va) (void)hipMemAddressFree(p->va, p->va_bytes);
    delete p;
}

} // namespace

extern "C" {

int hbs_pair_alloc(hbs_ctx* ctx, const void* d_peer, uint64_t peer_bytes, uint64_t bytes, void** out, hbs_pair_report* rep)
{
    if (!ctx || !out) return HBS_E_ARG;
    *out = nullptr;
    hbs_pair_report r;
    memset(&r, 0, sizeof(r));
    const int device = hbs_ctx_device(ctx);
    if (device < 0 || hipSetDevice(device) != hipSuccess) return HBS_E_NO_DEVICE;
    hipStream_t st = reinterpret_cast<hipStream_t>(hbs_ctx_get_stream(ctx));
    if (bytes == 0) bytes = 16;
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    hipMemAccessDesc acc;
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;

    Pair* p = new (std::nothrow) Pair();
    if (!p) return HBS_E_HIP;
    p->device = device;
    p->va_bytes = up(bytes, kGran);
    if (hipMemAddressReserve(&p->va, p->va_bytes, kGran, nullptr, 0) != hipSuccess) { delete p; return HBS_E_HIP; }
    uint8_t* const base = static_cast<uint8_t*>(p->va);

    /* probing needs the caller's input to be there: everything enqueued on the context's stream so far */
    const bool want_probe = d_peer != nullptr && peer_bytes >= kProbeMin && bytes >= kProbeMin && !getenv("HBS_PAIR_NO_PROBE");
    Prober* pr = want_probe ? new (std::nothrow) Prober(st, device) : nullptr;
    if (pr && !pr->ok) { delete pr; pr = nullptr; }
    std::vector<hipMemGenericAllocationHandle_t> pile;              /* chunks put aside: kept until the end, so that the next one is other memory */
    std::vector<uint64_t> pile_sizes;
    const uint64_t nchunks = (p->va_bytes + kChunk - 1) / kChunk;
    int budget = (int)nchunks + 16;                                  /* chunks that may be put aside in all */
    double ratio_sum = 0.0;
    int rc = 0;
    for (uint64_t k = 0; k < nchunks && rc == 0; ++k) {
        const uint64_t off = k * kChunk;
        const uint64_t size = std::min(kChunk, p->va_bytes - off);
        /* the input piece this chunk will be written from (K12: arena offset ~ stream offset; K3: the same the other way) */
        const uint64_t half = (size / 2) / kTile * kTile;
        const uint64_t peer_last = peer_bytes > half ? (uint64_t)((peer_bytes - half) & ~(uint64_t)15) : (uint64_t)0;
        const uint64_t peer_off = std::min(off, peer_last);
        const bool probe = pr != nullptr && half >= kProbeMin / 2 && peer_bytes >= half;
        for (;;) {
            hipMemGenericAllocationHandle_t h;
            if (hipMemCreate(&h, size, &prop, 0) != hipSuccess) {
                /* out of memory while chunks are on the pile: give one back and take what comes, unprobed */
                if (pile.empty()) { rc = HBS_E_HIP; break; }
                (void)hipMemRelease(pile.back()); pile.pop_back(); pile_sizes.pop_back();
                budget = 0;
                continue;
            }
            if (hipMemMap(base + off, size, 0, h, 0) != hipSuccess) { (void)hipMemRelease(h); rc = HBS_E_HIP; break; }
            if (hipMemSetAccess(base + off, size, &acc, 1) != hipSuccess) { (void)hipMemUnmap(base + off, size); (void)hipMemRelease(h); rc = HBS_E_HIP; break; }
            bool keep = true;
            if (probe && budget > 0) {
                /* the chunk against itself (one class by construction: the slow case), then the input piece against the chunk */
                const double t_self = pr->copy_ms(base + off, base + off + half, half);
                const double t_pair = pr->copy_ms(static_cast<const uint8_t*>(d_peer) + peer_off, base + off + half, half);
                if (t_self > 0 && t_pair > 0) {
                    r.probed += 1;
                    keep = t_pair < 0.985 * t_self;              /* the classes are ~3.5 % apart at this size, repeat measurements 0.3 % */
                    if (keep) ratio_sum += t_pair / t_self;
                }
            }
            if (keep) {
                p->handles.push_back(h); p->offs.push_back(off); p->sizes.push_back(size);
                if (probe && budget <= 0) r.unprobed_after_budget += 1;
                break;
            }
            (void)hipMemUnmap(base + off, size);
            pile.push_back(h); pile_sizes.push_back(size);
            r.rejected += 1;
            budget -= 1;
        }
    }
    for (auto h : pile) (void)hipMemRelease(h);
    delete pr;
    if (rc) { release_pair(p); return rc; }
    r.chunks = (uint32_t)p->handles.size();
    r.accepted_fast = (uint32_t)(r.probed - r.rejected);
    r.mean_ratio = r.accepted_fast ? (float)(ratio_sum / r.accepted_fast) : 0.f;
    {
        std::lock_guard<std::mutex> g(g_mu);
        g_pairs[p->va] = p;
    }
    *out = p->va;
    if (rep) *rep = r;
    return 0;
}

int hbs_pair_free(hbs_ctx* ctx, void* ptr)
{
    if (!ptr) return 0;
    Pair* p = nullptr;
    {
        std::lock_guard<std::mutex> g(g_mu);
        auto it = g_pairs.find(ptr);
        if (it == g_pairs.end()) return HBS_E_ARG;
        p = it->second;
        g_pairs.erase(it);
    }
    (void)hipSetDevice(p->device);
    if (ctx) (void)hbs_ctx_synchronize(ctx);
    (void)hipDeviceSynchronize();
    release_pair(p);
    return 0;
}

} // extern "C"
