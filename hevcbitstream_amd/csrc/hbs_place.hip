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
 * turnaround, on two ranks they do not.  HIP does not tell physical addresses, so where a buffer lies relative to an input
 * is MEASURED: a content-free copy with K12's geometry from a piece of the input into the piece of the candidate that will
 * be written from it, against the same copy inside the candidate (one allocation = one class = the slow case by
 * construction).  Since round 5 what is learned is kept: a per-device pool of 1 GiB chunks, each classed once (below).
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
#include <new>
#include <vector>
#include "hevcbitstream_amd.h"

extern "C" __attribute__((visibility("hidden"))) void hbs_ctx_set_error(hbs_ctx* c, const char* what, int hip_error);

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr uint64_t kChunk = 1ull << 30;           /* physical chunks: the classes were seen at this grain, and a probe needs ~0.5 GiB to tell them apart */
constexpr int kRows = 48;                          /* K12's wavefront: 48 rows of 1 KiB in registers */
constexpr uint64_t kTile = 4ull * kRows * 1024;    /* 192 KiB per workgroup */
constexpr double kFastRatio = 0.985;               /* a pairing is "fast" when it takes less than this x the chunk against itself: the classes are 3-4 % apart
                                                      at half a GiB, repeated measurements 0.3 % (256 MiB: 2.5 % apart, 64 MiB: nothing) */

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

std::mutex g_mu;
std::map<void*, int> g_pairs;                      /* pointer -> device */

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


constexpr uint64_t kGran = 2ull << 20;
uint64_t up(uint64_t v, uint64_t a) { return (v + a - 1) / a * a; }

/* Addresses for the chunked buffers: ONE reservation per process, handed out front to back and never given back or used
 * twice.  On this stack a virtual address that had been unmapped and mapped again -- inside one call, or freed and reserved
 * again by a later one -- now and then kept serving the OLD physical memory (wrong bytes in the second buffer of a
 * process; hipMemSetAccess failing with "invalid argument").  An address that is used once cannot go stale.  When the pool
 * is used up (kVaPool of addresses: some 500 buffers of 16 GiB with their scratch slots) the plain allocator takes over. */
constexpr uint64_t kVaPool = 32ull << 40;
std::mutex g_va_mu;
uint8_t* g_va_base = nullptr;
uint64_t g_va_size = 0, g_va_used = 0;
bool g_va_tried = false;
void* va_take(uint64_t bytes)
{
    std::lock_guard<std::mutex> g(g_va_mu);
    if (!g_va_tried) {
        g_va_tried = true;
        for (uint64_t want = kVaPool; want >= (1ull << 40); want /= 2) {
            void* p = nullptr;
            const hipError_t e = hipMemAddressReserve(&p, want, kGran, nullptr, 0);
            if (getenv("HBS_PAIR_DEBUG")) fprintf(stderr, "hbs_pair_alloc: address pool of %llu GiB: %s\n", (unsigned long long)(want >> 30), hipGetErrorString(e));
            if (e == hipSuccess && p) { g_va_base = static_cast<uint8_t*>(p); g_va_size = want; break; }
            (void)hipGetLastError();
        }
    }
    bytes = up(bytes, kGran);
    if (!g_va_base || g_va_used + bytes > g_va_size) return nullptr;
    void* r = g_va_base + g_va_used;
    g_va_used += bytes;
    return r;
}

/* ---- the pool (round 5) -----------------------------------------------------------------------------------------
 * Round 4 measured on EVERY allocation: 31-77 probe copies of half a GiB, 0.04-5.7 s, up to 128 GiB of ballast held meanwhile,
 * and everything it had learned about the physical memory it touched was thrown away with the candidates it released.  What a
 * chunk's class is does not change while the chunk exists, so the knowledge is kept with the chunk:
 *   - per device ONE reference chunk R, created with the pool and kept mapped: the class of anything is "R's" (0) or "the
 *     other" (1), from one probe copy against R and one against itself;
 *   - every 1 GiB chunk the pool ever created is classed ONCE; a buffer is put together from chunks of the class its peer
 *     piece is not in; hbs_pair_free unmaps a buffer's chunks and puts them back on the pool's free list, class attached;
 *   - candidates of the class nobody wants right now are not released either: they are the next call's supply (they used to
 *     be "kept aside, then freed" -- and handed out again by the driver to be measured again);
 *   - a peer that is itself a buffer of this pool is classed by table lookup, no probe at all.
 * A second allocation of 16 GiB therefore costs the probes of its peer's 16 pieces (~1 ms each; none for a pool buffer) and
 * the mapping calls.  The free list is bounded (kPoolKeepDefault, HBS_PAIR_POOL_KEEP_GIB, hbs_pair_pool_trim): what is above
 * the bound when a call ends goes back to the driver, the majority class first.  Addresses are still used once (va_take). */
struct Chunk { hipMemGenericAllocationHandle_t h; int cls; };       /* cls: 0 = the reference chunk's class, 1 = the other */
struct Pair {
    void* va = nullptr;
    uint64_t va_bytes = 0;
    std::vector<Chunk> chunks;                                       /* chunk k is mapped at va + k GiB */
    int device = 0;
};
struct Pool {
    int device = 0;
    bool ready = false, failed = false;
    hipMemGenericAllocationHandle_t ref_h;
    uint8_t* ref_va = nullptr;                                       /* the reference chunk, mapped for as long as the process lives */
    double ref_self_ms = 0;                                          /* R against itself: the slow case */
    std::vector<Chunk> free_chunks;
    uint64_t created = 0, classified = 0;
};
constexpr uint64_t kPoolKeepDefault = 48ull << 30;
std::map<int, Pool*> g_pools;                       /* under g_mu */
std::map<void*, Pair*> g_chunked;                   /* under g_mu */

void release_pair_to_pool(Pair* p, Pool* pool)      /* the chunks go back to the free list (no pool: to the driver); the addresses are spent */
{
    for (size_t i = 0; i < p->chunks.size(); ++i) {
        (void)hipMemUnmap(static_cast<uint8_t*>(p->va) + i * kChunk, kChunk);
        if (pool) pool->free_chunks.push_back(p->chunks[i]);
        else (void)hipMemRelease(p->chunks[i].h);
    }
    delete p;
}

uint64_t pool_keep_bytes()
{
    static const uint64_t keep = [] {
        const char* e = getenv("HBS_PAIR_POOL_KEEP_GIB");
        return e && atoi(e) >= 0 ? (uint64_t)atoi(e) << 30 : kPoolKeepDefault;
    }();
    return keep;
}

/* free chunks above `keep` bytes back to the driver, the class there is more of first; returns the bytes released */
uint64_t pool_trim(Pool* pool, uint64_t keep)
{
    uint64_t released = 0;
    while ((uint64_t)pool->free_chunks.size() * kChunk > keep) {
        size_t n[2] = {0, 0};
        for (auto& c : pool->free_chunks) n[c.cls & 1] += 1;
        const int victim = n[0] >= n[1] ? 0 : 1;
        for (size_t i = pool->free_chunks.size(); i-- > 0;)
            if (pool->free_chunks[i].cls == victim) {
                (void)hipMemRelease(pool->free_chunks[i].h);
                pool->free_chunks.erase(pool->free_chunks.begin() + (long)i);
                released += kChunk;
                break;
            }
    }
    return released;
}

} // namespace

/*
 * The chunked way, from the pool.  Physical memory comes in runs of one class in the order hipMemCreate hands it out (runs of
 * 2-10 GiB seen, 35 once):
 *   1. every piece of the peer is classed against R (a pool buffer: by lookup);
 *   2. chunk k of the buffer wants the class its peer piece is NOT in: taken from the free list when there is one, else new
 *      chunks are created, each mapped into an address slot of its own and classed against R; those of the class nobody wants
 *      go to the free list (which also makes the driver hand out OTHER memory next), and after two of those in a row an unmapped
 *      "ballast" allocation of 4, 8, 16 ... GiB skips ahead in the run (released when the call ends);
 *   3. the chosen chunks are mapped at their place in the buffer.
 * Bounded: at most nchunks + kExtraCands new chunks and kBallastMax of ballast per call, kKeepFree of the device left free;
 * when memory runs out or the bound is reached, whatever is at hand is used (the report says how many chunks were placed
 * knowingly).
 */
static int alloc_chunked(hbs_ctx* ctx, const void* d_peer, uint64_t peer_bytes, uint64_t bytes, void** out, hbs_pair_report* rep)
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
    hipError_t e = hipSuccess;
    const bool dbg = getenv("HBS_PAIR_DEBUG") != nullptr;
#define PAIR_FAIL(what) { hbs_ctx_set_error(ctx, "hbs_pair_alloc: " what, (int)e); rc = HBS_E_HIP; }

    /* layout: whole chunks of 1 GiB, the size rounded up (a remainder mapped as a chunk of its own size made hipMemSetAccess
     * fail with "invalid argument" on this stack, so every mapping is one GiB: up to a GiB more than asked for is held) */
    const uint64_t total = up(bytes, kGran);
    const bool want_probe = d_peer != nullptr && peer_bytes >= kChunk / 2 && total >= kChunk && !getenv("HBS_PAIR_NO_PROBE");
    if (!want_probe) return HBS_E_CAPACITY;                           /* nothing to place: the plain way */
    if ((reinterpret_cast<uintptr_t>(d_peer) & 15u) != 0) return HBS_E_ARG;     /* the probe loads 16 bytes a lane */
    const uint64_t nfull = (total + kChunk - 1) / kChunk;
    const uint64_t half = (kChunk / 2) / kTile * kTile;

    std::lock_guard<std::mutex> pool_guard(g_mu);                     /* one allocation at a time per process: the pool, the address pool, the probes */
    Pool*& slot = g_pools[device];
    if (!slot) { slot = new (std::nothrow) Pool(); if (slot) slot->device = device; }
    Pool* const pool = slot;
    if (!pool || pool->failed) return HBS_E_CAPACITY;
    Prober pr(st, device);
    if (!pr.ok) return HBS_E_CAPACITY;
    auto map_new = [&](uint8_t* at, hipMemGenericAllocationHandle_t h) -> bool {
        hipError_t me = hipMemMap(at, kChunk, 0, h, 0);
        if (me == hipSuccess) { me = hipMemSetAccess(at, kChunk, &acc, 1); if (me != hipSuccess) (void)hipMemUnmap(at, kChunk); }
        if (me != hipSuccess) { if (dbg) fprintf(stderr, "hbs_pair_alloc: mapping at %p: %s\n", (void*)at, hipGetErrorString(me)); e = me; (void)hipGetLastError(); return false; }
        return true;
    };
    if (!pool->ready) {                                               /* the reference chunk: once per device */
        uint8_t* va = static_cast<uint8_t*>(va_take(kChunk));
        if (!va || hipMemCreate(&pool->ref_h, kChunk, &prop, 0) != hipSuccess) { (void)hipGetLastError(); pool->failed = true; return HBS_E_CAPACITY; }
        if (!map_new(va, pool->ref_h)) { (void)hipMemRelease(pool->ref_h); pool->failed = true; return HBS_E_CAPACITY; }
        pool->ref_va = va;
        pool->ref_self_ms = pr.copy_ms(va, va + half, half);
        if (pool->ref_self_ms <= 0) { pool->failed = true; return HBS_E_CAPACITY; }
        pool->ready = true; pool->created = 1;
        if (dbg) fprintf(stderr, "hbs_pair_alloc: pool of device %d: reference chunk against itself %.4f ms\n", device, pool->ref_self_ms);
    }

    /* 1. what every chunk of the buffer wants: the class its peer piece is not in */
    std::vector<int> want(nfull, -1);
    uint64_t need[2] = {0, 0};
    {
        /* a peer that is a buffer of this pool: its chunks' classes are on record */
        const Pair* peer_pair = nullptr;
        uint64_t peer_pair_off = 0;
        for (auto& kv : g_chunked) {
            const uint8_t* lo = static_cast<const uint8_t*>(kv.first);
            if (static_cast<const uint8_t*>(d_peer) >= lo && static_cast<const uint8_t*>(d_peer) < lo + kv.second->va_bytes && kv.second->device == device) {
                peer_pair = kv.second; peer_pair_off = (uint64_t)(static_cast<const uint8_t*>(d_peer) - lo); break;
            }
        }
        const uint64_t peer_last = peer_bytes > half ? (uint64_t)((peer_bytes - half) & ~(uint64_t)15) : (uint64_t)0;
        for (uint64_t k = 0; k < nfull; ++k) {
            const uint64_t peer_off = std::min(k * kChunk, peer_last);
            int peer_cls = -1;
            if (peer_pair && (peer_pair_off & (kChunk - 1)) == 0) {
                const uint64_t pk = (peer_pair_off + peer_off) / kChunk;
                if (pk < peer_pair->chunks.size()) { peer_cls = peer_pair->chunks[pk].cls; r.from_table += 1; }
            }
            if (peer_cls < 0) {
                const double t = pr.copy_ms(static_cast<const uint8_t*>(d_peer) + peer_off, pool->ref_va + half, half);
                if (t <= 0) break;
                peer_cls = t < kFastRatio * pool->ref_self_ms ? 1 : 0;          /* fast against R: the other class */
                r.probed += 1;
                if (dbg) fprintf(stderr, "hbs_pair_alloc: peer piece %llu against the reference chunk: %.4f / %.4f = %.4f -> class %d\n",
                                 (unsigned long long)k, t, pool->ref_self_ms, t / pool->ref_self_ms, peer_cls);
            }
            want[k] = 1 - peer_cls;
            need[want[k]] += 1;
        }
    }

    /* 2. supply: the free list first, new chunks for the rest */
    uint64_t have[2] = {0, 0};
    for (auto& c : pool->free_chunks) have[c.cls & 1] += 1;
    const uint64_t from_pool0 = std::min(have[0], need[0]), from_pool1 = std::min(have[1], need[1]);
    r.from_pool = (uint32_t)(from_pool0 + from_pool1);
    /* Round 6: a bounded search by default -- 24 candidates past the chunks and 32 GiB of ballast (88 and 128 GiB until then: the
     * first allocation of a process took 3.4-5.7 s on boxes that hand out long runs of one class, for a median gain of 1.3 % of K12's
     * time over six processes, profiles/r06/pair_time.txt).  HBS_PAIR_EXTRA_CANDS / HBS_PAIR_BALLAST_GIB widen it again. */
    static const int kExtraCands = [] { const char* e = getenv("HBS_PAIR_EXTRA_CANDS"); const int v = e ? atoi(e) : 0; return v > 0 && v <= 512 ? v : 24; }();
    static const uint64_t kBallastMax = [] { const char* e = getenv("HBS_PAIR_BALLAST_GIB"); const int v = e ? atoi(e) : -1; return (uint64_t)(v >= 0 && v <= 256 ? v : 32) << 30; }();
    constexpr uint64_t kKeepFree = 24ull << 30;
    auto room_for = [&](uint64_t more) -> bool {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return false; }
        return (uint64_t)fr >= more + kKeepFree;
    };
    std::vector<hipMemGenericAllocationHandle_t> ballast;
    uint64_t ballast_bytes = 0, next_ballast = 4ull << 30, made = 0;
    int unwanted_in_a_row = 0;
    while ((have[0] < need[0] || have[1] < need[1]) && made < nfull + (uint64_t)kExtraCands) {
        if (unwanted_in_a_row >= 2 && ballast_bytes + next_ballast <= kBallastMax && room_for(next_ballast + kChunk)) {
            hipMemGenericAllocationHandle_t b;
            if (hipMemCreate(&b, next_ballast, &prop, 0) == hipSuccess) {
                ballast.push_back(b); ballast_bytes += next_ballast; next_ballast *= 2; unwanted_in_a_row = 0;
                if (dbg) fprintf(stderr, "hbs_pair_alloc: ballast %llu GiB\n", (unsigned long long)(ballast_bytes >> 30));
            } else {
                (void)hipGetLastError();
                next_ballast /= 2;
                if (next_ballast < (1ull << 30)) break;
                continue;
            }
        }
        if (!room_for(kChunk)) break;
        Chunk x; x.cls = -1;
        if (hipMemCreate(&x.h, kChunk, &prop, 0) != hipSuccess) { (void)hipGetLastError(); break; }
        uint8_t* xva = static_cast<uint8_t*>(va_take(kChunk));         /* a slot of its own, used once */
        if (!xva || !map_new(xva, x.h)) { (void)hipMemRelease(x.h); break; }
        const double self_x = pr.copy_ms(xva, xva + half, half);
        const double t = pr.copy_ms(pool->ref_va, xva + half, half);
        (void)hipStreamSynchronize(st);
        (void)hipMemUnmap(xva, kChunk);
        if (self_x <= 0 || t <= 0) { (void)hipMemRelease(x.h); break; }
        x.cls = t < kFastRatio * self_x ? 1 : 0;
        made += 1; pool->created += 1; pool->classified += 1;
        r.probed += 1;
        const bool wanted = have[x.cls] < need[x.cls];
        have[x.cls] += 1;
        unwanted_in_a_row = wanted ? 0 : unwanted_in_a_row + 1;
        if (!wanted) r.rejected += 1;
        pool->free_chunks.push_back(x);
        if (dbg) fprintf(stderr, "hbs_pair_alloc: new chunk %llu against the reference chunk: %.4f / %.4f = %.4f -> class %d%s\n",
                         (unsigned long long)pool->created, t, self_x, t / self_x, x.cls, wanted ? "" : " (to the free list)");
    }
    for (auto b : ballast) (void)hipMemRelease(b);

    /* 3. the buffer: chunk k from the free list, of its class when there is one, any other when not (a new one when the list is empty) */
    Pair* p = new (std::nothrow) Pair();
    if (!p) return HBS_E_HIP;
    p->device = device;
    p->va_bytes = nfull * kChunk;
    p->va = va_take(p->va_bytes);
    if (!p->va) { delete p; return HBS_E_CAPACITY; }                  /* the address pool is used up: the caller takes the plain way */
    uint8_t* const base = static_cast<uint8_t*>(p->va);
    int rc = 0;
    std::vector<int> chosen(nfull, 0);
    auto take = [&](int cls) -> long {
        for (size_t i = 0; i < pool->free_chunks.size(); ++i) if (cls < 0 || pool->free_chunks[i].cls == cls) return (long)i;
        return -1;
    };
    for (int pass = 0; pass < 2 && rc == 0; ++pass)
        for (uint64_t k = 0; k < nfull && rc == 0; ++k) {
            if (chosen[k]) continue;
            long i = pass == 0 ? (want[k] >= 0 ? take(want[k]) : -1) : take(-1);
            Chunk c;
            if (i >= 0) { c = pool->free_chunks[(size_t)i]; pool->free_chunks.erase(pool->free_chunks.begin() + i); }
            else if (pass == 0) continue;
            else {
                if (hipMemCreate(&c.h, kChunk, &prop, 0) != hipSuccess) { e = hipErrorOutOfMemory; (void)hipGetLastError(); PAIR_FAIL("hipMemCreate") break; }
                c.cls = -1; pool->created += 1;
            }
            (void)pass;
            chosen[k] = 1;
            if (c.cls >= 0 && c.cls == want[k]) r.accepted_fast += 1; else r.unprobed_after_budget += 1;
            p->chunks.resize(std::max<size_t>(p->chunks.size(), (size_t)k + 1));
            p->chunks[(size_t)k] = c;
        }
    /* (chunks were assigned in two passes: map them in order now) */
    size_t mapped = 0;
    for (uint64_t k = 0; k < nfull && rc == 0; ++k) {
        if (!map_new(base + k * kChunk, p->chunks[(size_t)k].h)) { PAIR_FAIL("hipMemMap") break; }
        mapped += 1;
    }
#undef PAIR_FAIL
    if (rc) {
        for (size_t k = 0; k < p->chunks.size(); ++k) {
            if (k < mapped) (void)hipMemUnmap(base + k * kChunk, kChunk);
            if (chosen[k]) { if (p->chunks[k].cls >= 0) pool->free_chunks.push_back(p->chunks[k]); else (void)hipMemRelease(p->chunks[k].h); }
        }
        delete p;
        (void)pool_trim(pool, pool_keep_bytes());
        return rc;
    }
    (void)pool_trim(pool, pool_keep_bytes());
    r.chunks = (uint32_t)p->chunks.size();
    g_chunked[p->va] = p;
    *out = p->va;
    if (rep) *rep = r;
    if (dbg) fprintf(stderr, "hbs_pair_alloc: %u chunks (%u placed knowingly, %u from the free list), %u probes, %llu chunks left on the free list\n",
                     r.chunks, r.accepted_fast, r.from_pool, r.probed, (unsigned long long)pool->free_chunks.size());
    return 0;
}


/*
 * The plain way.  Ordinary hipMalloc memory -- one allocation of the whole size.  Up to kTries allocations are made, each KEPT
 * while the next is made (so that the next one is other memory); every whole GiB of a candidate is measured against the
 * peer piece it will be written from (and against itself: the slow case by construction); the first candidate whose pieces
 * are all fast wins, else the one with the most fast pieces; the others are freed.  Three pieces (first, middle, last) are
 * measured first, and a candidate that is slow in all three is not measured further.  An allocation of many GiB usually
 * spans both classes somewhere (13-15 fast pieces of 15 were typical), which is why the chunked way comes first.
 */
static int alloc_plain(hbs_ctx* ctx, const void* d_peer, uint64_t peer_bytes, uint64_t bytes, void** out, hbs_pair_report* rep)
{
    if (!ctx || !out) return HBS_E_ARG;
    *out = nullptr;
    hbs_pair_report r;
    memset(&r, 0, sizeof(r));
    const int device = hbs_ctx_device(ctx);
    if (device < 0 || hipSetDevice(device) != hipSuccess) return HBS_E_NO_DEVICE;
    hipStream_t st = reinterpret_cast<hipStream_t>(hbs_ctx_get_stream(ctx));
    if (bytes == 0) bytes = 16;
    constexpr int kTries = 6;
    const uint64_t half = (kChunk / 2) / kTile * kTile;
    const uint64_t nfull = bytes / kChunk;
    const bool want_probe = d_peer != nullptr && peer_bytes >= half && nfull >= 1 && !getenv("HBS_PAIR_NO_PROBE");
    const bool dbg = getenv("HBS_PAIR_DEBUG") != nullptr;
    Prober* pr = want_probe ? new (std::nothrow) Prober(st, device) : nullptr;
    if (pr && !pr->ok) { delete pr; pr = nullptr; }
    r.chunks = (uint32_t)((bytes + kChunk - 1) / kChunk);

    struct Cand { uint8_t* p; uint32_t fast, measured; };
    std::vector<Cand> cands;
    int best = -1;
    hipError_t e = hipSuccess;
    for (int attempt = 0; attempt < (pr ? kTries : 1); ++attempt) {
        if (attempt) {                                    /* room for one more, with some to spare? */
            size_t fr = 0, tot = 0;
            if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr < bytes + (4ull << 30)) break;
        }
        Cand c; c.p = nullptr; c.fast = 0; c.measured = 0;
        e = hipMalloc(reinterpret_cast<void**>(&c.p), bytes);
        if (e != hipSuccess) { (void)hipGetLastError(); break; }
        cands.push_back(c);
        Cand& C = cands.back();
        if (!pr) { best = 0; break; }
        auto piece_is_fast = [&](uint64_t k) -> int {          /* 1 fast, 0 slow, -1 measurement failed */
            const uint64_t peer_last = peer_bytes > half ? (uint64_t)((peer_bytes - half) & ~(uint64_t)15) : (uint64_t)0;
            const uint64_t peer_off = std::min(k * kChunk, peer_last);
            uint8_t* lo = C.p + k * kChunk;
            const double t_self = pr->copy_ms(lo, lo + half, half);
            const double t_pair = pr->copy_ms(static_cast<const uint8_t*>(d_peer) + peer_off, lo + half, half);
            if (t_self <= 0 || t_pair <= 0) return -1;
            r.probed += 1;
            if (dbg) fprintf(stderr, "hbs_pair_alloc: candidate %d piece %llu: against the peer %.4f ms, against itself %.4f ms (%.4f)\n",
                             attempt, (unsigned long long)k, t_pair, t_self, t_pair / t_self);
            return t_pair < kFastRatio * t_self ? 1 : 0;
        };
        /* first, middle, last; then the rest unless those three are all slow */
        std::vector<uint64_t> order;
        order.push_back(0);
        if (nfull > 2) order.push_back(nfull / 2);
        if (nfull > 1) order.push_back(nfull - 1);
        const size_t quick = order.size();
        for (uint64_t k = 0; k < nfull; ++k) if (k != 0 && k != nfull / 2 && k != nfull - 1) order.push_back(k);
        bool failed = false;
        for (size_t q = 0; q < order.size(); ++q) {
            if (q == quick && C.fast == 0) break;                /* slow wherever it was looked at */
            const int f = piece_is_fast(order[q]);
            if (f < 0) { failed = true; break; }
            C.measured += 1; C.fast += (uint32_t)f;
        }
        if (failed) { best = best < 0 ? (int)cands.size() - 1 : best; break; }
        if (best < 0 || C.fast > cands[(size_t)best].fast) best = (int)cands.size() - 1;
        if (C.fast == nfull) break;                              /* every piece pairs fast: done */
        r.rejected += 1;
    }
    delete pr;
    if (best < 0) {
        hbs_ctx_set_error(ctx, "hbs_pair_alloc: hipMalloc", (int)(e == hipSuccess ? hipErrorOutOfMemory : e));
        for (auto& c : cands) (void)hipFree(c.p);
        return HBS_E_HIP;
    }
    (void)hipStreamSynchronize(st);
    for (size_t i = 0; i < cands.size(); ++i) if ((int)i != best) (void)hipFree(cands[i].p);
    if (r.rejected && cands[(size_t)best].fast == nfull) r.rejected = (uint32_t)cands.size() - 1;
    r.accepted_fast = cands[(size_t)best].fast;
    r.unprobed_after_budget = r.chunks - cands[(size_t)best].fast;
    {
        std::lock_guard<std::mutex> g(g_mu);
        g_pairs[cands[(size_t)best].p] = device;
    }
    *out = cands[(size_t)best].p;
    if (rep) *rep = r;
    return 0;
}

extern "C" {

/* Two ways.  (1) Chunked: the buffer is put together from 1 GiB physical chunks of the device's pool, each classed by measurement
 * ONCE, at addresses that are used once (see va_take) -- every piece can be placed, whatever the allocator hands out.  (2) Plain: whole
 * hipMalloc allocations as candidates, the best one kept -- ordinary memory, but an allocation usually spans both classes
 * somewhere.  (1) is used for buffers that get probed at all; (2) for small buffers (one plain allocation, no probing),
 * when HBS_PAIR_PLAIN is set, and when the address pool or the virtual-memory calls give out. */
int hbs_pair_alloc(hbs_ctx* ctx, const void* d_peer, uint64_t peer_bytes, uint64_t bytes, void** out, hbs_pair_report* rep)
{
    if (!ctx || !out) return HBS_E_ARG;
    const bool probe = d_peer != nullptr && peer_bytes >= kChunk / 2 && bytes >= kChunk && !getenv("HBS_PAIR_NO_PROBE");
    if (probe && !getenv("HBS_PAIR_PLAIN")) {
        const int rc = alloc_chunked(ctx, d_peer, peer_bytes, bytes, out, rep);
        (void)hipGetLastError();                       /* a HIP call that failed on the way (memory ran out under the candidates) is not the caller's error */
        if (rc == 0) return 0;
        if (getenv("HBS_PAIR_DEBUG")) fprintf(stderr, "hbs_pair_alloc: the chunked way failed (%d: %s), taking the plain one\n", rc, hbs_last_error(ctx));
        if (rc != HBS_E_CAPACITY && rc != HBS_E_HIP) return rc;
    }
    const int rc = alloc_plain(ctx, d_peer, peer_bytes, bytes, out, rep);
    (void)hipGetLastError();
    return rc;
}

int hbs_pair_free(hbs_ctx* ctx, void* ptr)
{
    if (!ptr) return 0;
    int device = 0;
    Pair* p = nullptr;
    {
        std::lock_guard<std::mutex> g(g_mu);
        auto ic = g_chunked.find(ptr);
        if (ic != g_chunked.end()) { p = ic->second; g_chunked.erase(ic); }
        else {
            auto it = g_pairs.find(ptr);
            if (it == g_pairs.end()) return HBS_E_ARG;
            device = it->second;
            g_pairs.erase(it);
        }
    }
    if (p) device = p->device;
    /* This runs wherever the last reference to a buffer dies (a destructor, a garbage collector, any thread): the caller's current
     * device is put back afterwards, and only what can still touch the buffer is waited for -- the context's stream when there is
     * a context, the device otherwise (round 4 switched the device under the caller and always stalled every stream). */
    int prev = -1;
    const bool switched = hipGetDevice(&prev) == hipSuccess && prev != device;
    if (switched) (void)hipSetDevice(device);
    if (ctx) (void)hbs_ctx_synchronize(ctx);
    else (void)hipDeviceSynchronize();
    int rc = 0;
    if (p) {
        std::lock_guard<std::mutex> g(g_mu);
        auto ip = g_pools.find(device);
        Pool* pool = ip != g_pools.end() ? ip->second : nullptr;
        release_pair_to_pool(p, pool);
        if (pool) (void)pool_trim(pool, pool_keep_bytes());
    } else {
        rc = hipFree(ptr) == hipSuccess ? 0 : HBS_E_HIP;
    }
    if (switched) (void)hipSetDevice(prev);
    return rc;
}

/* free chunks of the context's device above `keep_bytes` go back to the driver (0: all of them); returns the bytes released */
uint64_t hbs_pair_pool_trim(hbs_ctx* ctx, uint64_t keep_bytes)
{
    if (!ctx) return 0;
    const int device = hbs_ctx_device(ctx);
    std::lock_guard<std::mutex> g(g_mu);
    auto ip = g_pools.find(device);
    if (ip == g_pools.end() || !ip->second) return 0;
    return pool_trim(ip->second, keep_bytes);
}

/* out[0] chunks the pool created so far (the reference chunk included), [1] chunks classed by measurement, [2] free chunks of the
 * reference's class, [3] free chunks of the other class */
int hbs_pair_pool_stats(hbs_ctx* ctx, uint64_t out[4])
{
    if (!ctx || !out) return HBS_E_ARG;
    const int device = hbs_ctx_device(ctx);
    out[0] = out[1] = out[2] = out[3] = 0;
    std::lock_guard<std::mutex> g(g_mu);
    auto ip = g_pools.find(device);
    if (ip == g_pools.end() || !ip->second) return 0;
    out[0] = ip->second->created; out[1] = ip->second->classified;
    for (auto& c : ip->second->free_chunks) out[2 + (c.cls & 1)] += 1;
    return 0;
}

} // extern "C"
