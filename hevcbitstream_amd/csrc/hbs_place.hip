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
 * construction).  A candidate that pairs slowly is kept aside while another one is allocated, then freed.
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

struct Pair {
    void* va = nullptr;
    uint64_t va_bytes = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    std::vector<uint64_t> offs, sizes;
    int device = 0;
};
std::map<void*, Pair*> g_chunked;                  /* under g_mu */

void release_pair(Pair* p)                          /* the physical memory goes back; the addresses are spent */
{
    for (size_t i = 0; i < p->handles.size(); ++i) {
        (void)hipMemUnmap(static_cast<uint8_t*>(p->va) + p->offs[i], p->sizes[i]);
        (void)hipMemRelease(p->handles[i]);
    }
    delete p;
}

} // namespace

/*
 * The chunked way.  Physical memory comes in runs of one class in the order hipMemCreate hands it out (runs of 2-10 GiB seen):
 *   1. a reference chunk R; every piece of the peer is classed against it (same class as R / the other class);
 *   2. candidate chunks are created, each mapped into an address slot of its own, and classed against R the same way;
 *   3. a chunk of the buffer wants a candidate of the class its peer piece is NOT in; candidates nobody wants stay allocated
 *      until the end (so that the next ones are other memory), and after two of those in a row an unmapped "ballast"
 *      allocation of 4, 8, 16 ... GiB skips ahead in the run;
 *   4. the chosen candidates are mapped at their place in the buffer; everything else is released.
 * Bounded: at most nchunks + kExtraCands candidates and kBallastMax of ballast, 24 GiB of the device left free; when memory runs out or the bound is reached,
 * whatever is at hand is used (the report says how many chunks were placed knowingly).
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
    hipError_t e;
#define PAIR_FAIL(what) { hbs_ctx_set_error(ctx, "hbs_pair_alloc: " what, (int)e); rc = HBS_E_HIP; }

    /* layout: whole chunks of 1 GiB, the size rounded up (a remainder mapped as a chunk of its own size made hipMemSetAccess
     * fail with "invalid argument" on this stack, so every mapping is one GiB: up to a GiB more than asked for is held) */
    const uint64_t total = up(bytes, kGran);
    const bool want_probe = d_peer != nullptr && peer_bytes >= kChunk / 2 && total >= kChunk && !getenv("HBS_PAIR_NO_PROBE");
    if (!want_probe) return HBS_E_CAPACITY;                           /* nothing to place: the plain way */
    const uint64_t nfull = (total + kChunk - 1) / kChunk;
    const uint64_t rest = 0;
    Pair* p = new (std::nothrow) Pair();
    if (!p) return HBS_E_HIP;
    p->device = device;
    p->va_bytes = nfull * kChunk + rest;
    int rc = 0;
    p->va = va_take(p->va_bytes);
    if (!p->va) { delete p; return HBS_E_CAPACITY; }                  /* the address pool is used up: the caller takes the plain way */
    uint8_t* const base = static_cast<uint8_t*>(p->va);
    auto map_at = [&](uint64_t off, uint64_t size, hipMemGenericAllocationHandle_t h) -> bool {
        e = hipMemMap(base + off, size, 0, h, 0);
        if (e != hipSuccess) { PAIR_FAIL("hipMemMap") return false; }
        e = hipMemSetAccess(base + off, size, &acc, 1);
        if (e != hipSuccess) { PAIR_FAIL("hipMemSetAccess") (void)hipMemUnmap(base + off, size); return false; }
        p->handles.push_back(h); p->offs.push_back(off); p->sizes.push_back(size);
        return true;
    };
    if (rest) {
        hipMemGenericAllocationHandle_t h;
        e = hipMemCreate(&h, rest, &prop, 0);
        if (e != hipSuccess) { PAIR_FAIL("hipMemCreate") }
        else if (!map_at(nfull * kChunk, rest, h)) (void)hipMemRelease(h);
    }

    struct Cand { hipMemGenericAllocationHandle_t h; uint8_t* va; int other; bool used; };     /* other: 1 = not R's class, 0 = R's, -1 = unknown */
    std::vector<Cand> cands;
    std::vector<hipMemGenericAllocationHandle_t> ballast;
    /* (24 candidates past the chunks until round 4's last day: a box whose allocator handed out 35 chunks of the unwanted class in
     * a row ended with 5 of 16 chunks placed and 0.713 instead of 0.726.  Candidates and ballast go back at the end; both stop
     * while kKeepFree of the device's memory is still free.) */
    constexpr int kExtraCands = 88;
    constexpr uint64_t kBallastMax = 128ull << 30, kKeepFree = 24ull << 30;
    auto room_for = [&](uint64_t more) -> bool {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return false; }
        return (uint64_t)fr >= more + kKeepFree;
    };
    const uint64_t half = (kChunk / 2) / kTile * kTile;
    void* scratch = nullptr;
    const uint64_t scratch_slots = nfull + kExtraCands + 1;
    Prober* pr = nullptr;
    if (rc == 0 && nfull) {
        scratch = va_take(scratch_slots * kChunk);
        if (!scratch) { release_pair(p); return HBS_E_CAPACITY; }
        pr = new (std::nothrow) Prober(st, device);
        if (pr && !pr->ok) { delete pr; pr = nullptr; }
    }
    auto new_cand = [&]() -> bool {                      /* false: no more memory (or address slots) */
        if (cands.size() >= scratch_slots || (cands.size() >= nfull && !room_for(kChunk))) return false;
        Cand c; c.other = -1; c.used = false;
        hipError_t ce = hipMemCreate(&c.h, kChunk, &prop, 0);
        if (ce != hipSuccess) { if (getenv("HBS_PAIR_DEBUG")) fprintf(stderr, "hbs_pair_alloc: hipMemCreate: %s\n", hipGetErrorString(ce)); (void)hipGetLastError(); return false; }
        c.va = static_cast<uint8_t*>(scratch) + cands.size() * kChunk;
        ce = hipMemMap(c.va, kChunk, 0, c.h, 0);
        if (ce == hipSuccess) ce = hipMemSetAccess(c.va, kChunk, &acc, 1);
        if (ce != hipSuccess) {
            if (getenv("HBS_PAIR_DEBUG")) fprintf(stderr, "hbs_pair_alloc: mapping a candidate at %p: %s\n", (void*)c.va, hipGetErrorString(ce));
            (void)hipGetLastError(); (void)hipMemRelease(c.h); return false;
        }
        cands.push_back(c);
        return true;
    };
    std::vector<int> want(nfull, -1);                    /* class wanted for chunk k: 1 = not R's, 0 = R's, -1 = no preference */
    if (getenv("HBS_PAIR_DEBUG") && nfull) fprintf(stderr, "hbs_pair_alloc: %llu chunks + %llu bytes, prober %s\n", (unsigned long long)nfull, (unsigned long long)rest, pr ? "ready" : "NOT available");
    if (rc == 0 && nfull && pr && new_cand()) {
        Cand& R = cands[0];
        const double self_r = pr->copy_ms(R.va, R.va + half, half);
        R.other = 0;
        const bool dbg = getenv("HBS_PAIR_DEBUG") != nullptr;
        uint64_t need[2] = {0, 0};
        for (uint64_t k = 0; k < nfull && self_r > 0; ++k) {
            const uint64_t peer_last = peer_bytes > half ? (uint64_t)((peer_bytes - half) & ~(uint64_t)15) : (uint64_t)0;
            const uint64_t peer_off = std::min(k * kChunk, peer_last);
            const double t = pr->copy_ms(static_cast<const uint8_t*>(d_peer) + peer_off, R.va + half, half);
            if (t <= 0) break;
            const int peer_other = t < kFastRatio * self_r ? 1 : 0;          /* the peer piece is in the class R is not in */
            want[k] = 1 - peer_other;
            need[want[k]] += 1;
            r.probed += 1;
            if (dbg) fprintf(stderr, "hbs_pair_alloc: peer piece %llu against the reference chunk: %.4f / %.4f = %.4f -> chunk wants class %d\n",
                             (unsigned long long)k, t, self_r, t / self_r, want[k]);
        }
        uint64_t have[2] = {1, 0};
        uint64_t ballast_bytes = 0, next_ballast = 4ull << 30;
        int unwanted_in_a_row = 0;
        while ((have[0] < need[0] || have[1] < need[1]) && cands.size() < nfull + (uint64_t)kExtraCands) {
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
            if (!new_cand()) break;
            Cand& X = cands.back();
            const double self_x = pr->copy_ms(X.va, X.va + half, half);
            const double t = pr->copy_ms(cands[0].va, X.va + half, half);
            if (self_x <= 0 || t <= 0) break;
            X.other = t < kFastRatio * self_x ? 1 : 0;
            r.probed += 1;
            const bool wanted = have[X.other] < need[X.other];
            have[X.other] += 1;
            unwanted_in_a_row = wanted ? 0 : unwanted_in_a_row + 1;
            if (!wanted) r.rejected += 1;
            if (dbg) fprintf(stderr, "hbs_pair_alloc: candidate %zu against the reference chunk: %.4f / %.4f = %.4f -> class %d%s\n",
                             cands.size() - 1, t, self_x, t / self_x, X.other, wanted ? "" : " (not needed)");
        }
    }
    /* hand the candidates out: first to the chunks that want their class, then whatever is left to whoever is left */
    std::vector<int> pick(nfull, -1);
    for (int pass = 0; pass < 2 && rc == 0; ++pass)
        for (uint64_t k = 0; k < nfull; ++k) {
            if (pick[k] >= 0) continue;
            for (size_t c = 0; c < cands.size(); ++c) {
                if (cands[c].used) continue;
                if (pass == 0 && (want[k] < 0 || cands[c].other != want[k])) continue;
                pick[k] = (int)c; cands[c].used = true;
                if (pass == 0) r.accepted_fast += 1; else r.unprobed_after_budget += 1;
                break;
            }
        }
    for (uint64_t k = 0; k < nfull && rc == 0; ++k) {
        hipMemGenericAllocationHandle_t h;
        if (pick[k] >= 0) {
            Cand& c = cands[(size_t)pick[k]];
            e = hipMemUnmap(c.va, kChunk);
            if (e != hipSuccess) { PAIR_FAIL("hipMemUnmap") break; }
            h = c.h;
        } else {
            e = hipMemCreate(&h, kChunk, &prop, 0);
            if (e != hipSuccess) { PAIR_FAIL("hipMemCreate") break; }
            r.unprobed_after_budget += 1;
        }
        if (!map_at(k * kChunk, kChunk, h)) {
            if (pick[k] >= 0) { cands[(size_t)pick[k]].used = false; cands[(size_t)pick[k]].va = nullptr; }    /* released with the rest below */
            else (void)hipMemRelease(h);
            break;
        }
    }
    /* everything that was not chosen goes back */
    (void)hipStreamSynchronize(st);
    for (auto& c : cands) {
        if (c.used) continue;
        if (c.va) (void)hipMemUnmap(c.va, kChunk);
        (void)hipMemRelease(c.h);
    }
    for (auto b : ballast) (void)hipMemRelease(b);
    delete pr;
#undef PAIR_FAIL
    if (rc) { release_pair(p); return rc; }
    r.chunks = (uint32_t)p->handles.size();
    r.unprobed_after_budget += (uint32_t)(rest ? 1 : 0);
    {
        std::lock_guard<std::mutex> g(g_mu);
        g_chunked[p->va] = p;
    }
    *out = p->va;
    if (rep) *rep = r;
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

/* Two ways.  (1) Chunked: the buffer is put together from 1 GiB physical chunks, each classed by measurement, at addresses
 * that are used once (see va_take) -- every piece can be placed, whatever the allocator hands out.  (2) Plain: whole
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
    (void)hipSetDevice(device);
    if (ctx) (void)hbs_ctx_synchronize(ctx);
    (void)hipDeviceSynchronize();
    if (p) { release_pair(p); return 0; }
    return hipFree(ptr) == hipSuccess ? 0 : HBS_E_HIP;
}

} // extern "C"
