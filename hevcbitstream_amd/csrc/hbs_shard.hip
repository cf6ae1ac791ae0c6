/*
 * hbs_shard.hip -- the multi-GPU side of the path behind the C ABI (SURVEY.md 8(e), 8(b) "required new exports"):
 * one process per GPU, every rank indexes its own bytes, and the ONE exchange is the gather of the NAL index --
 * counts first (8 bytes per rank), then exactly count x 32 bytes per rank, to one root or to everybody, RCCL over
 * xGMI.  No collective touches stream bytes or RBSP arenas.
 *
 * RCCL is not linked: it is looked up at run time (dlopen of librccl.so.1) when the first communicator is made, so
 * that (a) the single-GPU library has no dependency on it and (b) inside a process that already carries an RCCL --
 * PyTorch ships its own -- the very same copy is used and a communicator the host application already owns can be
 * adopted (hbs_comm_adopt).
 *
 * Also here: cutting ONE stream into per-rank parts.  A part begins at the first start code (00 00 01) at or after
 * its nominal boundary -- hbs_find_cut_host, the same deterministic rule on both sides of a boundary, so neighbours
 * agree without talking -- and is scanned with the first bytes of the next part behind it, which terminate its last
 * NAL exactly as they do in the whole stream (find_nal_unit's end search, h264_nal.c:64-72); the NAL that those
 * halo bytes open belongs to the next part and is dropped (hbs_trim_part).  Concatenating the parts' indexes with
 * their cut offsets added is the index of the whole stream (tests/test_shard_gloo.py, tests/test_gpu_shard.py).
 */
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <new>
#include "hbs_common.h"

struct hbs_ctx;
extern "C" void* hbs_ctx_get_stream(hbs_ctx* ctx);
extern "C" int hbs_ctx_device(hbs_ctx* ctx);

namespace {

struct Rccl {
    void* so;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*);
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
    ncclResult_t (*GroupStart)();
    ncclResult_t (*GroupEnd)();
    const char* (*GetErrorString)(ncclResult_t);
};
Rccl g_rccl;

bool load_rccl()
{
    if (g_rccl.so) return true;
    /* HBS_RCCL_LIB (a supported override, include/hevcbitstream_amd.h): the path of the library to take RCCL's ten entry points
     * from instead of the librccl.so.1 the loader finds -- a site's own RCCL build, or, in this project's tests, a shared-memory
     * transport between processes that share ONE GPU (tests/sim/fake_rccl.cpp), which RCCL itself refuses.  Read once, at the
     * first communicator; a library named here that does not load is an error, never a silent fall-back. */
    const char* alt = getenv("HBS_RCCL_LIB");
    void* so = alt && *alt ? dlopen(alt, RTLD_NOW | RTLD_GLOBAL) : dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!so && !(alt && *alt)) so = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!so) { fprintf(stderr, "hevcbitstream_amd: RCCL not found (%s)\n", dlerror()); return false; }
    Rccl r;
    r.so = so;
#define HBS_SYM(field, name) *reinterpret_cast<void**>(&r.field) = dlsym(so, name); if (!r.field) { fprintf(stderr, "hevcbitstream_amd: %s missing in RCCL\n", name); return false; }
    HBS_SYM(GetUniqueId, "ncclGetUniqueId") HBS_SYM(CommInitRank, "ncclCommInitRank") HBS_SYM(CommDestroy, "ncclCommDestroy")
    HBS_SYM(AllGather, "ncclAllGather") HBS_SYM(Broadcast, "ncclBroadcast") HBS_SYM(Send, "ncclSend") HBS_SYM(Recv, "ncclRecv")
    HBS_SYM(GroupStart, "ncclGroupStart") HBS_SYM(GroupEnd, "ncclGroupEnd") HBS_SYM(GetErrorString, "ncclGetErrorString")
#undef HBS_SYM
    g_rccl = r;
    return true;
}

__global__ void k_rebase(const hbs_nal_entry* __restrict__ in, uint64_t n, uint64_t stream_base, uint64_t rbsp_base, hbs_nal_entry* __restrict__ out)
{
    for (uint64_t k = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        hbs_nal_entry e = in[k];
        e.start += stream_base; e.end += stream_base; e.rbsp_off += rbsp_base;
        out[k] = e;
    }
}

} // namespace

/* What every rank tells the others before a byte of payload moves: kGatherWords words per rank, all-gathered.  With them
 * every rank takes the SAME decision -- go on, or return an error -- so that nobody is left waiting in a collective the
 * others never enter (round 2's advice: a receiver that found its buffer too small used to return alone). */
enum : int { kGwCount = 0, kGwCap = 1, kGwStatus = 2, kGwStopped = 3, kGatherWords = 4 };
constexpr int kMaxWorld = 1024;

struct hbs_comm {
    ncclComm_t comm;
    int rank, world, owned, device;
    unsigned long long* d_counts;      /* (world + 1) x kGatherWords words: [0, world) everybody's, [world] mine */
    hbs_nal_entry* d_stage;            /* rebased copy of the local entries */
    uint64_t stage_cap;
};

extern "C" {

int hbs_comm_unique_id(uint8_t id[HBS_COMM_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) == HBS_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    if (!id) return HBS_E_ARG;
    if (!load_rccl()) return HBS_E_HIP;
    ncclUniqueId u;
    if (g_rccl.GetUniqueId(&u) != ncclSuccess) return HBS_E_HIP;
    memcpy(id, &u, sizeof(u));
    return 0;
}

static int comm_new(hbs_ctx* ctx, ncclComm_t c, int owned, int rank, int world, hbs_comm** out)
{
    hbs_comm* h = new (std::nothrow) hbs_comm();
    if (!h) return HBS_E_HIP;
    h->comm = c; h->rank = rank; h->world = world; h->owned = owned; h->device = hbs_ctx_device(ctx);
    h->d_stage = nullptr; h->stage_cap = 0;
    if (hipMalloc(reinterpret_cast<void**>(&h->d_counts), (size_t)(world + 1) * kGatherWords * sizeof(unsigned long long)) != hipSuccess) { delete h; return HBS_E_HIP; }
    *out = h;
    return 0;
}

int hbs_comm_create(hbs_ctx* ctx, const uint8_t id[HBS_COMM_ID_BYTES], int rank, int world, hbs_comm** out)
{
    if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world) return HBS_E_ARG;
    if (!load_rccl()) return HBS_E_HIP;
    if (hipSetDevice(hbs_ctx_device(ctx)) != hipSuccess) return HBS_E_NO_DEVICE;
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t c;
    const ncclResult_t r = g_rccl.CommInitRank(&c, world, u, rank);
    if (r != ncclSuccess) { fprintf(stderr, "hevcbitstream_amd: ncclCommInitRank: %s\n", g_rccl.GetErrorString(r)); return HBS_E_HIP; }
    const int rc = comm_new(ctx, c, 1, rank, world, out);
    if (rc) (void)g_rccl.CommDestroy(c);
    return rc;
}

int hbs_comm_adopt(hbs_ctx* ctx, void* nccl_comm, int rank, int world, hbs_comm** out)
{
    if (!ctx || !nccl_comm || !out || world < 1 || rank < 0 || rank >= world) return HBS_E_ARG;
    if (!load_rccl()) return HBS_E_HIP;
    return comm_new(ctx, static_cast<ncclComm_t>(nccl_comm), 0, rank, world, out);
}

void hbs_comm_destroy(hbs_comm* h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->d_counts) (void)hipFree(h->d_counts);
    if (h->d_stage) (void)hipFree(h->d_stage);
    if (h->owned && g_rccl.so) (void)g_rccl.CommDestroy(h->comm);
    delete h;
}

int hbs_comm_rank(const hbs_comm* h) { return h ? h->rank : HBS_E_ARG; }
int hbs_comm_world(const hbs_comm* h) { return h ? h->world : HBS_E_ARG; }

/* Workgroup slots to leave beside the persistent scan (hbs_ctx_reserve_workgroups) so that the gather's kernels find a CU while
 * the next scan runs.  A function of the world: the all-to-all form of the exchange (root = -1) is one grouped broadcast per
 * rank, and RCCL runs a group's transfers on one channel -- one workgroup -- per peer at least; measured at world 1 (round 3:
 * with 8 or 16 free slots RCCL still waited for the scan, with 32 a 54 MB copy ran beside it), so: 8 per peer, at least 32 (a
 * lone rank's local copy), at most 64 of the 512 (one eighth of the GPU: beyond it the scan loses more than the gather wins). */
int hbs_comm_reserve_hint(const hbs_comm* h)
{
    if (!h) return HBS_E_ARG;
    const int want = 8 * (h->world - 1);
    return want < 32 ? 32 : want > 64 ? 64 : want;
}

static int gather_impl(hbs_ctx* ctx, hbs_comm* h, const hbs_nal_entry* d_index, uint64_t n_local, int stopped,
                       uint64_t stream_base, uint64_t rbsp_base, int root,
                       hbs_nal_entry* d_all, uint64_t cap_all, uint64_t* counts_out)
{
    /* argument errors are the same on every rank or a bug of the caller: nothing collective has happened yet */
    if (!ctx || !h || (n_local && !d_index) || root >= h->world || !counts_out || h->world > kMaxWorld) return HBS_E_ARG;
    if (hipSetDevice(h->device) != hipSuccess) return HBS_E_NO_DEVICE;
    hipStream_t st = static_cast<hipStream_t>(hbs_ctx_get_stream(ctx));
    const int W = h->world;
    const bool receiver = root < 0 || root == h->rank;
    /* 0. everything that can fail locally happens BEFORE the first collective, and its outcome travels with the counts */
    unsigned long long status = 0;
    const bool rebase = n_local && (stream_base || rbsp_base);
    if (rebase && h->stage_cap < n_local) {
        if (h->d_stage) (void)hipFree(h->d_stage);
        h->d_stage = nullptr; h->stage_cap = 0;
        if (hipMalloc(reinterpret_cast<void**>(&h->d_stage), n_local * sizeof(hbs_nal_entry)) != hipSuccess) status = 1;
        else h->stage_cap = n_local;
    }
    /* 1. {count, capacity, status, stopped}: 32 bytes per rank, to everybody */
    unsigned long long mine[kGatherWords];
    mine[kGwCount] = n_local;
    mine[kGwCap] = receiver ? (d_all ? cap_all : 0ull) : ~0ull;         /* a rank that receives nothing has room for anything */
    mine[kGwStatus] = status;
    mine[kGwStopped] = stopped ? 1ull : 0ull;
    unsigned long long* d_mine = h->d_counts + (size_t)W * kGatherWords;
    if (hipMemcpyAsync(d_mine, mine, sizeof(mine), hipMemcpyHostToDevice, st) != hipSuccess) return HBS_E_HIP;
    if (g_rccl.AllGather(d_mine, h->d_counts, kGatherWords, ncclUint64, h->comm, st) != ncclSuccess) return HBS_E_HIP;
    static thread_local unsigned long long words[kMaxWorld * kGatherWords];
    if (hipMemcpyAsync(words, h->d_counts, (size_t)W * sizeof(mine), hipMemcpyDeviceToHost, st) != hipSuccess) return HBS_E_HIP;
    if (hipStreamSynchronize(st) != hipSuccess) return HBS_E_HIP;        /* the sizes of the receives are host values */
    /* 2. the same decision on every rank.  Parts of ONE stream (hbs_gather_parts): the whole-stream walk ends at the first empty
     * NAL (hevc_analyze.c:135), so the parts behind the first one that stopped there contribute nothing. */
    uint64_t total = 0, min_cap = ~0ull;
    bool failed = false, cut = false;
    for (int r = 0; r < W; ++r) {
        unsigned long long* w = words + (size_t)r * kGatherWords;
        if (cut) w[kGwCount] = 0;
        if (w[kGwStopped]) cut = true;
        counts_out[r] = w[kGwCount];
        total += w[kGwCount];
        if (w[kGwCap] < min_cap) min_cap = w[kGwCap];
        if (w[kGwStatus]) failed = true;
    }
    if (failed) return HBS_E_HIP;
    if (total > min_cap) return HBS_E_CAPACITY;                          /* on EVERY rank: some receiver's d_all is too small */
    const uint64_t n_send = counts_out[h->rank];                         /* n_local, or 0 behind a part that stopped */
    /* 3. my entries with global offsets (only when a base is given: independent shards keep theirs) */
    const hbs_nal_entry* src = d_index;
    if (n_send && rebase) {
        uint64_t blocks = (n_send + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        k_rebase<<<dim3((unsigned)blocks), 256, 0, st>>>(d_index, n_send, stream_base, rbsp_base, h->d_stage);
        src = h->d_stage;
    }
    /* 4. exactly count x 32 bytes per rank */
    uint64_t off = 0;
    ncclResult_t rr = g_rccl.GroupStart();
    for (int r = 0; r < W && rr == ncclSuccess; ++r) {
        const size_t bytes = (size_t)counts_out[r] * sizeof(hbs_nal_entry);
        if (bytes) {
            if (root < 0) {
                rr = g_rccl.Broadcast(r == h->rank ? (const void*)src : (const void*)(d_all + off), d_all + off, bytes, ncclUint8, r, h->comm, st);
            } else if (h->rank == root) {
                if (r == root) { if (hipMemcpyAsync(d_all + off, src, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) rr = ncclUnhandledCudaError; }
                else rr = g_rccl.Recv(d_all + off, bytes, ncclUint8, r, h->comm, st);
            } else if (r == h->rank) {
                rr = g_rccl.Send(src, bytes, ncclUint8, root, h->comm, st);
            }
        }
        off += counts_out[r];
    }
    const ncclResult_t re = g_rccl.GroupEnd();
    if (rr != ncclSuccess || re != ncclSuccess) {
        fprintf(stderr, "hevcbitstream_amd: hbs_gather_index: %s\n", g_rccl.GetErrorString(rr != ncclSuccess ? rr : re));
        return HBS_E_HIP;
    }
    return 0;
}

int hbs_gather_index(hbs_ctx* ctx, hbs_comm* h, const hbs_nal_entry* d_index, uint64_t n_local,
                     uint64_t stream_base, uint64_t rbsp_base, int root,
                     hbs_nal_entry* d_all, uint64_t cap_all, uint64_t* counts_out)
{
    return gather_impl(ctx, h, d_index, n_local, 0, stream_base, rbsp_base, root, d_all, cap_all, counts_out);
}

int hbs_gather_parts(hbs_ctx* ctx, hbs_comm* h, const hbs_nal_entry* d_index, uint64_t n_local, int stopped,
                     uint64_t stream_base, uint64_t rbsp_base, int root,
                     hbs_nal_entry* d_all, uint64_t cap_all, uint64_t* counts_out)
{
    return gather_impl(ctx, h, d_index, n_local, stopped, stream_base, rbsp_base, root, d_all, cap_all, counts_out);
}

/* ---- one stream, several parts ---------------------------------------------------------------------------- */

uint64_t hbs_find_cut_host(const uint8_t* bytes, uint64_t n, uint64_t from)
{
    if (!bytes) return ~0ull;
    /* a cut needs 8 bytes behind it: the part in front is scanned with them as its halo, and a start code in the last bytes
     * of a stream is subject to find_nal_unit's end-of-buffer clauses (h264_nal.c:52), which only the scan of the true end applies */
    for (uint64_t p = from; p + 8 <= n; ++p)
        if (bytes[p] == 0 && bytes[p + 1] == 0 && bytes[p + 2] == 1) return p;
    return ~0ull;
}

int hbs_trim_part(hbs_ctx* ctx, const hbs_nal_entry* d_index, uint64_t nal_count, uint64_t rbsp_bytes, uint64_t part_bytes,
                  uint64_t* n_kept, uint64_t* rbsp_kept)
{
    if (!ctx || !n_kept || !rbsp_kept || (nal_count && !d_index)) return HBS_E_ARG;
    if (hipSetDevice(hbs_ctx_device(ctx)) != hipSuccess) return HBS_E_NO_DEVICE;
    hipStream_t st = static_cast<hipStream_t>(hbs_ctx_get_stream(ctx));
    *n_kept = nal_count; *rbsp_kept = rbsp_bytes;
    /* the halo is at most one start code and a few bytes: at most two entries can begin in it */
    for (int back = 0; back < 2 && *n_kept > 0; ++back) {
        hbs_nal_entry e;
        if (hipMemcpyAsync(&e, d_index + (*n_kept - 1), sizeof(e), hipMemcpyDeviceToHost, st) != hipSuccess) return HBS_E_HIP;
        if (hipStreamSynchronize(st) != hipSuccess) return HBS_E_HIP;
        if (e.start < part_bytes + 3) break;              /* its start code begins inside the part: mine */
        *n_kept -= 1;
        *rbsp_kept = e.rbsp_off;
    }
    return 0;
}

} // extern "C"
