// fake_rccl.cpp -- TEST INFRASTRUCTURE, never part of the product: the ten RCCL entry points hbs_shard.hip looks up, carried
// by a POSIX shared-memory segment between processes that share ONE GPU.  RCCL itself refuses two ranks on one device, and the
// GPU boxes of this project have one: with this stand-in (the library loads it when HBS_RCCL_LIB names it) the multi-rank
// paths of hbs_gather_index / hbs_gather_parts -- grouped broadcasts to all, send / receive to a root, the collective error
// decisions -- run with world > 1 in tests/test_gpu_shard.py.
//
// What it checks that real RCCL would answer with a hang: every posted send / broadcast must be taken by exactly the ranks it
// is addressed to, with the same byte count, inside the same group; a rank that never arrives makes the barrier time out.
// Every operation is synchronous (stream drained, bytes staged through the segment): semantics, not speed.
//
// Environment: HBS_FAKE_RCCL_SHM = name of the segment (created and sized by the test), HBS_FAKE_RCCL_SLOT = bytes of data
// area per rank.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

namespace {

constexpr int kMaxRanks = 8, kMaxMsgs = 64;
struct Msg { int dst; int pad; unsigned long long bytes, off; std::atomic<int> taken; };
struct RankBox { std::atomic<int> nmsgs; Msg msgs[kMaxMsgs]; };
struct Header {
    std::atomic<int> arrived, generation;
    std::atomic<int> failed;             // any rank saw a protocol error: everybody returns an error from the current call on
    RankBox box[kMaxRanks];
};
struct Comm { int rank, world; Header* h; unsigned char* data; size_t slot; };
struct Op { int kind; const void* send; void* recv; size_t bytes; int peer; };      // kind: 0 send, 1 recv, 2 broadcast (peer = root)

thread_local std::vector<Op> g_group;
thread_local int g_depth = 0;
thread_local Comm* g_group_comm = nullptr;
thread_local hipStream_t g_group_stream = nullptr;
thread_local Comm* g_last_comm = nullptr;          // a group in which THIS rank posts nothing still takes part in the round
thread_local hipStream_t g_last_stream = nullptr;

size_t dtype_bytes(ncclDataType_t t)
{
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 2;
    }
}

bool barrier(Comm* c)
{
    Header* h = c->h;
    const int gen = h->generation.load();
    if (h->arrived.fetch_add(1) + 1 == c->world) { h->arrived.store(0); h->generation.fetch_add(1); return h->failed.load() == 0; }
    const auto t0 = std::chrono::steady_clock::now();
    while (h->generation.load() == gen) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {
            fprintf(stderr, "fake_rccl: rank %d waited 20 s at a barrier: another rank left the protocol\n", c->rank);
            h->failed.store(1);
            return false;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    return h->failed.load() == 0;
}

unsigned char* slot_of(Comm* c, int r) { return c->data + (size_t)r * c->slot; }

// one exchange round: everybody posts, barrier, everybody takes, barrier, everybody checks that its posts were taken
ncclResult_t run_group(Comm* c, hipStream_t st, const std::vector<Op>& ops)
{
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    RankBox& mine = c->h->box[c->rank];
    size_t used = 0;
    int n = 0;
    bool bad = false;
    for (const Op& o : ops) {
        const bool post = o.kind == 0 || (o.kind == 2 && o.peer == c->rank);
        if (!post) continue;
        if (n >= kMaxMsgs || used + o.bytes > c->slot) { fprintf(stderr, "fake_rccl: rank %d: outbox too small\n", c->rank); bad = true; break; }
        Msg& m = mine.msgs[n++];
        m.dst = o.kind == 0 ? o.peer : -1; m.bytes = o.bytes; m.off = used; m.taken.store(0);
        if (hipMemcpy(slot_of(c, c->rank) + used, o.send, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) bad = true;
        used += o.bytes;
    }
    mine.nmsgs.store(n);
    if (bad) c->h->failed.store(1);
    if (!barrier(c)) return ncclInternalError;
    for (const Op& o : ops) {
        int from = -1;
        if (o.kind == 1) from = o.peer;
        else if (o.kind == 2 && o.peer != c->rank) from = o.peer;
        else if (o.kind == 2 && o.peer == c->rank && o.recv != o.send) {            // the root's own copy
            if (hipMemcpy(o.recv, o.send, o.bytes, hipMemcpyDeviceToDevice) != hipSuccess) bad = true;
            continue;
        } else continue;
        RankBox& b = c->h->box[from];
        bool found = false;
        for (int i = 0; i < b.nmsgs.load() && !found; ++i) {
            Msg& m = b.msgs[i];
            const bool addressed = o.kind == 1 ? m.dst == c->rank : m.dst == -1;
            if (!addressed) continue;
            int seen = m.taken.load();
            if (o.kind == 1 && seen != 0) continue;                                  // a second receive takes the next send
            if (o.kind == 2 && (seen & (1 << c->rank))) continue;
            if (m.bytes != o.bytes) { fprintf(stderr, "fake_rccl: rank %d expects %zu bytes from rank %d, which posted %llu\n", c->rank, o.bytes, from, m.bytes); bad = true; found = true; break; }
            if (hipMemcpy(o.recv, slot_of(c, from) + m.off, o.bytes, hipMemcpyHostToDevice) != hipSuccess) bad = true;
            if (o.kind == 1) m.taken.store(1); else m.taken.fetch_or(1 << c->rank);
            found = true;
        }
        if (!found) { fprintf(stderr, "fake_rccl: rank %d waits for %zu bytes from rank %d that were never posted (real RCCL would hang here)\n", c->rank, o.bytes, from); bad = true; }
    }
    if (bad) c->h->failed.store(1);
    if (!barrier(c)) return ncclInternalError;
    for (int i = 0; i < n; ++i) {
        const Msg& m = mine.msgs[i];
        const int want = m.dst >= 0 ? 1 : (((1 << c->world) - 1) & ~(1 << c->rank));
        if (m.taken.load() != want) { fprintf(stderr, "fake_rccl: rank %d posted %llu bytes (to %d) that nobody took (real RCCL would hang here)\n", c->rank, m.bytes, m.dst); bad = true; }
    }
    if (bad) c->h->failed.store(1);
    if (!barrier(c)) return ncclInternalError;
    return ncclSuccess;
}

} // namespace

extern "C" {

// bytes of the segment the test has to create for `nranks` ranks with `slot` bytes of data area each
unsigned long long fake_rccl_segment_bytes(int nranks, unsigned long long slot) { return sizeof(Header) + (unsigned long long)nranks * slot; }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) { memset(id, 0x5A, sizeof(*id)); return ncclSuccess; }

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId, int rank)
{
    const char* name = getenv("HBS_FAKE_RCCL_SHM");
    const char* slot = getenv("HBS_FAKE_RCCL_SLOT");
    if (!name || !slot || nranks > kMaxRanks) return ncclInvalidArgument;
    const int fd = shm_open(name, O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    Comm* c = new Comm();
    c->rank = rank; c->world = nranks; c->slot = (size_t)atoll(slot);
    const size_t total = sizeof(Header) + (size_t)nranks * c->slot;
    void* p = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->h = static_cast<Header*>(p);
    c->data = static_cast<unsigned char*>(p) + sizeof(Header);
    *out = reinterpret_cast<ncclComm_t>(c);
    return barrier(c) ? ncclSuccess : ncclInternalError;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete reinterpret_cast<Comm*>(comm); return ncclSuccess; }

ncclResult_t ncclGroupStart() { if (g_depth++ == 0) { g_group.clear(); g_group_comm = nullptr; } return ncclSuccess; }

ncclResult_t ncclGroupEnd()
{
    if (--g_depth != 0) return ncclSuccess;
    if (!g_group_comm) { g_group_comm = g_last_comm; g_group_stream = g_last_stream; }      // nothing posted here: the others may have
    if (!g_group_comm) return ncclSuccess;
    return run_group(g_group_comm, g_group_stream, g_group);
}

static ncclResult_t add_op(ncclComm_t comm, hipStream_t st, Op o)
{
    Comm* c = reinterpret_cast<Comm*>(comm);
    g_last_comm = c; g_last_stream = st;
    if (g_depth == 0) { std::vector<Op> one{o}; return run_group(c, st, one); }
    g_group_comm = c; g_group_stream = st; g_group.push_back(o);
    return ncclSuccess;
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t st)
{
    return add_op(comm, st, Op{0, buf, nullptr, count * dtype_bytes(t), peer});
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t st)
{
    return add_op(comm, st, Op{1, nullptr, buf, count * dtype_bytes(t), peer});
}
ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t t, int root, ncclComm_t comm, hipStream_t st)
{
    return add_op(comm, st, Op{2, send, recv, count * dtype_bytes(t), root});
}

// all-gather = every rank broadcasts its piece (an exchange round of its own; an empty group elsewhere is not allowed to mix with it)
ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t st)
{
    Comm* c = reinterpret_cast<Comm*>(comm);
    g_last_comm = c; g_last_stream = st;
    const size_t bytes = count * dtype_bytes(t);
    std::vector<Op> ops;
    for (int r = 0; r < c->world; ++r) ops.push_back(Op{2, send, static_cast<unsigned char*>(recv) + (size_t)r * bytes, bytes, r});
    return run_group(c, st, ops);
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake_rccl: protocol error (see stderr)"; }

} // extern "C"
