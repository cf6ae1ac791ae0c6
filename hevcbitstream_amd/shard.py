"""Multi-GPU sharding of the hot path: one process per GPU, each rank indexes its own bytes; the only exchange is the
gather of the NAL index -- counts first, then exactly count x 32 bytes per rank.  No collective touches the data path:
stream bytes and RBSP arenas stay on their GPUs.

Two carriers of the same exchange:
  * LibraryComm -- the C ABI (include/hevcbitstream_amd.h: hbs_comm_*, hbs_gather_index / hbs_gather_parts; RCCL over xGMI,
    looked up by the library at run time).  torch.distributed only carries the 128-byte communicator id to the ranks.  What a
    C caller uses, and what `bench.py --gpus N` times (PipelinedLibraryGather below keeps the exchange of one step under the
    scan of the next).
  * gather_index / IndexGatherer -- the same exchange over torch.distributed collectives, padded to the largest count
    (`gloo` on CPU tensors in the world-size-2 CPU tests, which exercise the host-side protocol without a GPU; `nccl` on GPU
    tensors).  Not what the benchmark measures.
Also: parts of ONE stream (cut_points / part_ranges over hbs_find_cut_host)."""
import ctypes as C

import numpy as np

from .api import NAL_ENTRY, HbsError, load_library

ENTRY_BYTES = NAL_ENTRY.itemsize
HALO_BYTES = 8                      # bytes of the next part scanned behind a part (they end its last NAL)


class LibraryComm:
    """hbs_comm of the C ABI.  Collective: every rank constructs it; rank 0's id travels through `dist` (any backend)."""

    def __init__(self, ctx, dist, rank, world, group=None, ident=None):
        """ident: the 128-byte id when the caller carries it to the ranks by its own means (then `dist` is not used)"""
        self.ctx, self.lib, self.rank, self.world = ctx, load_library(), rank, world
        lib = self.lib
        lib.hbs_comm_unique_id.argtypes = [C.c_char_p]
        lib.hbs_comm_create.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        lib.hbs_comm_destroy.argtypes = [C.c_void_p]
        lib.hbs_comm_destroy.restype = None
        lib.hbs_gather_index.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int,
                                         C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        lib.hbs_gather_parts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_uint64, C.c_uint64, C.c_int,
                                         C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        lib.hbs_comm_world.argtypes = [C.c_void_p]
        if ident is None:
            buf = C.create_string_buffer(128)
            if rank == 0 and lib.hbs_comm_unique_id(buf) != 0:
                raise HbsError("hbs_comm_unique_id failed (RCCL not found?)")
            box = [bytes(buf.raw)]
            if world > 1:
                dist.broadcast_object_list(box, src=0, group=group)
        else:
            box = [bytes(ident)]
        h = C.c_void_p()
        rc = lib.hbs_comm_create(ctx.h, box[0], rank, world, C.byref(h))
        if rc != 0:
            raise HbsError("hbs_comm_create(rank %d of %d) failed: %d" % (rank, world, rc))
        self.h = h

    def gather_index(self, local_index, n_local, all_index, stream_base=0, rbsp_base=0, root=-1, stopped=None):
        """Enqueue the exchange on the context's stream.  local_index / all_index: device uint8 tensors of entries.  Returns the
        per-rank counts (list of int); the payload lands in all_index (receiving ranks) in stream order.
        stopped (parts of ONE stream, hbs_gather_parts): this part's scan ended at an empty NAL -- the parts behind it then
        contribute nothing, as in the reference's walk of the whole stream."""
        self.ctx._bind_stream()
        counts = (C.c_uint64 * self.world)()
        d_all = C.c_void_p(all_index.data_ptr()) if all_index is not None else None
        cap = (all_index.numel() // ENTRY_BYTES) if all_index is not None else 0
        if stopped is None:
            rc = self.lib.hbs_gather_index(self.ctx.h, self.h, C.c_void_p(local_index.data_ptr()), n_local, stream_base, rbsp_base, root,
                                           d_all, cap, counts)
        else:
            rc = self.lib.hbs_gather_parts(self.ctx.h, self.h, C.c_void_p(local_index.data_ptr()), n_local, 1 if stopped else 0,
                                           stream_base, rbsp_base, root, d_all, cap, counts)
        if rc != 0:
            e = HbsError("hbs_gather_index failed: %d" % rc)
            e.code = rc
            raise e
        return [int(c) for c in counts]

    def world_seen(self):
        """the communicator's own idea of its size"""
        return int(self.lib.hbs_comm_world(self.h))

    def reserve_hint(self):
        """workgroup slots to leave free beside the scan for this communicator's exchange (hbs_comm_reserve_hint)"""
        self.lib.hbs_comm_reserve_hint.argtypes = [C.c_void_p]
        return int(self.lib.hbs_comm_reserve_hint(self.h))

    def close(self):
        if getattr(self, "h", None):
            self.lib.hbs_comm_destroy(self.h)
            self.h = None


class PipelinedLibraryGather:
    """hbs_gather_index (the C ABI's exchange) kept off the scan's critical path: the gather of step i runs on a stream of its
    own, on a second context, while the scan of step i + 1 is already running on the caller's stream.  The host side of
    hbs_gather_index waits for the 8-byte counts -- that is, for scan i -- so the caller enqueues scan i + 1 BEFORE it
    submits gather i: the GPU always has the next scan queued.  `depth` index buffers alternate; release(k) makes the caller's
    stream wait for the gather that still reads buffer k.  Every gather is bracketed by events on its stream: gather_ms()."""

    def __init__(self, torch, hbs_module, device_index, dist, rank, world, capacity_per_rank, depth=2):
        self.torch = torch
        self.ctx = hbs_module.Context(device_index)                  # supplies the stream and the device of the exchange
        self.stream = torch.cuda.Stream(device=device_index)
        self.comm = LibraryComm(self.ctx, dist, rank, world)
        self.world, self.depth = world, depth
        dev = torch.device("cuda", device_index)
        self.recv = [torch.empty(world * capacity_per_rank * ENTRY_BYTES, dtype=torch.uint8, device=dev) for _ in range(depth)]
        self.scan_done = [torch.cuda.Event() for _ in range(depth)]
        self.gather_done = [None] * depth
        self.t0 = [torch.cuda.Event(enable_timing=True) for _ in range(depth)]
        self.t1 = [torch.cuda.Event(enable_timing=True) for _ in range(depth)]
        self.ms = []
        self.counts = [None] * depth
        self.timed = [False] * depth

    def mark_scan(self, k):
        """the scan that fills buffer k has just been enqueued on the current stream"""
        self.scan_done[k].record()

    def submit(self, k, local_index, n_local):
        """gather buffer k (host: returns once the counts are known, i.e. once scan k is done; payload asynchronous)"""
        self._collect(k)
        torch = self.torch
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(self.scan_done[k])
            self.t0[k].record()
            self.counts[k] = self.comm.gather_index(local_index, n_local, self.recv[k])
            self.t1[k].record()
            ev = torch.cuda.Event()
            ev.record()
            self.gather_done[k] = ev
            self.timed[k] = True
        return self.counts[k]

    def _collect(self, k):
        if self.timed[k]:
            self.t1[k].synchronize()
            self.ms.append(self.t0[k].elapsed_time(self.t1[k]))
            self.timed[k] = False

    def release(self, k):
        """before buffer k is written again: the current stream waits for the gather that reads it"""
        if self.gather_done[k] is not None:
            self.torch.cuda.current_stream().wait_event(self.gather_done[k])

    def drain(self):
        self.stream.synchronize()
        for k in range(self.depth):
            self._collect(k)

    def gather_ms(self, last):
        """durations of the last `last` gathers (events on the exchange's stream: counts all-gather, the wait for them, payload)"""
        return self.ms[-last:]

    def result(self, k):
        self.stream.synchronize()
        return self.recv[k], self.counts[k]

    def close(self):
        self.comm.close()
        self.ctx.close()


def find_cut_host(host_bytes, start):
    """hbs_find_cut_host: offset of the first start code (00 00 01) at or after `start` that has 8 bytes behind it, or None"""
    lib = load_library()
    lib.hbs_find_cut_host.restype = C.c_uint64
    lib.hbs_find_cut_host.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
    a = np.ascontiguousarray(host_bytes, dtype=np.uint8)
    p = lib.hbs_find_cut_host(a.ctypes.data, len(a), start)
    return None if p == 0xFFFFFFFFFFFFFFFF else int(p)


def part_ranges(host_bytes, world):
    """Cut ONE stream for `world` ranks: [(lo, hi, hi_with_halo)] per rank.  Rank r's part starts at the first start code at or
    after r x n / world (rank 0: at 0) and is scanned up to hi_with_halo = min(n, hi + HALO_BYTES).  A rank whose nominal
    range holds no start code gets an empty part.  Every rank computes this from bytes it can read; in a real deployment
    a rank only needs [its nominal start, its nominal end + the distance to the next start code) of the file."""
    n = len(host_bytes)
    cuts = [0]
    for r in range(1, world):
        c = find_cut_host(host_bytes, max(cuts[-1], r * n // world))
        cuts.append(n if c is None else c)
    cuts.append(n)
    return [(cuts[r], cuts[r + 1], min(n, cuts[r + 1] + HALO_BYTES) if cuts[r + 1] < n else n) for r in range(world)]


def trim_part(entries, part_bytes, is_last):
    """host-side mirror of hbs_trim_part for entries already on the host: the NAL(s) the halo opens belong to the next part"""
    if is_last:
        return entries
    keep = len(entries)
    while keep and int(entries["start"][keep - 1]) >= part_bytes + 3:
        keep -= 1
    return entries[:keep]


def shard_seed(base_seed, rank):
    """independent synthetic shard per rank (BASELINE.json config 5: seeds 0x1234 + g)"""
    return base_seed + rank


def gather_index(torch, dist, local_index, local_count, local_stream_bytes, local_rbsp_bytes, capacity, group=None):
    """All-gather the per-rank NAL indexes.

    local_index: uint8 tensor of `capacity * 32` bytes (hbs_nal_entry records, first
    `local_count` valid) on the rank's device (or CPU for gloo).  Returns
    (all_index [world, capacity*32] uint8, meta [world, 3] int64 = count, stream bytes, rbsp bytes).
    Offsets stay shard-relative; global_entries() rebases them."""
    world = dist.get_world_size(group)
    dev = local_index.device
    meta_local = torch.tensor([local_count, local_stream_bytes, local_rbsp_bytes], dtype=torch.int64, device=dev)
    meta = torch.empty(world * 3, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(meta, meta_local, group=group)
    all_index = torch.empty(world * capacity * ENTRY_BYTES, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(all_index, local_index[: capacity * ENTRY_BYTES].contiguous(), group=group)
    return all_index.view(world, capacity * ENTRY_BYTES), meta.view(world, 3)


class IndexGatherer:
    """The same exchange, pipelined: the gather of one step's index runs on the collective's own
    stream while the next step's scan is already writing the other index buffer.  `depth` results
    are kept in flight (preallocated receive buffers); submit() hands back the slot to pass to
    result() later.  The caller alternates its OWN index buffers likewise, `depth` of them, and calls
    release(buffer_number) BEFORE it enqueues the scan that rewrites a buffer: that makes the current
    stream wait for the gather that still reads it (without it a scan could overwrite -- its prologue
    even clears -- an index the collective is in the middle of sending)."""

    def __init__(self, torch, dist, capacity, device, depth=2, group=None):
        self.torch, self.dist, self.group = torch, dist, group
        self.capacity, self.depth = capacity, depth
        self.world = dist.get_world_size(group)
        self.recv = [torch.empty(self.world * capacity * ENTRY_BYTES, dtype=torch.uint8, device=device) for _ in range(depth)]
        self.meta = [torch.empty(self.world * 3, dtype=torch.int64, device=device) for _ in range(depth)]
        self.meta_local = [torch.empty(3, dtype=torch.int64, device=device) for _ in range(depth)]
        self.work = [None] * depth
        self.step = 0

    def submit(self, local_index, local_count, local_stream_bytes, local_rbsp_bytes):
        slot = self.step % self.depth
        self.step += 1
        self._wait(slot)                                     # the result that used this slot is overwritten now
        t = self.torch
        self.meta_local[slot].copy_(t.tensor([local_count, local_stream_bytes, local_rbsp_bytes], dtype=t.int64), non_blocking=True)
        w0 = self.dist.all_gather_into_tensor(self.meta[slot], self.meta_local[slot], group=self.group, async_op=True)
        w1 = self.dist.all_gather_into_tensor(self.recv[slot], local_index[: self.capacity * ENTRY_BYTES], group=self.group, async_op=True)
        self.work[slot] = (w0, w1)
        return slot

    def release(self, slot):
        """the gather that read the caller's buffer `slot` (slot = step % depth) is done, as far as the current stream is concerned"""
        self._wait(slot)

    def _wait(self, slot):
        if self.work[slot] is not None:
            for w in self.work[slot]:
                w.wait()
            self.work[slot] = None

    def result(self, slot):
        """(all_index [world, capacity*32], meta [world, 3]) of a submitted step; waits for it"""
        self._wait(slot)
        return self.recv[slot].view(self.world, self.capacity * ENTRY_BYTES), self.meta[slot].view(self.world, 3)

    def drain(self):
        for slot in range(self.depth):
            self._wait(slot)


def global_entries(all_index, meta):
    """Host-side view of a gathered index as ONE entry array over the concatenation of the
    shards: start/end shifted by the bytes of the shards in front, rbsp_off by their RBSP bytes."""
    idx = all_index.cpu().numpy()
    m = meta.cpu().numpy()
    out = []
    sbase = rbase = 0
    for r in range(idx.shape[0]):
        n = int(m[r, 0])
        e = idx[r, : n * ENTRY_BYTES].view(NAL_ENTRY).copy()
        e["start"] += np.uint64(sbase)
        e["end"] += np.uint64(sbase)
        e["rbsp_off"] += np.uint64(rbase)
        out.append(e)
        sbase += int(m[r, 1])
        rbase += int(m[r, 2])
    return np.concatenate(out) if out else np.zeros(0, dtype=NAL_ENTRY)
