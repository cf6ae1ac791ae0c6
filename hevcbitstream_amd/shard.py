"""Multi-GPU sharding of the hot path: one process per GPU, each rank indexes
its own independent stream shard; the only exchange is the gather of the NAL
index (counts first, then the padded entry arrays) -- RCCL over xGMI with the
`nccl` backend, `gloo` in the CPU tests.  No collective touches the data path:
stream bytes and RBSP arenas stay on their GPUs."""
import numpy as np

from .api import NAL_ENTRY

ENTRY_BYTES = NAL_ENTRY.itemsize


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
    result() later.  The caller alternates its OWN index buffers likewise: a buffer handed to
    submit() may be written again once the submit() `depth` steps later has returned."""

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
