"""One rank of tests/test_gpu_shard.py::test_two_ranks_share_one_gpu: the C ABI's index gather with world > 1 on a box with
ONE GPU.  RCCL refuses two ranks on one device, so the library is pointed (HBS_RCCL_LIB) at tests/sim/libfake_rccl.so, a
shared-memory stand-in that also reports what real RCCL would answer with a hang (a send nobody receives, a rank that left).
usage: shard_worker.py RANK WORLD STREAM.npy [STREAM.npy ...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hevcbitstream_amd as hbs           # noqa: E402
from hevcbitstream_amd import shard       # noqa: E402
from tests import _orc                    # noqa: E402

rank, world = int(sys.argv[1]), int(sys.argv[2])
orc = _orc.oracle()
ctx = hbs.Context(0)
comm = shard.LibraryComm(ctx, None, rank, world, ident=b"\x5a" * 128)
assert comm.world_seen() == world
assert comm.reserve_hint() == min(64, max(32, 8 * (world - 1))), comm.reserve_hint()      # slots for the exchange beside the scan: a function of the world
E = shard.ENTRY_BYTES


def entries(n, tag):
    e = np.zeros(n, dtype=hbs.NAL_ENTRY)
    e["start"] = np.arange(n) * 10 + tag
    e["end"] = e["start"] + 7
    e["rbsp_off"] = np.arange(n) * 7
    e["rbsp_len"] = 7
    return e


def dev(e):
    return torch.from_numpy(np.ascontiguousarray(e).view(np.uint8).copy()).cuda() if len(e) else torch.zeros(E, dtype=torch.uint8, device="cuda")


# 1. independent shards of different sizes (one of them empty), to everybody and to each root
sizes = [1000, 0, 37, 5, 0, 211, 64, 3][:world] if world <= 8 else [100 + r for r in range(world)]      # (world 8: two empty ranks, seven peers per group)
mine = entries(sizes[rank], 1000 * rank)
want_all = np.concatenate([entries(sizes[r], 1000 * r) for r in range(world)])
for root in [-1] + list(range(world)):
    recv = root < 0 or root == rank
    all_index = torch.zeros((sum(sizes) + 3) * E, dtype=torch.uint8, device="cuda") if recv else None
    counts = comm.gather_index(dev(mine), len(mine), all_index, root=root)
    torch.cuda.synchronize()
    assert counts == sizes, (root, counts)
    if recv:
        got = all_index[: sum(sizes) * E].cpu().numpy().view(hbs.NAL_ENTRY)
        assert np.array_equal(got, want_all), root

# 2. capacity: ONE receiver's buffer is too small -> EVERY rank gets HBS_E_CAPACITY, nobody is left in a collective, and the
#    communicator works afterwards (round 2's advice: the receiver used to return alone and the senders hung)
for root, small_rank in [(-1, world - 1), (0, 0)]:
    recv = root < 0 or root == rank
    cap = sum(sizes) - 1 if rank == small_rank else sum(sizes) + 3
    all_index = torch.zeros(cap * E, dtype=torch.uint8, device="cuda") if recv else None
    try:
        comm.gather_index(dev(mine), len(mine), all_index, root=root)
        raise SystemExit("rank %d: no error although rank %d's buffer is too small" % (rank, small_rank))
    except hbs.HbsError as e:
        assert e.code == -4, e                    # HBS_E_CAPACITY, on this rank too
    all_index = torch.zeros((sum(sizes) + 3) * E, dtype=torch.uint8, device="cuda") if recv else None
    assert comm.gather_index(dev(mine), len(mine), all_index, root=root) == sizes

# 3. ONE stream in `world` parts -- without, and with an empty NAL (00 00 01 00 00 01) in one part or another: the walk of the
#    whole stream stops there (hevc_analyze.c:135), and so must the gathered index (hbs_gather_parts)
for path in sys.argv[3:]:
    host = np.load(path)
    want = orc.index_stream(host)[0]
    parts = shard.part_ranges(host, world)
    lo, hi, hi_halo = parts[rank]
    d = torch.from_numpy(host[lo:hi_halo].copy()).cuda() if hi_halo > lo else torch.zeros(16, dtype=torch.uint8, device="cuda")[:0]
    index, rbsp, summary, cap = ctx.alloc_outputs(max(hi_halo - lo, 16))
    ctx.index_extract_async(d, index, cap, rbsp, summary)
    s = ctx.read_summary(summary)
    n, stopped = int(s["nal_count"]), int(s["stop_reason"]) == 1
    ent = index[: n * E].cpu().numpy().view(hbs.NAL_ENTRY)
    if not stopped:
        ent = shard.trim_part(ent, hi - lo, rank == world - 1)
    all_index = torch.zeros((len(want) + 8) * E, dtype=torch.uint8, device="cuda")
    counts = comm.gather_index(index, len(ent), all_index, stream_base=lo, root=-1, stopped=stopped)
    torch.cuda.synchronize()
    got = all_index[: sum(counts) * E].cpu().numpy().view(hbs.NAL_ENTRY)
    assert sum(counts) == len(want), (path, counts, len(want))
    assert np.array_equal(got["start"], want["start"]) and np.array_equal(got["end"], want["end"]), path
comm.close()
print("rank %d ok" % rank)
