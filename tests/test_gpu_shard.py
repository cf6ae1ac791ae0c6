"""The index exchange of the N > 1 path on real RCCL: this pool has one GPU per box, so a one-rank `nccl` group
is the most that runs here -- it still puts the collectives on RCCL's stream next to the scan on the context's
stream, which is what the pipelined gatherer has to get right (the 2-rank semantics are covered by the gloo test)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_pipelined_gather_on_rccl_one_rank():
    import torch
    import torch.distributed as dist
    import hevcbitstream_amd as hbs
    from hevcbitstream_amd import shard
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        ctx = hbs.Context(0)
        n = 4000
        g = ctx.synth_stream(0x4321, n, 0)
        sb, rb = g["stream_bytes"], g["rbsp_bytes"]
        stream = g["stream"][:sb]
        index, rbsp, summary, cap = ctx.alloc_outputs(sb, index_cap=n + 8)
        indexes = [index, torch.empty_like(index)]
        gat = shard.IndexGatherer(torch, dist, cap, index.device, depth=2)
        slots = []
        for step in range(5):                                  # scans and gathers in flight together
            buf = indexes[step % 2]
            ctx.index_extract_async(stream, buf, cap, rbsp, summary)
            slots.append(gat.submit(buf, n, sb, rb))
        all_index, meta = gat.result(slots[-1])
        gat.drain()
        torch.cuda.synchronize()
        assert meta.cpu().tolist() == [[n, sb, rb]]
        got = all_index[0, : n * 32].view(torch.int64).view(n, 4)
        want = g["index"][: n * 32].view(torch.int64).view(n, 4)
        assert torch.equal(got[:, :3], want[:, :3])               # start, end, rbsp_off as the generator laid them out
        glob = shard.global_entries(all_index, meta)
        assert len(glob) == n and int(glob["end"][-1]) == sb
        ctx.close()
    finally:
        dist.destroy_process_group()
