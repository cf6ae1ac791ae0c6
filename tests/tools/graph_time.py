"""hbs_index_extract captured in a HIP graph against the same call enqueued directly (dev aid): what the
launches of a call cost on streams of tens of MiB.  usage: python3 tests/tools/graph_time.py [reps of 16 MiB ...]"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import hevcbitstream_amd as hbs
from tests import _orc

orc = _orc.oracle()
base, idx, arena = orc.gen_stream(0x1234, 1600, 0)          # ~16 MiB
d1 = torch.from_numpy(base).cuda()
for reps in [int(a) for a in sys.argv[1:]] or [1, 4, 16, 64]:
    d = d1.repeat(reps)
    n = d.numel()
    ctx = hbs.Context(0)
    index, rbsp, summary, cap = ctx.alloc_outputs(n, index_cap=1600 * reps + 16)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            ctx.index_extract_async(d, index, cap, rbsp, summary)
        torch.cuda.synchronize()
        assert int(ctx.read_summary(summary)["nal_count"]) == 1600 * reps

        def timed(fn):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
            ev[0].record()
            for i in range(20):
                fn()
                ev[i + 1].record()
            torch.cuda.synchronize()
            return min(ev[i].elapsed_time(ev[i + 1]) for i in range(20))

        direct = timed(lambda: ctx.index_extract_async(d, index, cap, rbsp, summary))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            ctx.index_extract_async(d, index, cap, rbsp, summary)
        g.replay()
        torch.cuda.synchronize()
        assert int(ctx.read_summary(summary)["nal_count"]) == 1600 * reps
        graph = timed(g.replay)
    print("%5d MiB: direct %.3f ms (%.0f GB/s), graph replay %.3f ms (%.0f GB/s)"
          % (n >> 20, direct, n / direct / 1e6, graph, n / graph / 1e6))
    ctx.close()
