import sys, os, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "scripts"))
import torch, numpy as np
import hevcbitstream_amd as hbs
import nal_sweep as ns
ctx = hbs.Context(0)
for mean in (256, 512, 1024, 1536, 2048):
    arena, rb, idx, n, stream_buf, sb = ns.make_stream(torch, np, ctx, mean, 2 * 2**30)
    stream = stream_buf[:sb]
    index, rbsp, summary, cap = ctx.alloc_outputs(sb, index_cap=n + 64)
    row = {"mean": mean}
    for k in (2, 4):
        ctx.set_kernel(k)
        ms = ns.best_ms(torch, lambda: ctx.index_extract_async(stream, index, cap, rbsp, summary))
        s = ctx.read_summary(summary)
        assert int(s["error"]) == 0 and int(s["nal_count"]) == n
        row["k%d_ms" % k] = round(ms, 3); row["k%d_frac" % k] = round((sb + rb + 32 * n) / ms / 1e6 / 8000, 3)
    print(json.dumps(row), flush=True)
    del arena, idx, stream_buf, stream, index, rbsp
    torch.cuda.empty_cache()
