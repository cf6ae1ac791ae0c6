import os, sys, json
sys.path.insert(0, ".")
import torch
import hevcbitstream_amd as hbs
ctx = hbs.Context(0); ctx.enable_timing(True)
n = 1_677_000
g = ctx.synth_stream(0x1234, n, 0); sb = g["stream_bytes"]; stream = g["stream"][:sb]
index, rbsp, summary, cap = ctx.alloc_outputs(sb, index_cap=n + 8)
ks = []
for i in range(6):
    ctx.index_extract_async(stream, index, cap, rbsp, summary)
    if i: ks.append(ctx.kernel_ms())
torch.cuda.synchronize()
print(os.environ.get("HBS_LIB", "default")[-20:], [round(k, 3) for k in ks])
