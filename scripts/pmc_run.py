"""Run K12 a few times on a 1 GiB stream (for rocprofv3 --pmc passes).  usage: pmc_run.py <kernel 2|3> <mode>"""
import sys
import torch
sys.path.insert(0, ".")
import hevcbitstream_amd as hbs
k = int(sys.argv[1]); mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = hbs.Context(0)
ctx.set_kernel(k)
g = ctx.synth_stream(0x1234, 104858, mode)
sb = g["stream_bytes"]
index, rbsp, summary, cap = ctx.alloc_outputs(sb, index_cap=104858 + 8)
for _ in range(3):
    ctx.index_extract_async(g["stream"][:sb], index, cap, rbsp, summary)
torch.cuda.synchronize()
print(ctx.read_summary(summary))
