#!/usr/bin/env python3
"""Is K12's run-to-run spread (5.93 / 6.08 / 6.23 ms per 16 GiB, constant within a process) a matter of where the buffers lie?
One process, one stream; the outputs are allocated again and again (with spacers of different sizes in between, kept alive, so
that the allocator hands out other addresses) and the kernel timed on each placement.  Dev aid (round 3)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hevcbitstream_amd as hbs
ctx = hbs.Context(0)
ctx.enable_timing(True)
n = 1_677_000
g = ctx.synth_stream(0x1234, n, 0)
sb = g["stream_bytes"]
stream = g["stream"][:sb]
del g["rbsp"], g["index"]
torch.cuda.empty_cache()
spacers = []
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    index, rbsp, summary, cap = ctx.alloc_outputs(sb, index_cap=n + 8)
    ks = []
    for i in range(5):
        ctx.index_extract_async(stream, index, cap, rbsp, summary)
        if i:
            ks.append(ctx.kernel_ms())
    print(json.dumps({"trial": trial, "stream": hex(stream.data_ptr()), "rbsp": hex(rbsp.data_ptr()), "index": hex(index.data_ptr()),
                      "rbsp_minus_stream_MiB": (rbsp.data_ptr() - stream.data_ptr()) / 2**20, "kernel_ms": [round(k, 3) for k in ks]}))
    del index, rbsp, summary
    spacers.append(torch.empty((trial * 37 + 5) << 20, dtype=torch.uint8, device="cuda"))     # shifts the next allocation
    torch.cuda.empty_cache()
