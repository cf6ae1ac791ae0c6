#!/usr/bin/env python3
"""K12's time against the placement of the INDEX buffer alone (stream and arena fixed): sub-tensors of one pool at different
offsets.  scripts/experiments/placement_probe.py had shown 5.95 ... 6.17 ms on one stream / arena with five index addresses."""
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
index, rbsp, summary, cap = ctx.alloc_outputs(sb, index_cap=n + 8)
need = index.numel()
pool = torch.empty(need + (160 << 20), dtype=torch.uint8, device="cuda")
offs = [0, 256, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 4 << 20, 6 << 20, 8 << 20, 12 << 20, 16 << 20, 24 << 20, 32 << 20, 48 << 20, 64 << 20, 96 << 20, 128 << 20]
mode = sys.argv[1] if len(sys.argv) > 1 else "index"
if mode == "rbsp":
    need = rbsp.numel()
    pool = torch.empty(need + (160 << 20), dtype=torch.uint8, device="cuda")
for rep in range(2):
    for off in offs:
        sub = pool[off: off + need]
        ks = []
        for i in range(4):
            if mode == "index":
                ctx.index_extract_async(stream, sub, cap, rbsp, summary)
            else:
                ctx.index_extract_async(stream, index, cap, sub, summary)
            if i:
                ks.append(ctx.kernel_ms())
        print(json.dumps({"what": mode, "offset": off, "addr": hex(sub.data_ptr()), "kernel_ms": round(sorted(ks)[1], 3)}))
