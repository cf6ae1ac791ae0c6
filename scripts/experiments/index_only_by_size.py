import os, sys, json
sys.path.insert(0, ".")
import torch
import hevcbitstream_amd as hbs
ctx = hbs.Context(0); ctx.enable_timing(True)
for nals in (13107, 26214, 52428, 78643, 104857, 209715):
    g = ctx.synth_stream(0x1234, nals, 0); sb = g["stream_bytes"]; stream = g["stream"][:sb]
    index, rbsp, summary, cap = ctx.alloc_outputs(sb, index_cap=nals + 8)
    row = {"MiB": sb >> 20}
    for k in (4, 5):
        ctx.set_kernel(k)
        ks = []
        for i in range(7):
            ctx.index_extract_async(stream, index, cap, None, summary)
            if i: ks.append(ctx.kernel_ms())
        s = ctx.read_summary(summary); assert int(s["nal_count"]) == nals
        row["k%d_ms" % k] = round(sorted(ks)[len(ks)//2], 4)
    print(json.dumps(row))
