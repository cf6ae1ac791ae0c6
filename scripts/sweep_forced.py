#!/usr/bin/env python3
"""NAL-size sweep with the scan kernel pinned (dev aid): extract and index-only per kernel, checked against the arena the stream
was made from and the automatic mode's index.   HBS_LIB=... python scripts/sweep_forced.py --sizes 128,256,384 --kernels 0,4"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import hevcbitstream_amd as hbs
import nal_sweep

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="128,192,256,384,512,1024")
ap.add_argument("--kernels", default="0,4")
ap.add_argument("--gib", type=float, default=2.0)
args = ap.parse_args()
ctx = hbs.Context(0)
ctx.enable_timing(True)
lib = os.path.basename(os.path.dirname(os.environ.get("HBS_LIB", "/default/x")))
for mean in [int(x) for x in args.sizes.split(",")]:
    arena, rb, idx, n, sbuf, sb = nal_sweep.make_stream(torch, np, ctx, mean, int(args.gib * 2**30))
    stream = sbuf[:sb]
    index, rbsp, summary, cap = ctx.alloc_outputs(sb, index_cap=n + 64, peer=stream)
    ref = None
    row = {"lib": lib, "mean": mean, "nals": n}
    for k in [int(x) for x in args.kernels.split(",")]:
        ctx.set_kernel(k)
        ks = []
        for i in range(5):
            ctx.index_extract_async(stream, index, cap, rbsp, summary)
            if i:
                ks.append(ctx.kernel_ms())
        s = ctx.read_summary(summary)
        assert int(s["error"]) == 0 and int(s["nal_count"]) == n and int(s["rbsp_bytes"]) == rb, (mean, k, s)
        assert torch.equal(rbsp[:rb], arena[:rb]), (mean, k)
        if ref is None:
            ref = index[: n * 32].clone()
        else:
            assert torch.equal(ref, index[: n * 32]), (mean, k)
        ks.sort()
        ms = ks[len(ks) // 2]
        ki = []
        index_b = torch.empty_like(index)
        for i in range(5):
            ctx.index_extract_async(stream, index_b, cap, None, summary)
            if i:
                ki.append(ctx.kernel_ms())
        s = ctx.read_summary(summary)
        a = ref.view(torch.int64).view(n, 4)
        b = index_b[: n * 32].view(torch.int64).view(n, 4)
        assert int(s["nal_count"]) == n and torch.equal(a[:, :2], b[:, :2]), (mean, k)
        ki.sort()
        msi = ki[len(ki) // 2]
        row["k%d" % k] = {"ran": ctx.last_kernel(), "extract_ms": round(ms, 4), "extract_frac": round((sb + rb + 32 * n) / ms / 1e6 / 8000, 4),
                          "index_ms": round(msi, 4), "index_frac": round((sb + 32 * n) / msi / 1e6 / 8000, 4)}
        del index_b
    print(json.dumps(row), flush=True)
    del arena, idx, sbuf, stream, index, rbsp, ref
    torch.cuda.empty_cache()
