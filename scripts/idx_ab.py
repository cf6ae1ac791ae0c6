#!/usr/bin/env python3
"""Index-only scan A/B (dev aid): one process = one library (HBS_LIB), kernel ms by the library's events on
  the 16 GiB bench stream, its mixed form, a 1 GiB stream, and 2 GiB streams of several mean NAL sizes.
    HBS_LIB=build/variants/x/libhbs.so python scripts/idx_ab.py [--sizes 512,1024,...] [--no16 1]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import hevcbitstream_amd as hbs
import nal_sweep

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="512,1024,2048,4096,10240")
ap.add_argument("--no16", type=int, default=0)
ap.add_argument("--reps", type=int, default=6)
args = ap.parse_args()
ctx = hbs.Context(0)
ctx.enable_timing(True)
lib = os.path.basename(os.path.dirname(os.environ.get("HBS_LIB", "/default/x")))
tag = {k: os.environ[k] for k in ("HBS5_FORCE_TICKET", "HBS5_TILE_ROWS") if k in os.environ}


def run(stream, sb, n_cap, reps=args.reps):
    index = torch.empty(n_cap * 32, dtype=torch.uint8, device="cuda")
    summary = torch.zeros(64, dtype=torch.uint8, device="cuda")
    ks = []
    for i in range(reps + 1):
        ctx.index_extract_async(stream, index, n_cap, None, summary)
        if i:
            ks.append(ctx.kernel_ms())
    s = ctx.read_summary(summary)
    assert int(s["error"]) == 0, s
    ks.sort()
    return ks[0], ks[len(ks) // 2], int(s["nal_count"]), index


out = {"lib": lib, "env": tag}
if not args.no16:
    n = 1_677_000
    g = ctx.synth_stream(0x1234, n, 0)
    sb = g["stream_bytes"]
    stream = g["stream"][:sb]
    lo, med, cnt, index = run(stream, sb, n + 8)
    assert cnt == n
    a = index[: n * 32].view(torch.int64).view(n, 4)
    b = g["index"][: n * 32].view(torch.int64).view(n, 4)
    assert torch.equal(a[:, :2], b[:, :2])
    out["16GiB"] = {"ms_min": round(lo, 4), "ms_med": round(med, 4), "read_frac": round((sb + 32 * n) / med / 1e6 / 8000, 4)}
    del index, a, b
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    mixed, _ = bench.make_mixed(torch, stream, sb)
    lo2, med2, cnt2, index = run(mixed, sb, n + 64, reps=4)
    out["16GiB_mixed"] = {"ms_med": round(med2, 4), "over_uniform": round(med2 / med, 3)}
    del mixed, index
    # 1 GiB prefix-like stream of its own
    g1 = ctx.synth_stream(0x1234, 104_858, 0)
    sb1 = g1["stream_bytes"]
    lo, med, cnt, index = run(g1["stream"][:sb1], sb1, 104_858 + 8, reps=10)
    assert cnt == 104_858
    out["1GiB"] = {"ms_min": round(lo, 4), "ms_med": round(med, 4), "read_frac": round((sb1 + 32 * cnt) / med / 1e6 / 8000, 4)}
    del g, g1, stream, index
    torch.cuda.empty_cache()
for mean in [int(x) for x in args.sizes.split(",") if x]:
    _, _, _, n, sbuf, sb = nal_sweep.make_stream(torch, np, ctx, mean, 2 << 30)
    lo, med, cnt, index = run(sbuf[:sb], sb, n + 64)
    assert cnt == n, (cnt, n)
    out["2GiB_%d" % mean] = {"ms_min": round(lo, 4), "ms_med": round(med, 4), "read_frac": round((sb + 32 * n) / lo / 1e6 / 8000, 4), "kernel": ctx.last_kernel()}
    del sbuf, index
    torch.cuda.empty_cache()
print(json.dumps(out), flush=True)
