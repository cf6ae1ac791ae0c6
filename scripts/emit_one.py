#!/usr/bin/env python3
"""hbs_emit_annexb on ONE 2 GiB arena of NALs of --mean bytes, six calls (dev aid: the command rocprofv3 --kernel-trace --stats is run over)"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import hevcbitstream_amd as hbs
import nal_sweep
from hevcbitstream_amd.api import SUMMARY
ap = argparse.ArgumentParser(); ap.add_argument("--mean", type=int, default=64); args = ap.parse_args()
ctx = hbs.Context(0)
arena, rb, idx, n, sbuf, sb = nal_sweep.make_stream(torch, np, ctx, args.mean, 2 << 30)
out = torch.empty(sb + 4096, dtype=torch.uint8, device="cuda")
esum = torch.zeros(SUMMARY.itemsize, dtype=torch.uint8, device="cuda")
ms = nal_sweep.best_ms(torch, lambda: ctx.emit_annexb_async(arena, rb, idx, n, 1, out, None, esum))
assert torch.equal(out[:sb], sbuf[:sb])
print(json.dumps({"mean": args.mean, "ms": round(ms, 4), "traffic_frac": round((sb + rb) / ms / 1e6 / 8000, 4)}))
