#!/usr/bin/env python3
"""hbs_emit_annexb against the mean NAL size (dev aid; the emit column of scripts/nal_sweep.py by itself, with the pinned paths):
    python scripts/emit_sweep.py --sizes 64,128,192,256,384 [--gib 2]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import hevcbitstream_amd as hbs
import nal_sweep
from hevcbitstream_amd.api import SUMMARY

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="64,128,192,256,384,512")
ap.add_argument("--gib", type=float, default=2.0)
args = ap.parse_args()
ctx = hbs.Context(0)
lib = os.path.basename(os.path.dirname(os.environ.get("HBS_LIB", "/default/x")))
for mean in [int(x) for x in args.sizes.split(",")]:
    arena, rb, idx, n, sbuf, sb = nal_sweep.make_stream(torch, np, ctx, mean, int(args.gib * 2**30))
    out = ctx.pair_alloc(arena, sb + 4096)[0]
    esum = torch.zeros(SUMMARY.itemsize, dtype=torch.uint8, device="cuda")
    idx_out = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
    row = {"lib": lib, "mean": mean, "nals": n}
    for name, io in (("no_index_out", None), ("index_out", idx_out)):
        ms = nal_sweep.best_ms(torch, lambda: ctx.emit_annexb_async(arena, rb, idx, n, 1, out, io, esum))
        es = ctx.read_summary(esum)
        assert int(es["error"]) == 0 and int(es["stream_bytes"]) == sb and torch.equal(out[:sb], sbuf[:sb]), (mean, name, es)
        row[name] = {"ms": round(ms, 4), "traffic_frac": round((sb + rb) / ms / 1e6 / 8000, 4)}
    print(json.dumps(row), flush=True)
    del arena, idx, sbuf, out, idx_out
    torch.cuda.empty_cache()
