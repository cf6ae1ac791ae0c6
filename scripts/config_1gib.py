#!/usr/bin/env python3
"""BASELINE's own 1 GiB configs by themselves (bench.py's other_kernels.configs_1GiB: config 2 = scan + index + RBSP
extraction and the scan alone, config 4 = re-emission), without the host-side comparison: the command the rocprofv3 kernel
statistics of profiles/r05/ are taken over.  HBS_LIB selects a development build (make variant).
    python3 scripts/config_1gib.py [--reps R]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=12)
    ap.add_argument("--exclusive", type=int, default=1, help="hbs_ctx_set_device_exclusive (bench.py sets 1: one process per GPU)")
    args = ap.parse_args()
    import torch
    import hevcbitstream_amd as hbs
    import bench
    ctx = hbs.Context(0)
    ctx.enable_timing(True)
    if args.exclusive:
        ctx.set_device_exclusive(1)
    res = bench.configs_1gib(torch, hbs, ctx, check=False, reps=args.reps)
    res["lib"] = os.environ.get("HBS_LIB", "default")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
