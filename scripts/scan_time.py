#!/usr/bin/env python3
"""Quick timing of the scan kernels on the bench stream (dev aid): kernel ms by HIP events for
scan + extract and for index only, several repetitions each; prints one JSON line.
    python scripts/scan_time.py [--nals N] [--reps R] [--mode M] [--kernels 0,4,5]
HBS_LIB selects a development build of the library (make variant)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nals", type=int, default=1_677_000)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--mode", type=int, default=0)
    ap.add_argument("--check", type=int, default=1)
    args = ap.parse_args()
    import torch
    import hevcbitstream_amd as hbs
    ctx = hbs.Context(0)
    ctx.enable_timing(True)
    n = args.nals
    g = ctx.synth_stream(0x1234, n, args.mode)
    sb, rb = g["stream_bytes"], g["rbsp_bytes"]
    stream = g["stream"][:sb]
    index, rbsp, summary, cap = ctx.alloc_outputs(sb, index_cap=n + 8, peer=None if os.environ.get("HBS_PLAIN_ALLOC") else stream)
    res = {"lib": os.environ.get("HBS_LIB", "default"), "stream_bytes": sb, "nals": n}
    ks = []
    for i in range(args.reps + 1):
        ctx.index_extract_async(stream, index, cap, rbsp, summary)
        if i:
            ks.append(ctx.kernel_ms())
    s = ctx.read_summary(summary)
    assert int(s["error"]) == 0 and int(s["nal_count"]) == n, s
    if args.check:
        assert torch.equal(rbsp[:rb], g["rbsp"][:rb]), "extracted RBSP != generated RBSP"
        a = index[: n * 32].view(torch.int64).view(n, 4)
        b = g["index"][: n * 32].view(torch.int64).view(n, 4)
        assert torch.equal(a[:, :3], b[:, :3]), "NAL index != generator's index"
    ks.sort()
    algo = sb + rb + 32 * n
    res["extract"] = {"kernel": ctx.last_kernel(), "ms_min": round(ks[0], 4), "ms_med": round(ks[len(ks) // 2], 4),
                      "traffic_TBs_med": round(algo / ks[len(ks) // 2] / 1e9, 3), "frac_med": round(algo / ks[len(ks) // 2] / 1e9 / 8.0, 4)}
    ks = []
    for i in range(args.reps + 1):
        ctx.index_extract_async(stream, index, cap, None, summary)
        if i:
            ks.append(ctx.kernel_ms())
    s = ctx.read_summary(summary)
    assert int(s["error"]) == 0 and int(s["nal_count"]) == n, s
    if args.check:
        a = index[: n * 32].view(torch.int64).view(n, 4)
        assert torch.equal(a[:, :2], b[:, :2]), "index-only NAL index != generator's index"
    ks.sort()
    algo = sb + 32 * n
    res["index_only"] = {"kernel": ctx.last_kernel(), "ms_min": round(ks[0], 4), "ms_med": round(ks[len(ks) // 2], 4),
                         "read_TBs_med": round(algo / ks[len(ks) // 2] / 1e9, 3), "frac_med": round(algo / ks[len(ks) // 2] / 1e9 / 8.0, 4)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
