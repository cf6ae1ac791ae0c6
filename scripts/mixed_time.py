#!/usr/bin/env python3
"""A stream that is sparse on average with dense regions the density probe does not see (dev aid, and the `mixed_stream`
line of bench.py): the uniform bench stream with 1 % of its bytes overwritten by `00 00 03` padding (cabac_zero_words-like
runs, valid inside a NAL) in regions of 640 KiB placed midway BETWEEN the probe's 64 sample windows.  Times the automatic
mode on it and on the unmodified stream; checks the outputs against the LDS-image kernel (2), whose cost does not depend on the
data, entry by entry and byte by byte on the device.
    python scripts/mixed_time.py [--nals N] [--region-kib K] [--percent P]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


from bench import make_mixed          # the generator lives with the benchmark that reports it


def run(ctx, torch, stream, sb, cap, reps=5, want_rbsp=True):
    index, rbsp, summary, _ = ctx.alloc_outputs(sb, index_cap=cap, want_rbsp=want_rbsp)
    ks = []
    for i in range(reps + 1):
        ctx.index_extract_async(stream, index, cap, rbsp, summary)
        if i:
            ks.append(ctx.kernel_ms())
    s = ctx.read_summary(summary)
    ks.sort()
    return index, rbsp, s, ks[len(ks) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nals", type=int, default=419000)
    ap.add_argument("--region-kib", type=int, default=640)
    ap.add_argument("--percent", type=float, default=1.0)
    args = ap.parse_args()
    import torch
    import hevcbitstream_amd as hbs
    ctx = hbs.Context(0)
    ctx.enable_timing(True)
    g = ctx.synth_stream(0x1234, args.nals, 0)
    sb = g["stream_bytes"]
    stream = g["stream"][:sb]
    mixed, dense_bytes = make_mixed(torch, stream, sb, args.percent, args.region_kib << 10)
    cap = args.nals + 64
    ctx.set_kernel(0)
    _, _, s_u, ms_u = run(ctx, torch, stream, sb, cap)
    k_u = ctx.last_kernel()
    idx_m, rbsp_m, s_m, ms_m = run(ctx, torch, mixed, sb, cap)
    k_m = ctx.last_kernel()
    # index only (no arena): the streaming kernel from 0.75 GiB up
    _, _, s_iu, ms_iu = run(ctx, torch, stream, sb, cap, want_rbsp=False)
    k_iu = ctx.last_kernel()
    idx_im, _, s_im, ms_im = run(ctx, torch, mixed, sb, cap, reps=2, want_rbsp=False)
    k_im = ctx.last_kernel()
    ctx.set_kernel(2)
    idx_2, rbsp_2, s_2, ms_2 = run(ctx, torch, mixed, sb, cap, reps=2)
    ctx.set_kernel(0)
    assert int(s_m["error"]) == 0 and int(s_2["error"]) == 0
    n = int(s_2["nal_count"])
    assert int(s_m["nal_count"]) == n and int(s_m["rbsp_bytes"]) == int(s_2["rbsp_bytes"])
    assert torch.equal(idx_m[: n * 32], idx_2[: n * 32]), "mixed stream: index differs from the LDS-image kernel's"
    a = idx_im[: n * 32].view(torch.int64).view(n, 4)
    b = idx_2[: n * 32].view(torch.int64).view(n, 4)
    assert int(s_im["nal_count"]) == n and torch.equal(a[:, :2], b[:, :2]), "mixed stream, index only: entries differ"
    rb = int(s_2["rbsp_bytes"])
    assert torch.equal(rbsp_m[:rb], rbsp_2[:rb]), "mixed stream: RBSP differs from the LDS-image kernel's"
    print(json.dumps({"stream_bytes": sb, "dense_bytes": dense_bytes, "dense_percent": round(100.0 * dense_bytes / sb, 3),
                      "uniform": {"kernel": k_u, "ms": round(ms_u, 4), "GBs": round(sb / ms_u / 1e6, 1)},
                      "mixed": {"kernel": k_m, "ms": round(ms_m, 4), "GBs": round(sb / ms_m / 1e6, 1), "nals": n},
                      "mixed_over_uniform": round(ms_m / ms_u, 3),
                      "index_only": {"uniform": {"kernel": k_iu, "ms": round(ms_iu, 4)}, "mixed": {"kernel": k_im, "ms": round(ms_im, 4)},
                                     "mixed_over_uniform": round(ms_im / ms_iu, 3)},
                      "lds_image_kernel_on_mixed": {"ms": round(ms_2, 4), "GBs": round(sb / ms_2 / 1e6, 1)}}))


if __name__ == "__main__":
    main()
