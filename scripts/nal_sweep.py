#!/usr/bin/env python3
"""Stream-shape sweep (dev aid and bench.py's other_kernels.nal_size_sweep): throughput of extract (hbs_index_extract with an
arena), index only and emit (hbs_emit_annexb) against the mean NAL size, on streams of random payload with 3- and 4-byte start
codes (gap_mode 1).  Every extract is compared on the device with the LDS-image kernel (whose code path does not depend on the
data), every emitted stream with the stream it came from.
    python scripts/nal_sweep.py [--gib 2] [--sizes 64,512,2048,4096,10240,65536,524288]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HBM_PEAK_GBS = 8000.0


def make_stream(torch, np, ctx, mean, total_bytes, seed=5):
    """arena of random bytes cut into NALs of 0.75-1.25 x mean bytes (first bytes 02 01, last byte 80), index, and K3's stream"""
    from hevcbitstream_amd.api import NAL_ENTRY, SUMMARY
    rng = np.random.RandomState(seed)
    n = max(1, int(total_bytes // mean))
    lens = rng.randint(max(3, int(mean * 0.75)), max(4, int(mean * 1.25)) + 1, size=n).astype(np.int64)
    off = np.concatenate([[0], np.cumsum(lens)[:-1]])
    total = int(lens.sum())
    dev = torch.device("cuda", ctx.device)
    arena = torch.randint(0, 256, (total + 64,), dtype=torch.uint8, device=dev)
    o = torch.from_numpy(off).to(dev)
    l = torch.from_numpy(lens).to(dev)
    arena[o] = 0x02
    arena[o + 1] = 0x01
    arena[o + l - 1] = 0x80
    ent = np.zeros(n, dtype=NAL_ENTRY)
    ent["rbsp_off"] = off
    ent["rbsp_len"] = lens
    idx = torch.from_numpy(ent.view(np.uint8).copy()).to(dev)
    cap_out = int(ctx.lib.hbs_annexb_bound(total, n))
    stream = torch.empty(cap_out, dtype=torch.uint8, device=dev)
    idx_out = torch.empty(n * 32, dtype=torch.uint8, device=dev)
    summ = torch.zeros(SUMMARY.itemsize, dtype=torch.uint8, device=dev)
    ctx.emit_annexb_async(arena, total, idx, n, 1, stream, idx_out, summ)
    s = ctx.read_summary(summ)
    assert int(s["error"]) == 0, s
    return arena, total, idx, n, stream, int(s["stream_bytes"])


def best_ms(torch, fn, reps=4):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    fn()
    for i in range(reps + 1):
        ev[i].record()
        if i < reps:
            fn()
    torch.cuda.synchronize()
    return min(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))


def sweep(torch, hbs, ctx, sizes, gib, check=True):
    import numpy as np
    from hevcbitstream_amd.api import SUMMARY
    rows = []
    for mean in sizes:
        arena, rb, idx, n, stream_buf, sb = make_stream(torch, np, ctx, mean, int(gib * 2**30))
        stream = stream_buf[:sb]
        plain = bool(os.environ.get("HBS_PLAIN_ALLOC"))          # default: outputs placed against their inputs (hbs_pair_alloc)
        index, rbsp, summary, cap = ctx.alloc_outputs(sb, index_cap=n + 64, peer=None if plain else stream)
        ctx.set_kernel(0)
        ms_x = best_ms(torch, lambda: ctx.index_extract_async(stream, index, cap, rbsp, summary))
        s = ctx.read_summary(summary)
        kern = ctx.last_kernel()
        assert int(s["error"]) == 0 and int(s["nal_count"]) == n and int(s["rbsp_bytes"]) == rb, (mean, s)
        assert torch.equal(rbsp[:rb], arena[:rb]), "mean %d: extracted RBSP != the arena the stream was made from" % mean
        if check and kern != 2:
            index2, rbsp2, summary2, _ = ctx.alloc_outputs(sb, index_cap=n + 64)
            ctx.set_kernel(2)
            ctx.index_extract_async(stream, index2, cap, rbsp2, summary2)
            s2 = ctx.read_summary(summary2)
            ctx.set_kernel(0)
            assert int(s2["nal_count"]) == n and torch.equal(index[: n * 32], index2[: n * 32]), "mean %d: index differs from the LDS-image kernel's" % mean
            del index2, rbsp2
        index_b = torch.empty_like(index)
        ms_i = best_ms(torch, lambda: ctx.index_extract_async(stream, index_b, cap, None, summary))
        s = ctx.read_summary(summary)
        kern_i = ctx.last_kernel()
        a = index[: n * 32].view(torch.int64).view(n, 4)
        b = index_b[: n * 32].view(torch.int64).view(n, 4)
        assert int(s["nal_count"]) == n and torch.equal(a[:, :2], b[:, :2]), "mean %d: index-only start / end differ" % mean
        del index_b
        out = torch.empty(sb + 4096, dtype=torch.uint8, device=stream.device) if plain else ctx.pair_alloc(rbsp, sb + 4096)[0]
        esum = torch.zeros(SUMMARY.itemsize, dtype=torch.uint8, device=stream.device)
        ctx.set_emit_path(-1)
        ms_e = best_ms(torch, lambda: ctx.emit_annexb_async(rbsp, rb, index, n, 0, out, None, esum))
        es = ctx.read_summary(esum)
        assert int(es["error"]) == 0 and int(es["stream_bytes"]) == sb and torch.equal(out[:sb], stream), "mean %d: emitted stream != the stream" % mean
        rows.append({"mean_nal_bytes": mean, "nals": n, "stream_GiB": round(sb / 2**30, 3),
                     "extract": {"kernel": kern, "ms": round(ms_x, 4), "GBs_scanned": round(sb / ms_x / 1e6, 1),
                                 "traffic_frac": round((sb + rb + 32 * n) / ms_x / 1e6 / HBM_PEAK_GBS, 4)},
                     "index_only": {"kernel": kern_i, "ms": round(ms_i, 4), "GBs_scanned": round(sb / ms_i / 1e6, 1),
                                    "read_frac": round((sb + 32 * n) / ms_i / 1e6 / HBM_PEAK_GBS, 4)},
                     "emit": {"ms": round(ms_e, 4), "GBs_emitted": round(sb / ms_e / 1e6, 1),
                              "traffic_frac": round((sb + rb) / ms_e / 1e6 / HBM_PEAK_GBS, 4)}})
        del arena, idx, stream_buf, stream, index, rbsp, out
        torch.cuda.empty_cache()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=2.0)
    ap.add_argument("--sizes", default="64,512,2048,4096,10240,65536,524288")
    args = ap.parse_args()
    import torch
    import hevcbitstream_amd as hbs
    ctx = hbs.Context(0)
    for r in sweep(torch, hbs, ctx, [int(x) for x in args.sizes.split(",")], args.gib):
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
