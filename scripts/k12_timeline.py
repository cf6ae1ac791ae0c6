#!/usr/bin/env python3
"""dev aid (diagnostic build: `make diag`, HBS_LIB=build/diag/libhbs_diag.so): the per-tile timeline of K12 on one synthetic
stream -- taken / aggregate known / look-back done / finished, wall clock, 10 ns ticks -- summarised: how long a tile's phases
take, how long tiles wait in their look-back and for whom (the tile whose aggregate came last among the 512 in front).
    HBS_LIB=build/diag/libhbs_diag.so python scripts/k12_timeline.py [nals] [mode: 0 uniform, 1 zero-heavy] [mean-nal-bytes]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    import torch
    import hevcbitstream_amd as hbs
    ctx = hbs.Context(0)
    ctx.enable_timing(True)
    ctx.set_kernel(4)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 419_000
    mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    if len(sys.argv) > 3:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import nal_sweep
        _, _, _, n, sbuf, sb = nal_sweep.make_stream(torch, np, ctx, int(sys.argv[3]), 2 << 30)
        stream = sbuf[:sb]
    else:
        g = ctx.synth_stream(0x1234, n, mode)
        sb = g["stream_bytes"]
        stream = g["stream"][:sb]
    index, rbsp, summary, cap = ctx.alloc_outputs(sb, index_cap=n + 64, peer=stream)
    tiles = (sb + (192 << 10) - 1) // (192 << 10)
    for _ in range(3):
        ctx.index_extract_async(stream, index, cap, rbsp, summary)
    s = ctx.read_summary(summary)
    assert int(s["error"]) == 0
    ms = ctx.kernel_ms()
    tl = np.zeros((tiles, 4), dtype=np.uint64)
    ctx.lib.hbs_debug_timeline4.argtypes = [C.c_void_p, C.c_uint]
    assert ctx.lib.hbs_debug_timeline4(tl.ctypes.data, tiles) == 0
    os.makedirs("gpurun_out/timeline", exist_ok=True)
    np.save("gpurun_out/timeline/k12_mode%d_%d.npy" % (mode, n), tl)
    t = tl.astype(np.int64)
    d = (t[:, 1] & 1) == 1
    u = (t - t[:, 0].min()) / 100.0
    a, w, c = u[:, 1] - u[:, 0], u[:, 2] - u[:, 1], u[:, 3] - u[:, 2]
    q = lambda x: "mean %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f" % (x.mean(), *np.percentile(x, [10, 50, 90, 99]))
    print("stream %.2f GiB, %d tiles (%d dense), kernel %.3f ms (diagnostic build)" % (sb / 2**30, tiles, int(d.sum()), ms))
    print("  taken -> aggregate known  us: " + q(a))
    print("  look-back wait            us: " + q(w))
    print("  aggregate -> finished     us: " + q(c))
    print("  whole tile                us: " + q(u[:, 3] - u[:, 0]))
    # who a tile waited for: the tile with the latest aggregate among the 512 in front of it
    waited = np.zeros(tiles)
    for i in range(1, tiles):
        lo = max(0, i - 512)
        waited[i] = max(0.0, u[lo:i, 1].max() - u[i, 1])
    print("  aggregate of the slowest tile in front, behind my own  us: " + q(waited))


if __name__ == "__main__":
    main()
