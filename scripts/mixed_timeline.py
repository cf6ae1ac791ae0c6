#!/usr/bin/env python3
"""dev aid (diagnostic build: `make diag`, HBS_LIB=build/diag/libhbs_diag.so): a timeline per 192 KiB tile of K12 -- taken /
aggregate known / look-back done / finished, wall clock -- on the bench stream and on the mixed stream (bench.py's make_mixed),
with and without the dense tiles counted ahead.  Writes gpurun_out/timeline/<name>.npy ([tiles][4] uint64, 10 ns ticks; bit 0 of
column 1: walked as a dense tile) and prints what the tiles waited for.
    HBS_LIB=build/diag/libhbs_diag.so python scripts/mixed_timeline.py [nals]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    import torch
    import hevcbitstream_amd as hbs
    import bench
    ctx = hbs.Context(0)
    ctx.enable_timing(True)
    ctx.set_kernel(4)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 419_000
    g = ctx.synth_stream(0x1234, n, 0)
    sb = g["stream_bytes"]
    stream = g["stream"][:sb]
    mixed, dense = bench.make_mixed(torch, stream, sb)
    index, rbsp, summary, cap = ctx.alloc_outputs(sb, index_cap=n + 64, peer=stream)
    tiles = (sb + (192 << 10) - 1) // (192 << 10)
    os.makedirs("gpurun_out/timeline", exist_ok=True)
    ctx.lib.hbs_debug_timeline4.argtypes = [C.c_void_p, C.c_uint]
    for name, s in (("uniform", stream), ("mixed", mixed)):
        for mode in (0, 2):
            ctx.set_count_ahead(mode)
            for _ in range(3):
                ctx.index_extract_async(s, index, cap, rbsp, summary)
            ctx.read_summary(summary)
            ms = ctx.kernel_ms()
            tl = np.zeros((tiles, 4), dtype=np.uint64)
            assert ctx.lib.hbs_debug_timeline4(tl.ctypes.data, tiles) == 0
            np.save("gpurun_out/timeline/%s_ahead%d.npy" % (name, mode), tl)
            who = np.zeros(tiles, dtype=np.uint64)
            ctx.lib.hbs_debug_timeline_who4.argtypes = [C.c_void_p, C.c_uint]
            assert ctx.lib.hbs_debug_timeline_who4(who.ctypes.data, tiles) == 0
            np.save("gpurun_out/timeline/%s_ahead%d_who.npy" % (name, mode), who)
            t = tl.astype(np.int64)
            d = (t[:, 1] & 1) == 1
            t0 = t[:, 0].min()
            us = lambda a: a / 100.0
            wait = us(t[:, 2] - t[:, 1])
            print("%s count-ahead %d: kernel %.3f ms, span %.1f us, %d dense tiles; sparse tiles: taken->aggregate %.1f us, look-back %.2f us "
                  "(sum %.0f us), whole %.1f us; dense tiles: taken->aggregate %.1f, look-back %.2f, whole %.1f us" % (
                      name, mode, ms, us(t[:, 3].max() - t0), int(d.sum()),
                      us(t[~d, 1] - t[~d, 0]).mean(), wait[~d].mean(), wait[~d].sum(), us(t[~d, 3] - t[~d, 0]).mean(),
                      us(t[d, 1] - t[d, 0]).mean() if d.any() else 0, wait[d].mean() if d.any() else 0, us(t[d, 3] - t[d, 0]).mean() if d.any() else 0),
                  flush=True)


if __name__ == "__main__":
    main()
