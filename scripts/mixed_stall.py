#!/usr/bin/env python3
"""dev aid (diagnostic build: make diag; HBS_LIB=build/diag/libhbs_diag.so): how much of the mixed stream's penalty is WAITING?
K12 on the bench stream and on the same stream with 1 % of it in 640 KiB stretches of 00 00 03 padding (bench.py's make_mixed),
with the real look-back and with none at all (hbs_debug_fake_lb4: wrong results, timing only -- nobody waits for anybody)."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import hevcbitstream_amd as hbs
    import bench
    ctx = hbs.Context(0)
    ctx.enable_timing(True)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_677_000
    g = ctx.synth_stream(0x1234, n, 0)
    sb = g["stream_bytes"]
    stream = g["stream"][:sb]
    mixed, dense = bench.make_mixed(torch, stream, sb)
    index, rbsp, summary, cap = ctx.alloc_outputs(sb, index_cap=n + 64, peer=stream)
    res = {"stream_bytes": sb, "dense_bytes": dense}
    for fake in (0, 1):
        assert ctx.lib.hbs_debug_fake_lb4(C.c_int(fake)) == 0
        for name, s in (("uniform", stream), ("mixed", mixed)):
            ks = []
            for i in range(5):
                ctx.index_extract_async(s, index, cap, rbsp, summary)
                if i:
                    ks.append(ctx.kernel_ms())
            ctx.read_summary(summary)
            ks.sort()
            res["%s_%s" % (name, "no_lookback" if fake else "real")] = round(ks[len(ks) // 2], 4)
    ctx.lib.hbs_debug_fake_lb4(C.c_int(0))
    res["mixed_over_uniform_real"] = round(res["mixed_real"] / res["uniform_real"], 3)
    res["mixed_over_uniform_no_lookback"] = round(res["mixed_no_lookback"] / res["uniform_no_lookback"], 3)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
