#!/usr/bin/env python3
"""dev aid: K12's time against the NUMBER OF TILES in one process, on prefixes of one stream into one arena (same physical memory
for every size): is the fixed cost of a call a whole-round effect of the 512 persistent workgroups?
    python3 scripts/k12_rounds.py [first_tiles last_tiles step]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import hevcbitstream_amd as hbs
    lo, hi, step = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (4608, 6656, 128)
    ctx = hbs.Context(0)
    ctx.enable_timing(True)
    tile = 192 * 1024
    n = (hi * tile) // 10200 + 2000
    g = ctx.synth_stream(0x1234, n, 0)
    sb = g["stream_bytes"]
    assert sb >= hi * tile, (sb, hi * tile)
    stream = g["stream"][:sb]
    index, rbsp, summary, cap = ctx.alloc_outputs(sb, index_cap=n + 8, peer=stream)
    rows = []
    for tiles in range(lo, hi + 1, step):
        nbytes = tiles * tile - 4096                     # a little under a whole number of tiles
        ks = []
        for i in range(9):
            ctx.index_extract_async(stream[:nbytes], index, cap, rbsp, summary)
            if i:
                ks.append(ctx.kernel_ms())
        s = ctx.read_summary(summary)
        assert int(s["error"]) == 0
        ks.sort()
        rows.append({"tiles": tiles, "rounds": round(tiles / 512.0, 2), "ms_med": round(ks[len(ks) // 2], 4), "ms_min": round(ks[0], 4),
                     "us_per_tile": round(ks[len(ks) // 2] * 1e3 / tiles * 512, 2)})
        print(json.dumps(rows[-1]), flush=True)


if __name__ == "__main__":
    main()
