#!/usr/bin/env python3
"""Where does the index-only scan's process-to-process spread (2.7 ... 3.0 ms on the 16 GiB bench stream, same box, same library)
come from?  One process, ONE library: the same stream bytes copied into several allocations (torch's allocator, i.e. hipMalloc, and
hbs_pair_alloc's 1 GiB chunks), the scan timed on each in turn, three rounds.  If the time belongs to the BUFFER -- the same buffer
the same time in every round, different buffers different times -- the spread is where the stream's pages lie, not the process."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hevcbitstream_amd as hbs

ctx = hbs.Context(0)
ctx.enable_timing(True)
n = 1_677_000
g = ctx.synth_stream(0x1234, n, 0)
sb = g["stream_bytes"]
src = g["stream"][:sb]
bufs = [("generator's", src)]
for k in range(3):
    t = torch.empty(sb + 64, dtype=torch.uint8, device="cuda")
    t[:sb] = src
    bufs.append(("torch#%d" % k, t[:sb]))
for k in range(2):
    t, rep = ctx.pair_alloc(src, sb + 64)
    t[:sb] = src
    bufs.append(("pair_alloc#%d" % k, t[:sb]))
index = torch.empty((n + 8) * 32, dtype=torch.uint8, device="cuda")
summary = torch.zeros(64, dtype=torch.uint8, device="cuda")
rows = {name: [] for name, _ in bufs}
for rnd in range(3):
    for name, s in bufs:
        ks = []
        for i in range(4):
            ctx.index_extract_async(s, index, n + 8, None, summary)
            if i:
                ks.append(ctx.kernel_ms())
        assert int(ctx.read_summary(summary)["nal_count"]) == n
        rows[name].append(round(sorted(ks)[1], 4))
for name, _ in bufs:
    print(json.dumps({"buffer": name, "addr": hex(dict(bufs)[name].data_ptr()), "kernel_ms_by_round": rows[name]}), flush=True)
