"""dev aid: extract / index only on 2 GiB streams of small NALs with the kernel pinned (4: event-sparse, 5: streaming index-only, 2: LDS image)"""
import sys, os
import numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "scripts")
if os.environ.get("HBS_LIB"):
    import hevcbitstream_amd.api as _api
    _api.library_path = lambda: os.environ["HBS_LIB"]
import hevcbitstream_amd as hbs
import nal_sweep
ctx = hbs.Context(0)
for mean in [int(x) for x in sys.argv[1].split(",")]:
    arena, total, idx, n, stream, sb = nal_sweep.make_stream(torch, np, ctx, mean, 2 << 30)
    s = stream[:sb]
    index, rbsp, summary, cap = ctx.alloc_outputs(sb, index_cap=n + 8)
    row = {"mean": mean}
    for k in ([int(x) for x in os.environ["HBS_PIN"].split(",")] if os.environ.get("HBS_PIN") else (2, 4, 5)):
        ctx.set_kernel(k)
        for arena_on in (True, False):
            fn = (lambda: ctx.index_extract_async(s, index, cap, rbsp if arena_on else None, summary))
            ms = nal_sweep.best_ms(torch, fn)
            sm = ctx.read_summary(summary)
            assert int(sm["error"]) == 0 and int(sm["nal_count"]) == n, (k, arena_on, sm)
            if arena_on:
                assert torch.equal(rbsp[:total], arena[:total]), "kernel %d: extracted arena != the arena the stream was made from" % k
            row["k%d %s" % (k, "extract" if arena_on else "index")] = round(ms, 3)
    ctx.set_kernel(0)
    print(row)
