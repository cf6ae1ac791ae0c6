#!/usr/bin/env python3
"""Index-only scan on the bench stream, kernel ms by HIP events (dev aid): python scripts/index5_time.py [reps]
HBS_LIB picks a development build, HBS5_WAVES_PER_CU the grid."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import hevcbitstream_amd as hbs
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ctx = hbs.Context(0)
ctx.enable_timing(True)
if os.environ.get("HBS5_NAL_MEAN"):            # 2 GiB of random payload in NALs of that mean size (scripts/nal_sweep.py) instead: no index to compare with
    import numpy as np
    import nal_sweep
    _, _, _, n, sbuf, sb = nal_sweep.make_stream(torch, np, ctx, int(os.environ["HBS5_NAL_MEAN"]), 2 << 30)
    stream = sbuf[:sb]
    os.environ["HBS5_NOCHECK"] = "1"
else:
    n = 1_677_000
    g = ctx.synth_stream(0x1234, n, 0)
    sb = g["stream_bytes"]
    stream = g["stream"][:sb]
index, rbsp, summary, cap = ctx.alloc_outputs(sb, index_cap=n + 8)
del rbsp
ks = []
for i in range(reps + 1):
    ctx.index_extract_async(stream, index, cap, None, summary)
    if i:
        ks.append(ctx.kernel_ms())
s = ctx.read_summary(summary)
if os.environ.get("HBS5_NOCHECK"):
    print(json.dumps({"lib": os.path.basename(os.path.dirname(os.environ.get("HBS_LIB", "/default/x"))), "waves_per_cu": os.environ.get("HBS5_WAVES_PER_CU", "20"),
                      "ms_min": round(min(ks), 4), "ms_med": round(sorted(ks)[len(ks) // 2], 4), "unchecked": True}))
    sys.exit(0)
assert int(s["error"]) == 0 and int(s["nal_count"]) == n, s
a = index[: n * 32].view(torch.int64).view(n, 4)
b = g["index"][: n * 32].view(torch.int64).view(n, 4)
assert torch.equal(a[:, :2], b[:, :2]), "index-only NAL index != generator's index"
ks.sort()
print(json.dumps({"lib": os.path.basename(os.path.dirname(os.environ.get("HBS_LIB", "/default/x"))), "waves_per_cu": os.environ.get("HBS5_WAVES_PER_CU", "20"),
                  "kernel": ctx.last_kernel(), "ms_min": round(ks[0], 4), "ms_med": round(ks[len(ks) // 2], 4),
                  "read_TBs_med": round((sb + 32 * n) / ks[len(ks) // 2] / 1e9, 3)}))
