"""First-look timing of K12 on a replicated synthetic stream (dev aid, not the bench)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import os
if os.environ.get("HBS_LIB"):
    import hevcbitstream_amd.api as _api
    _api.library_path = lambda: os.environ["HBS_LIB"]
import hevcbitstream_amd as hbs
from tests import _orc

orc = _orc.oracle()
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 64
base, idx, arena = orc.gen_stream(0x1234, 1600, mode)      # ~16 MiB
d1 = torch.from_numpy(base).cuda()
d = d1.repeat(reps)
n = d.numel()
ctx = hbs.Context(0)
import os
if os.environ.get('HBS_KERNEL'): print('kernel variant', ctx.kernel())
index, rbsp, summary, cap = ctx.alloc_outputs(n, index_cap=1600 * reps + 16)
print("stream bytes", n, "grid", "cap", cap)
for it in range(3):
    ctx.index_extract_async(d, index, cap, rbsp, summary)
torch.cuda.synchronize()
s = ctx.read_summary(summary)
print(s)
assert int(s["nal_count"]) == 1600 * reps, s
ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
ev[0].record()
for it in range(10):
    ctx.index_extract_async(d, index, cap, rbsp, summary)
    ev[it + 1].record()
torch.cuda.synchronize()
ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(10)]
print("ms per call:", ["%.3f" % t for t in ts])
best = min(ts)
print("best %.3f ms -> %.1f GB/s stream, %.1f GB/s traffic(2B/B)" % (best, n / best / 1e6, 2 * n / best / 1e6))
# index-only
for it in range(2):
    ctx.index_extract_async(d, index, cap, None, summary)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ctx.index_extract_async(d, index, cap, None, summary); e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1)
print("index-only %.3f ms -> %.1f GB/s" % (t, n / t / 1e6))
