"""Timing of K3 (hbs_emit_annexb, default path: arena tiles on these arenas) on a synthetic arena (dev aid, not the bench): ~1 GiB by default,
HBS_EMIT_NALS=1677000 for the bench's 16 GiB."""
import sys
import torch
sys.path.insert(0, ".")
import os
if os.environ.get("HBS_LIB"):
    import hevcbitstream_amd.api as _api
    _api.library_path = lambda: os.environ["HBS_LIB"]
import hevcbitstream_amd as hbs
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
N = int(os.environ.get("HBS_EMIT_NALS", 104858))
ctx = hbs.Context(0)
g = ctx.synth_stream(0x1234, N, mode)          # runs the generator + K3 once
rb, sb = g["rbsp_bytes"], g["stream_bytes"]
print("rbsp bytes", rb, "stream bytes", sb)
out = torch.empty(sb + 4096, dtype=torch.uint8, device="cuda") if os.environ.get("HBS_PLAIN_ALLOC") else ctx.pair_alloc(g["rbsp"], sb + 4096)[0]
idx_out = torch.empty(N * 32, dtype=torch.uint8, device="cuda")
summary = torch.zeros(64, dtype=torch.uint8, device="cuda")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
for i in range(6):
    ctx.emit_annexb_async(g["rbsp"], rb, g["index"], N, 1, out, idx_out, summary)
    ev[i].record()
torch.cuda.synchronize()
ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(5)]
best = min(ts)
print("ms per call", ["%.3f" % t for t in ts])
print("best %.3f ms -> %.1f GB/s emitted, %.1f GB/s of HBM traffic (read once + written once)" % (best, sb / best / 1e6, (rb + sb) / best / 1e6))
assert torch.equal(out[:sb], g["stream"][:sb])
