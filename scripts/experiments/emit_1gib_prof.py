"""K3 on the 1 GiB arena (BASELINE config 4), default path, a few calls: for rocprofv3 --kernel-trace --stats (dev aid)."""
import sys, torch
sys.path.insert(0, ".")
import hevcbitstream_amd as hbs
N = 104858
ctx = hbs.Context(0)
g = ctx.synth_stream(0x1234, N, 0)
rb, sb = g["rbsp_bytes"], g["stream_bytes"]
out = torch.zeros(sb + 4096, dtype=torch.uint8, device="cuda")
idx_out = torch.zeros(N * 32, dtype=torch.uint8, device="cuda")
summary = torch.zeros(64, dtype=torch.uint8, device="cuda")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
for i in range(9):
    ev[i].record()
    if i < 8:
        ctx.emit_annexb_async(g["rbsp"], rb, g["index"], N, 1, out, idx_out, summary)
torch.cuda.synchronize()
print("ms per call", ["%.3f" % ev[i].elapsed_time(ev[i + 1]) for i in range(8)])
assert torch.equal(out[:sb], g["stream"][:sb])
