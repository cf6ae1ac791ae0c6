"""K3 on a size mix like coded video (dev aid): per 60 pictures one ~400 KiB picture, the rest 20-60 KiB, plus
parameter-set sized NALs.  usage: python3 scripts/emit_real.py [GiB]"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import os
if os.environ.get("HBS_LIB"):
    import hevcbitstream_amd.api as _api
    _api.library_path = lambda: os.environ["HBS_LIB"]
import hevcbitstream_amd as hbs
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
rng = np.random.default_rng(9)
lens = []
tot = 0
while tot < gib * 2**30:
    lens += [30, 60, 10]                                     # VPS/SPS/PPS sized
    lens.append(int(rng.integers(250_000, 600_000)))          # IDR picture
    lens += [int(x) for x in rng.integers(20_000, 60_000, size=59)]
    tot = sum(lens)
n = len(lens)
idx = np.zeros(n, dtype=hbs.NAL_ENTRY)
idx["rbsp_len"] = lens
idx["rbsp_off"] = np.concatenate([[0], np.cumsum(lens)[:-1]])
idx["start"] = idx["rbsp_off"] + 4 * (np.arange(n) + 1)      # 4-byte start codes
idx["end"] = idx["start"] + idx["rbsp_len"]
ctx = hbs.Context(0)
arena = torch.randint(1, 256, (tot,), dtype=torch.uint8, device="cuda")
d_idx = torch.from_numpy(idx.view(np.uint8).copy()).cuda()
out = torch.empty(tot + 8 * n + 4096, dtype=torch.uint8, device="cuda")
idx_out = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
summary = torch.zeros(64, dtype=torch.uint8, device="cuda")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
for i in range(5):
    ctx.emit_annexb_async(arena, tot, d_idx, n, 0, out, idx_out, summary)
    ev[i].record()
torch.cuda.synchronize()
s = ctx.read_summary(summary)
ms = min(ev[i].elapsed_time(ev[i + 1]) for i in range(4))
print("NALs %d, %.2f GiB, largest %d KiB: %.3f ms -> %.1f GB/s emitted (error %d)" % (n, tot / 2**30, max(lens) >> 10, ms, int(s["stream_bytes"]) / ms / 1e6, int(s["error"])))
# spot check: every NAL's bytes are where the output index says (no zero pairs in this arena: nothing inserted)
o = idx_out.cpu().numpy().view(hbs.NAL_ENTRY)
for k in (0, 3, 4, n // 2, n - 1):
    a, b = int(o["start"][k]), int(o["end"][k])
    assert b - a == lens[k], (k, a, b, lens[k])
    assert torch.equal(out[a:b], arena[int(idx["rbsp_off"][k]): int(idx["rbsp_off"][k]) + lens[k]]), k
print("spot checks OK")
