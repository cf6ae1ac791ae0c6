"""Kernel 2 vs kernel 4 vs the automatic choice as zero bytes get denser (dev aid).
usage: python3 scripts/density_sweep.py [MiB]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import hevcbitstream_amd as hbs

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = mib << 20
rng = np.random.default_rng(5)
ctx = hbs.Context(0)
ctx.enable_timing(True)
print("zero prob | flagged chunks | kernel 4 GB/s | kernel 2 GB/s | automatic GB/s (picked)")
for pz in (1 / 256, 0.008, 0.012, 0.016, 0.02, 0.03, 0.05, 0.1, 0.3):
    s = rng.integers(1, 256, size=n, dtype=np.uint8)
    s[rng.random(n) < pz] = 0
    for p in range(1000, n - 8, 10007):
        s[p:p + 4] = (0, 0, 1, 0x42)
    z = s == 0
    pair = z[:-1] & z[1:]
    flagged = np.zeros(n // 16 + 2, dtype=bool)
    pos = np.nonzero(pair)[0]
    flagged[pos // 16] = True
    flagged[(pos + 2) // 16] = True
    frac = flagged.mean()
    d = torch.from_numpy(s).cuda()
    index, rbsp, summary, cap = ctx.alloc_outputs(n, index_cap=n // 64)
    row = []
    for v in (4, 2, 0):
        ctx.set_kernel(v)
        best = 1e9
        for _ in range(4):
            ctx.index_extract_async(d, index, cap, rbsp, summary)
            best = min(best, ctx.kernel_ms())
        row.append(n / best / 1e6)
    print("%.4f | %.4f | %8.1f | %8.1f | %8.1f (%d)" % (pz, frac, row[0], row[1], row[2], ctx.last_kernel()))
    del d, index, rbsp
