"""Diagnostic for the register-resident kernel (hbs_scan3.hip): per-phase shader-clock shares."""
import ctypes as C, os, subprocess, sys, glob
import numpy as np, torch
sys.path.insert(0, ".")
so = "build/diag/libhbs_diag.so"      # built by `make diag` in the dev container
assert os.path.exists(so), "run `make diag` first"
import hevcbitstream_amd.api as api
api.library_path = lambda: so
import hevcbitstream_amd as hbs
from tests import _orc
orc = _orc.oracle()
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
want_rbsp = int(sys.argv[2]) if len(sys.argv) > 2 else 1
base, idx, arena = orc.gen_stream(0x1234, 1600, mode)
d = torch.from_numpy(base).cuda().repeat(64)
ctx = hbs.Context(0)
ctx.set_kernel(3)
index, rbsp, summary, cap = ctx.alloc_outputs(d.numel(), index_cap=1600 * 64 + 16)
for _ in range(3):
    ctx.index_extract_async(d, index, cap, rbsp if want_rbsp else None, summary)
torch.cuda.synchronize()
out = np.zeros((1024, 8), dtype=np.uint64)
lib = api.load_library()
lib.hbs_debug_phase_cycles3.argtypes = [C.c_void_p]
assert lib.hbs_debug_phase_cycles3(out.ctypes.data) == 0
names = ["passA classify", "wave-agg exch", "lookback", "passB emit+copy", "end barrier"]
act = out[:512, :5].astype(np.float64)
tiles = d.numel() / 65536 / 512
tot = act.sum(axis=1).mean()
print("v3 mode", mode, "rbsp", want_rbsp, "tiles/WG %.1f -> cycles/tile %.0f" % (tiles, tot / tiles))
for i, nm in enumerate(names):
    print("  %-18s %8.0f cyc/tile  %5.1f%%   (min WG %.0f, max WG %.0f)" % (nm, act[:, i].mean() / tiles, 100 * act[:, i].mean() / tot,
                                                                   act[:, i].min() / tiles, act[:, i].max() / tiles))
