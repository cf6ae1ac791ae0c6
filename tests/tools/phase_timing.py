"""Diagnostic: build the library with -DHBS_PHASE_TIMING into gpurun_out/diag and print
where a workgroup of K12 spends its shader clocks (shares, not absolute speed)."""
import ctypes as C, os, subprocess, sys
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
ctx = hbs.Context(0)
ctx.set_kernel(2)                                        # the LDS-image kernel, whatever the density probe would say
if os.environ.get("HBS2_NAL_MEAN"):                      # a 1 GiB stream of random payload in NALs of that mean size (scripts/nal_sweep.py)
    sys.path.insert(0, "scripts")
    import nal_sweep
    _, _, _, nn, sbuf, sb = nal_sweep.make_stream(torch, np, ctx, int(os.environ["HBS2_NAL_MEAN"]), 1 << 30)
    d = sbuf[:sb]
    index, rbsp, summary, cap = ctx.alloc_outputs(d.numel(), index_cap=nn + 64)
else:
    base, idx, arena = orc.gen_stream(0x1234, 1600, mode)
    d = torch.from_numpy(base).cuda().repeat(64)
    index, rbsp, summary, cap = ctx.alloc_outputs(d.numel(), index_cap=1600 * 64 + 16)
for _ in range(3):
    ctx.index_extract_async(d, index, cap, rbsp if want_rbsp else None, summary)
torch.cuda.synchronize()
out = np.zeros((1024, 8), dtype=np.uint64)
lib = api.load_library()
lib.hbs_debug_phase_cycles.argtypes = [C.c_void_p]
assert lib.hbs_debug_phase_cycles(out.ctypes.data) == 0
names = ["stage+wait", "classify", "scan", "lookback", "emit", "gather-fast", "gather-slow+sync", "-"]
act = out[:512].astype(np.float64)
tiles = d.numel() / 65536 / 512
tot = act.sum(axis=1).mean()
print("mode", mode, "rbsp", want_rbsp, "tiles/WG %.1f  total cycles/WG %.0f  -> cycles/tile %.0f" % (tiles, tot, tot / tiles))
lb = out[:512, 7]
print("  look-back steps/tile %.2f, of which stalled %.2f" % ((lb & 0xFFFFFFFF).astype(np.float64).mean() / tiles, (lb >> 32).astype(np.float64).mean() / tiles))
act[:, 7] = 0
tot = act.sum(axis=1).mean()
for i, nm in enumerate(names[:7]):
    print("  %-18s %8.0f cyc/tile  %5.1f%%   (min WG %.0f, max WG %.0f)" % (nm, act[:, i].mean() / tiles, 100 * act[:, i].mean() / tot,
                                                                   act[:, i].min() / tiles, act[:, i].max() / tiles))
