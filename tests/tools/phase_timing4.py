"""Diagnostic for the event-sparse kernel (hbs_scan4.hip): per-phase shader-clock sums per workgroup."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
so = os.environ.get("HBS_DIAG_LIB", "build/diag/libhbs_diag.so")      # built by `make diag` in the dev container
assert os.path.exists(so), "run `make diag` first"
import hevcbitstream_amd.api as api
api.library_path = lambda: so
import hevcbitstream_amd as hbs
from tests import _orc
orc = _orc.oracle()
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
want_rbsp = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = hbs.Context(0)
if os.environ.get('HBS4_NAL_MEAN'):          # a stream of random payload cut into NALs of that mean size (scripts/nal_sweep.py)
    sys.path.insert(0, "scripts")
    import nal_sweep
    _, _, _, nn, sbuf, sb = nal_sweep.make_stream(torch, np, ctx, int(os.environ['HBS4_NAL_MEAN']), 1 << 30)
    d = sbuf[:sb]
    ncap = nn + 64
else:
    base, idx, arena = orc.gen_stream(0x1234, 1600, mode)
    rep = int(os.environ.get('HBS4_REPEAT', 64))
    d = torch.from_numpy(base).cuda().repeat(rep)
    ncap = 1600 * rep + 16
variant = int(os.environ.get("HBS4_VARIANT", 4))          # 6: the 24-row geometry (HBS4_TILE=98304)
ctx.set_kernel(variant)
blocks, per_cu = ctx.grid()
index, rbsp, summary, cap = ctx.alloc_outputs(d.numel(), index_cap=ncap)
lib = api.load_library()
if os.environ.get('HBS4_FAKE_LB'):
    assert lib.hbs_debug_fake_lb4(C.c_int(1)) == 0
for _ in range(3):
    ctx.index_extract_async(d, index, cap, rbsp if want_rbsp else None, summary)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ctx.index_extract_async(d, index, cap, rbsp if want_rbsp else None, summary); e1.record(); torch.cuda.synchronize()
print('one call %.3f ms -> %.1f GB/s' % (e0.elapsed_time(e1), d.numel() / e0.elapsed_time(e1) / 1e6))
out = np.zeros((1024, 8), dtype=np.uint64)
lib = api.load_library()
fn = lib.hbs_debug_phase_cycles4 if variant == 4 else lib.hbs_debug_phase_cycles4_r24
fn.argtypes = [C.c_void_p]
assert fn(out.ctypes.data) == 0
names = ["ticket+fetch issue", "flags+list", "elements A", "lookback", "elements B", "copy"]
nwg = min(blocks, 1024)
act = out[:nwg, :6].astype(np.float64)
tile_bytes = int(os.environ.get("HBS4_TILE", 196608 if variant == 4 else 98304))
tiles = d.numel() / tile_bytes / blocks
tot = act.sum(axis=1).mean()
print("v4 mode", mode, "rbsp", want_rbsp, "grid", blocks, "per CU", per_cu, "tiles/WG %.1f -> cycles/tile %.0f" % (tiles, tot / tiles))
lb = out[:nwg, 7]
print("  look-back steps/tile %.2f, of which stalled %.2f" % ((lb & 0xFFFFFFFF).astype(np.float64).mean() / tiles, (lb >> 32).astype(np.float64).mean() / tiles))
for i, nm in enumerate(names):
    print("  %-18s %8.0f cyc/tile  %5.1f%%   (min WG %.0f, max WG %.0f)" % (nm, act[:, i].mean() / tiles, 100 * act[:, i].mean() / tot,
                                                                   act[:, i].min() / tiles, act[:, i].max() / tiles))
