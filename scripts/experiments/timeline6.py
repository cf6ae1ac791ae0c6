"""Diagnostic for hbs_scan6.hip: shader-clock stamps of the first workgroups' wavefronts (make diag)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
so = "build/diag/libhbs_diag.so"
assert os.path.exists(so), "run `make diag` first"
import hevcbitstream_amd.api as api
api.library_path = lambda: so
import hevcbitstream_amd as hbs
ctx = hbs.Context(0)
ctx.set_kernel(6)
n = 419000
g = ctx.synth_stream(0x1234, n, 0)
sb = g["stream_bytes"]
index, rbsp, summary, cap = ctx.alloc_outputs(sb, index_cap=n + 8)
want_rbsp = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for _ in range(3):
    ctx.index_extract_async(g["stream"][:sb], index, cap, rbsp if want_rbsp else None, summary)
torch.cuda.synchronize()
lib = api.load_library()
out = np.zeros((8, 4, 48, 12), dtype=np.uint64)
lib.hbs_debug_timeline6.argtypes = [C.c_void_p]
assert lib.hbs_debug_timeline6(out.ctypes.data) == 0
names = ["top", "flagsA", "-", "flagsB(+E)", "A:stamp seen", "resolve+emit done", "A:copied", "A:next+load", "B:stamp seen", "-", "B:copied", "B:next+load"]
for wg in (0, 3):
    t0 = int(out[wg, :, 20, 0].min())
    print("workgroup", wg, "iterations 20..23, cycles relative to the first wavefront's top of iteration 20")
    for it in range(20, 24):
        for w in range(4):
            row = out[wg, w, it].astype(np.int64)
            print("  it %d wave %d: " % (it, w) + " ".join("%s=%d" % (names[i], row[i] - t0) for i in range(12) if row[i] and i not in (2, 9)))
    d = out[wg, 0, 10:40, 0].astype(np.int64)
    print("  mean iteration length (wave 0, iterations 10..39): %.0f cycles" % np.diff(d).mean())
