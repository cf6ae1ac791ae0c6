"""dev aid: k3_tiles' phases (diag build, shader clock per tile) on a 2 GiB arena of NALs of a given mean size
usage: make diag; python3 scripts/emit_phase_mean.py 1024"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
sys.path.insert(0, "scripts")
so = "build/diag/libhbs_diag.so"
import hevcbitstream_amd.api as api
api.library_path = lambda: so
import hevcbitstream_amd as hbs
import nal_sweep
mean = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = hbs.Context(0)
lib = api.load_library()
arena, total, idx, n, stream, sb = nal_sweep.make_stream(torch, np, ctx, mean, 2 << 30)
out = torch.empty(sb + 4096, dtype=torch.uint8, device="cuda")
idx_out = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
summary = torch.zeros(64, dtype=torch.uint8, device="cuda")
ctx.set_emit_path(2)
for _ in range(2):
    ctx.emit_annexb_async(arena, total, idx, n, 1, out, idx_out, summary)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ctx.emit_annexb_async(arena, total, idx, n, 1, out, idx_out, summary); e1.record(); torch.cuda.synchronize()
print("mean %d: one call %.3f ms (with the timing marks)" % (mean, e0.elapsed_time(e1)))
cyc = np.zeros((1024, 8), dtype=np.uint64)
lib.hbs_debug_phase_cycles_emit.argtypes = [C.c_void_p]
assert lib.hbs_debug_phase_cycles_emit(cyc.ctypes.data) == 0
act = cyc[:512, :7].astype(np.float64)
names = ["ticket (+ wait for the previous tile's stores)", "row loads + NAL starts", "flags + list", "elements (count)", "look-back", "barrier", "elements (emit) + copy"]
tiles = total / 196608 / 512
tot = act.sum(axis=1).mean()
print("tiles per workgroup %.1f; cycles per tile %.0f" % (tiles, tot / tiles))
for i, nm in enumerate(names):
    print("  %-46s %7.0f cyc/tile %5.1f %%" % (nm, act[:, i].mean() / tiles, 100 * act[:, i].mean() / tot))
