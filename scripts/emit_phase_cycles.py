"""Per-phase shader-clock sums of the single-pass K3 kernel (diagnostic build, `make diag`)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
so = "build/diag/libhbs_diag.so"
assert os.path.exists(so), "run `make diag` first"
import hevcbitstream_amd.api as api
api.library_path = lambda: so
import hevcbitstream_amd as hbs
ctx = hbs.Context(0)
ctx.set_emit_path(0)      # the single pass by NALs (k3_fused); scripts/emit_phase_tiles.py is the same for the arena-tile kernel
lib = api.load_library()
n = int(os.environ.get("HBS_EMIT_NALS", 104858))
g = ctx.synth_stream(0x1234, n, 0)
rb, sb = g["rbsp_bytes"], g["stream_bytes"]
out = torch.empty(sb + 4096, dtype=torch.uint8, device="cuda")
idx_out = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
summary = torch.zeros(64, dtype=torch.uint8, device="cuda")
for _ in range(2):
    ctx.emit_annexb_async(g["rbsp"], rb, g["index"], n, 1, out, idx_out, summary)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ctx.emit_annexb_async(g["rbsp"], rb, g["index"], n, 1, out, idx_out, summary); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
print("one call %.3f ms (with the timing marks)" % ms)
cyc = np.zeros((1024, 8), dtype=np.uint64)
lib.hbs_debug_phase_cycles_emit.argtypes = [C.c_void_p]
assert lib.hbs_debug_phase_cycles_emit(cyc.ctypes.data) == 0
act = cyc[:512, :7].astype(np.float64)
names = ["ticket", "index entries", "row loads", "flags+count", "barrier A", "look-back + barrier", "emit"]
tot = act.sum(axis=1).mean()
print("groups per workgroup %.1f; s_memtime ticks per workgroup %.0f (%.1f ticks/us if the kernel is the whole call)" % (n / 16 / 512, tot, tot / (ms * 1e3)))
for i, nm in enumerate(names):
    print("  %-22s %5.1f %%" % (nm, 100 * act[:, i].mean() / tot))
