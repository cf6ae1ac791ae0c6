"""K3 with parts switched off (diagnostic build, `make diag`): what the single-pass kernel spends its time on."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, ".")
so = "build/diag/libhbs_diag.so"
assert os.path.exists(so), "run `make diag` first"
import hevcbitstream_amd.api as api
api.library_path = lambda: so
import hevcbitstream_amd as hbs
ctx = hbs.Context(0)
lib = api.load_library()
n = 104858
g = ctx.synth_stream(0x1234, n, 0)
rb, sb = g["rbsp_bytes"], g["stream_bytes"]
out = torch.empty(sb + (64 << 20), dtype=torch.uint8, device="cuda")
idx_out = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
summary = torch.zeros(64, dtype=torch.uint8, device="cuda")
for exp, name in ((0, "everything"), (1, "no stores"), (2, "no look-back wait"), (4, "no count"), (3, "no stores, no look-back"),
                  (7, "loads + tickets only")):
    assert lib.hbs_debug_k3_exp(C.c_int(exp)) == 0
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    for i in range(5):
        ctx.emit_annexb_async(g["rbsp"], rb, g["index"], n, 1, out, idx_out, summary)
        ev[i].record()
    torch.cuda.synchronize()
    best = min(ev[i].elapsed_time(ev[i + 1]) for i in range(4))
    print("%-28s %.3f ms" % (name, best))
