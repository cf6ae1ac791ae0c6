"""Timing of K4 (hbs_parse_headers) on config 3: synthetic 4K30 stream, ~100k NALs (dev aid)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import os
if os.environ.get("HBS_LIB"):
    import hevcbitstream_amd.api as _api
    _api.library_path = lambda: os.environ["HBS_LIB"]
import hevcbitstream_amd as hbs
from hevcbitstream_amd.api import PARSED, SUMMARY
from hevcbitstream_amd.hevc_synth import stream_4k30
stream, n = stream_4k30(11, n_pictures=12500, slices_per_picture=8, idr_every=60, payload_bytes=(60, 120))
s = np.frombuffer(stream, dtype=np.uint8).copy()
ctx = hbs.Context(0)
d = torch.from_numpy(s).cuda()
index, rbsp, summary, cap = ctx.alloc_outputs(d.numel())
ctx.index_extract_async(d, index, cap, rbsp, summary)
n = int(ctx.read_summary(summary)["nal_count"])
parsed, structs = ctx.parse_headers(rbsp, index, n)
pt = torch.empty(n * PARSED.itemsize, dtype=torch.uint8, device="cuda")
sm = torch.zeros(SUMMARY.itemsize, dtype=torch.uint8, device="cuda")
ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
for i in range(6):
    ctx.parse_headers_async(rbsp, index, n, pt, structs, sm)
    ev[i].record()
torch.cuda.synchronize()
ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(5)]
best = min(ts)
print("stream bytes", len(s), "nals", n, "struct arena bytes", structs.numel())
print("ms per call", ["%.3f" % t for t in ts])
print("best %.3f ms -> %.2f M NAL/s; struct arena written at %.1f GB/s" % (best, n / best / 1e3, structs.numel() / best / 1e6))

# K5: write the whole batch back (RBSP only)
cap = 256
parsed_dev = torch.from_numpy(parsed.view(np.uint8).copy()).cuda()
written, out = ctx.write_headers(parsed_dev, structs, n, cap)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
wr = torch.empty(n * 16, dtype=torch.uint8, device="cuda")
import ctypes as C
for i in range(4):
    ctx._bind_stream()
    rc = ctx.lib.hbs_write_headers(ctx.h, C.c_void_p(parsed_dev.data_ptr()), n, C.c_void_p(structs.data_ptr()), None, None,
                                   C.c_void_p(out.data_ptr()), cap, C.c_void_p(wr.data_ptr()))
    assert rc == 0
    ev[i].record()
torch.cuda.synchronize()
tw = min(ev[i].elapsed_time(ev[i + 1]) for i in range(3))
print("K5 write_headers: best %.3f ms -> %.2f M NAL/s (rc<0: %d)" % (tw, n / tw / 1e3, int((written["rc"] < 0).sum())))

# the opt-in sequential walk (one wavefront, NAL after NAL, one set of RPS tables): exact on any input, and this slow
ctx.set_sequential_parse(True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ctx.parse_headers_async(rbsp, index, n, pt, structs, sm); e1.record(); torch.cuda.synchronize()
print("sequential parse of the same batch: %.1f ms -> %.2f M NAL/s" % (e0.elapsed_time(e1), n / e0.elapsed_time(e1) / 1e3))
ctx.set_sequential_parse(False)
