"""K3 three ways on the same arena (dev aid): the item kernel (path 0), count / scan / emit (1), arena tiles (2) -- outputs and
output indexes compared, best-of-5 time of each.  HBS_EMIT_NALS sets the arena (default 104858 NALs ~ 1 GiB).
HBS_EMIT_MIXED=1: 1 % of the arena overwritten by 00 00 03 padding in 640 KiB regions placed between the density probe's 64 windows
(the probe says sparse; the tile kernel meets tiles dense in elements, gives up, and the kernel by NALs does the call)."""
import os
import sys
import torch
sys.path.insert(0, ".")
if os.environ.get("HBS_LIB"):
    import hevcbitstream_amd.api as _api
    _api.library_path = lambda: os.environ["HBS_LIB"]
import hevcbitstream_amd as hbs
N = int(os.environ.get("HBS_EMIT_NALS", 104858))
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ctx = hbs.Context(0)
ctx.set_emit_path(0)
g = ctx.synth_stream(0x1234, N, mode)
rb, sb = g["rbsp_bytes"], g["stream_bytes"]
mixed = os.environ.get("HBS_EMIT_MIXED") in ("1", "2")          # 2: the regions are pure zeros (what cabac_zero_words leave in an RBSP)
if mixed:
    region = 640 << 10
    stride = (rb // 64) & ~15
    pat = torch.tensor([0, 0, 3], dtype=torch.uint8, device="cuda").repeat(region // 3 + 1)[:region]
    if os.environ.get("HBS_EMIT_MIXED") == "2":
        pat = torch.zeros(region, dtype=torch.uint8, device="cuda")
    for k in range(max(1, int(rb * 0.01 / region))):
        off = (k % 64) * stride + stride // 2 + (k // 64) * (region + 4096)
        if off + region < rb:
            g["rbsp"][off: off + region] = pat
summary = torch.zeros(64, dtype=torch.uint8, device="cuda")
ref_out = ref_idx = None
for path in ((2,) if os.environ.get("HBS_ONLY_TILES") else (0, 1, 2)):
    ctx.set_emit_path(path)
    out = torch.zeros(sb + (sb // 40 if mixed else 0) + 4096, dtype=torch.uint8, device="cuda")
    idx_out = torch.zeros(N * 32, dtype=torch.uint8, device="cuda")
    ts = []
    for i in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ctx.emit_annexb_async(g["rbsp"], rb, g["index"], N, 1, out, idx_out, summary)
        e1.record()
        torch.cuda.synchronize()
        if i:
            ts.append(e0.elapsed_time(e1))
    s = ctx.read_summary(summary)
    assert int(s["error"]) == 0 and (mixed or int(s["stream_bytes"]) == sb), (path, s)
    if ref_out is None:
        ref_out, ref_idx = out, idx_out
        if mixed:
            sb = int(s["stream_bytes"])
        else:
            assert torch.equal(out[:sb], g["stream"][:sb])
    else:
        if not os.environ.get("HBS_NO_CHECK"):
            assert torch.equal(out[:sb], ref_out[:sb]), "path %d: bytes differ" % path
            assert torch.equal(idx_out, ref_idx), "path %d: output index differs" % path
    if os.environ.get("HBS_DZ_TIMING") and path == 2:
        import ctypes, numpy as np
        buf = (ctypes.c_ulonglong * 8)()
        ctx.lib.hbs_debug_dz_cycles(buf, 1)
        v = list(buf)
        calls = 6
        tiles = max(1, v[7])
        print("dense tiles per call %.0f; per tile us: before-dense %.1f | count %.1f | wait-waves %.1f | entry-count %.1f | look-back %.1f | emit %.1f" % (
            tiles / calls, v[5] / tiles / 100, v[0] / tiles / 100, v[1] / tiles / 100, v[2] / tiles / 100, v[3] / tiles / 100, v[4] / tiles / 100))
    print("path %d: best %.3f ms -> %.1f GB/s emitted, %.1f GB/s of traffic" % (path, min(ts), sb / min(ts) / 1e6, (rb + sb) / min(ts) / 1e6))
