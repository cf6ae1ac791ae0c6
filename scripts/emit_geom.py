"""K3 timing for experimental geometries built into build/exp/libhbs_<slots>_<rows>.so (dev aid)."""
import glob, subprocess, sys
import os
N = int(os.environ.get("HBS_EMIT_NALS", 104858))
if len(sys.argv) > 1:
    so = sys.argv[1]
    import torch
    sys.path.insert(0, ".")
    import hevcbitstream_amd.api as api
    api.library_path = lambda: so
    import hevcbitstream_amd as hbs
    ctx = hbs.Context(0)
    n = N
    for mode in (0, 1):
        g = ctx.synth_stream(0x1234, n, mode)
        rb, sb = g["rbsp_bytes"], g["stream_bytes"]
        out = torch.empty(sb + 4096, dtype=torch.uint8, device="cuda")
        idx_out = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
        summary = torch.zeros(64, dtype=torch.uint8, device="cuda")
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        for i in range(6):
            ctx.emit_annexb_async(g["rbsp"], rb, g["index"], n, 1, out, idx_out, summary)
            ev[i].record()
        torch.cuda.synchronize()
        best = min(ev[i].elapsed_time(ev[i + 1]) for i in range(5))
        ok = torch.equal(out[:sb], g["stream"][:sb])
        print("%s nals %d mode %d: %.3f ms -> %.1f GB/s emitted %s" % (so, n, mode, best, sb / best / 1e6, "OK" if ok else "MISMATCH"))
else:
    for so in sorted(glob.glob("build/exp/libhbs_*.so")):
        subprocess.run([sys.executable, __file__, so])
