"""K3 single-pass vs the three-step path as zero bytes get denser (dev aid)."""
import os, subprocess, sys
if len(sys.argv) > 1:
    import numpy as np, torch
    sys.path.insert(0, ".")
    import hevcbitstream_amd as hbs
    ctx = hbs.Context(0)
    rng = np.random.default_rng(3)
    n_nals, L = 20000, 10240
    tot = n_nals * L
    idx = np.zeros(n_nals, dtype=hbs.NAL_ENTRY)
    idx["rbsp_len"] = L
    idx["rbsp_off"] = np.arange(n_nals, dtype=np.uint64) * L
    idx["start"] = idx["rbsp_off"] + 4 * (np.arange(n_nals) + 1)
    idx["end"] = idx["start"] + L
    d_idx = torch.from_numpy(idx.view(np.uint8).copy()).cuda()
    for pz in (1 / 256, 0.01, 0.02, 0.03, 0.05, 0.1):
        a = rng.integers(1, 256, size=tot, dtype=np.uint8)
        a[rng.random(tot) < pz] = 0
        arena = torch.from_numpy(a).cuda()
        out = torch.empty(tot + tot // 2 + 4096, dtype=torch.uint8, device="cuda")
        idx_out = torch.empty(n_nals * 32, dtype=torch.uint8, device="cuda")
        summary = torch.zeros(64, dtype=torch.uint8, device="cuda")
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        for i in range(4):
            ctx.emit_annexb_async(arena, tot, d_idx, n_nals, 0, out, idx_out, summary)
            ev[i].record()
        torch.cuda.synchronize()
        ms = min(ev[i].elapsed_time(ev[i + 1]) for i in range(3))
        print("%s p(zero) %.4f: %.3f ms -> %.1f GB/s emitted" % (sys.argv[1], pz, ms, int(ctx.read_summary(summary)["stream_bytes"]) / ms / 1e6))
else:
    for tp in ("0", "1", None):
        env = dict(os.environ)
        env.pop("HBS_EMIT_TWO_PASS", None)
        if tp is not None:
            env["HBS_EMIT_TWO_PASS"] = tp
        subprocess.run([sys.executable, __file__, {"1": "three-step", "0": "single-pass", None: "automatic"}[tp]], env=env)
