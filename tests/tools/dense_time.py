"""Worst cases for the event-sparse kernel: streams where every chunk is an element (dev aid)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import hevcbitstream_amd as hbs
from tests import _orc
orc = _orc.oracle()
ctx = hbs.Context(0)
for name, make in (("all zeros", lambda n: np.zeros(n, dtype=np.uint8)),
                   ("00 00 03 repeated", lambda n: np.tile(np.array([0, 0, 3], dtype=np.uint8), n // 3 + 1)[:n]),
                   ("00 00 01 xx repeated", lambda n: np.tile(np.array([0, 0, 1, 0x42], dtype=np.uint8), n // 4)[:n])):
    for mib in (4, 64):
        n = mib << 20
        s = make(n)
        d = torch.from_numpy(s).cuda()
        index, rbsp, summary, cap = ctx.alloc_outputs(n, index_cap=n // 3 + 8)
        ctx.index_extract_async(d, index, cap, rbsp, summary); torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.index_extract_async(d, index, cap, rbsp, summary); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        sm = ctx.read_summary(summary)
        line = "%-22s %3d MiB: %.1f ms -> %.1f GB/s  nals %d err %d" % (name, mib, dt * 1e3, n / dt / 1e9, int(sm["nal_count"]), int(sm["error"]))
        if mib == 4:
            want_idx, want_arena, why = orc.index_extract(s)
            got_idx, got_arena, s2 = ctx.index_extract(d)
            ok = len(got_idx) == len(want_idx) and all(np.array_equal(got_idx[f], want_idx[f]) for f in ("start", "end", "rbsp_off", "rbsp_len", "status")) and np.array_equal(got_arena[:len(want_arena)], want_arena)
            line += "  parity " + ("OK" if ok else "MISMATCH")
        print(line)
