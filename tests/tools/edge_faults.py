"""Buffers that end where their allocation ends, at sizes that are whole numbers of every kernel's tile (dev aid, round 6): a kernel
that reads or writes behind its input or output takes a GPU memory fault here instead of reading bytes nobody looks at.  Each case
runs in a process of its own (a fault kills it) and is compared with the oracle.
usage: python3 tests/tools/edge_faults.py                 (all cases)
       python3 tests/tools/edge_faults.py CASE SIZE EXTRA (one case, in this process)"""
import subprocess, sys
import numpy as np

MIB = 1 << 20


def stream_of(size, rng, mean):
    s = rng.integers(1, 256, size=size, dtype=np.uint8)
    s[rng.random(size) < 0.01] = 0
    pos = 0
    while pos + 8 < size:
        s[pos:pos + 4] = (0, 0, 1, 0x42)
        pos += int(rng.integers(mean // 2, mean * 3 // 2))
    return s


def one(case, size, extra):
    import torch
    sys.path.insert(0, ".")
    import hevcbitstream_amd as hbs
    from tests import _orc
    orc = _orc.oracle()
    rng = np.random.default_rng(size + extra)
    n = size + extra
    c = hbs.Context(0)
    if case.startswith("scan"):
        kernel, rbsp = int(case[4]), case.endswith("r")
        s = stream_of(n, rng, 9000 if kernel != 2 else 300)
        want_idx, want_arena, why = orc.index_extract(s)
        d = torch.empty(n, dtype=torch.uint8, device="cuda")          # >= 10 MiB: an allocation of its own, rounded to 2 MiB
        d.copy_(torch.from_numpy(s))
        c.set_kernel(kernel)
        got_idx, got_arena, sm = c.index_extract(d, want_rbsp=rbsp)
        ok = int(sm["error"]) == 0 and len(got_idx) == len(want_idx) and all(np.array_equal(got_idx[f], want_idx[f]) for f in ("start", "end", "rbsp_off", "rbsp_len", "status"))
        if rbsp:
            tot = int(want_idx["rbsp_off"][-1] + want_idx["rbsp_len"][-1])
            ok = ok and np.array_equal(got_arena[:tot], want_arena[:tot])
    else:
        tiny = case == "emitg"                    # arenas of ~100-byte NALs on the automatic path: the group kernels (hbs_emit_groups.h)
        path = -1 if tiny else int(case[4:])
        from tests.test_gpu_emit import fake_index
        nn = max(1, n // (100 if tiny else 9000))
        cuts = np.sort(rng.integers(1, n, size=nn - 1)) if nn > 1 else np.zeros(0, dtype=np.int64)
        lens = [int(x) for x in np.diff(np.concatenate(([0], cuts, [n])))]
        arena = rng.integers(0, 256, size=n, dtype=np.uint8)
        idx = fake_index(lens, [3 + (k & 1) for k in range(len(lens))])
        d = torch.empty(n, dtype=torch.uint8, device="cuda")
        d.copy_(torch.from_numpy(arena))
        c.set_emit_path(path)
        got, _ = c.emit_annexb(d, idx)
        want = orc.emit_annexb(arena, idx)
        ok = len(got) == len(want) and np.array_equal(got, want)
    print("case %-7s size %4d MiB %+3d: %s" % (case, size // MIB, extra, "ok" if ok else "WRONG"), flush=True)


if len(sys.argv) == 4:
    one(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
else:
    bad = 0
    cases = ["scan0r", "scan2r", "scan4r", "scan6r", "scan0", "scan2", "scan5", "emit-1", "emit0", "emit1", "emit2", "emitg"]
    # (torch gives a tensor of >= 10 MiB an allocation of its own, rounded up to 2 MiB: only sizes that are multiples of 2 MiB end
    # with their allocation -- 12 MiB is a whole number of every kernel's tiles, 10 and 14 MiB of none but the 256 KiB ones)
    for size in (10 * MIB, 12 * MIB, 14 * MIB, 192 * MIB, 1536 * MIB):
        for extra in (0, 5, -5, 16, -16):
            for case in cases:
                if size > 192 * MIB and (extra not in (0, 5) or case in ("scan2r", "scan2", "emitg")):
                    continue
                r = subprocess.run([sys.executable, __file__, case, str(size), str(extra)], capture_output=True, text=True)
                lines = [x for x in (r.stdout + r.stderr).splitlines() if x.startswith("case") or "fault" in x]
                line = lines[-1] if lines else "case %s size %d %+d: rc %d, no output: %s" % (case, size // MIB, extra, r.returncode, (r.stderr or "")[-200:])
                if not line.endswith(": ok"):
                    bad += 1
                    line = "case %-7s size %4d MiB %+3d: %s" % (case, size // MIB, extra, line) if "fault" in line else line
                print(line, flush=True)
    print("cases that did not come back ok:", bad)
