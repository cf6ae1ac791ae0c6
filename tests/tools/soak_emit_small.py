"""Randomised soak of hbs_emit_annexb around the one-launch path's limits (256 NALs, 32 KiB) against the oracle,
every emit path on the same input (dev aid).  usage: python3 tests/tools/soak_emit_small.py [seconds] [seed]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import hevcbitstream_amd as hbs
from tests import _orc
from tests._orc import NAL_ENTRY

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
orc = _orc.oracle()
ctxs = {}
for path in (-1, 0, 1, 2):               # 2: the arena tiles whenever the index allows them, whatever the size
    ctxs[path] = hbs.Context(0)
    ctxs[path].set_emit_path(path)
ALPHA = np.array([0, 0, 0, 0, 1, 1, 2, 3, 3, 4, 0x40, 0x80, 0xFF], dtype=np.uint8)
t_end = time.time() + budget
it = bad = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed0 * 7919 + it)
    nn = int(rng.choice([1, 2, 3, 7, 64, 255, 256, 257, 300, 1000, 5000]))      # (from 257 NALs of a mean below 448 bytes: the group kernel, round 6)
    total = int(rng.choice([0, 1, 17, 1000, 1024, 4096, 5000, 32767, 32768, 32769, 40000, 44784, 196608, 196624, 200000, 393216, 700000]))
    cuts = np.sort(rng.integers(0, total + 1, size=nn - 1)) if nn > 1 else np.zeros(0, dtype=np.int64)
    if rng.random() < 0.3 and total > 64:            # a stretch in which every chunk has a NAL start: the tile is walked by rows
        k16 = int(rng.integers(2, max(3, min(600, total // 16))))
        at = 16 * int(rng.integers(0, total // 16 - k16 + 1))
        cuts = np.sort(np.concatenate((cuts, at + 16 * np.arange(1, k16 + 1)))).clip(0, total)
    lens = np.diff(np.concatenate(([0], cuts, [total]))).astype(np.int64)
    if rng.random() < 0.3:                           # empty NALs where the arena ends (round 6: lost by the walk by rows when the arena is whole chunks)
        lens = np.concatenate((lens, np.zeros(int(rng.integers(1, 4)), dtype=np.int64)))
    nn = len(lens)
    kind = rng.integers(0, 5)
    if kind == 0:
        arena = ALPHA[rng.integers(0, len(ALPHA), size=total)].copy()
    elif kind == 1:
        arena = rng.integers(0, 256, size=total, dtype=np.uint8)
    elif kind == 2:
        arena = np.zeros(total, dtype=np.uint8)
    elif kind == 3:
        arena = rng.integers(1, 256, size=total, dtype=np.uint8)
        arena[rng.random(total) < 0.05] = 0
    else:                                            # a run of zeros of a few KiB somewhere
        arena = rng.integers(1, 256, size=total, dtype=np.uint8)
        if total > 8:
            a = int(rng.integers(0, total - 1))
            arena[a: a + int(rng.integers(1, 5000))] = 0
    idx = np.zeros(nn, dtype=NAL_ENTRY)
    off = pos = 0
    for k in range(nn):
        g = int(rng.integers(3, 7))
        idx["start"][k] = pos + g; idx["end"][k] = pos + g + lens[k]
        idx["rbsp_off"][k] = off; idx["rbsp_len"][k] = lens[k]
        pos += g + int(lens[k]); off += int(lens[k])
    want = orc.emit_annexb(arena, idx)
    d = torch.from_numpy(arena).cuda() if total else torch.zeros(0, dtype=torch.uint8, device="cuda")
    for path, c in ctxs.items():
        if os.environ.get("SOAK_TRACE"):             # a fault kills the process: say what is about to run
            print("iter", it, "path", path, "nals", nn, "bytes", total, "kind", int(kind), flush=True)
        got, got_idx = c.emit_annexb(d, idx)
        ok = np.array_equal(got, want) and (nn == 0 or int(got_idx["end"][-1]) == len(want))
        if not ok:
            bad += 1
            print("EMIT MISMATCH path", path, "iter", it, "nals", nn, "bytes", total, "kind", kind)
    it += 1
print("iterations", it, "mismatches", bad)
