"""Does the AUTOMATIC emit path take the arena tiles on a 200 MiB arena of whole chunks that ends in an empty NAL (round 6's soak finding)?
usage: [HBS_LIB=...] python3 scripts/experiments/repro_emit_tail_200m.py"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import hevcbitstream_amd as hbs
from tests import _orc
from tests.test_gpu_emit import fake_index

orc = _orc.oracle()
ctx = hbs.Context(0)
rng = np.random.RandomState(607)
for extra in (0, 1):
    total = (200 << 20) + extra
    nn = total // 10240
    cuts = np.sort(rng.choice(np.arange(1, total, 7), size=nn - 1, replace=False))
    lens = [int(x) for x in np.diff(np.concatenate(([0], cuts, [total])))] + [0]
    gaps = [3 + (k & 1) for k in range(len(lens))]
    arena = rng.randint(0, 256, size=total).astype(np.uint8)
    arena[rng.randint(0, total, size=total // 5000)] = 0
    arena[total - 50000: total - 47000] = 0          # three KiB of zeros in the last tile: it is walked by rows
    idx = fake_index(lens, gaps)
    want = orc.emit_annexb(arena, idx)
    for path in (-1, 2):
        ctx.set_emit_path(path)
        got, _ = ctx.emit_annexb(torch.from_numpy(arena).cuda(), idx)
        print("extra", extra, "path", path, "by tiles", ctx.lib.hbs_ctx_last_emit_by_tiles(ctx.h), "len", len(got), "want", len(want), "equal", len(got) == len(want) and bool(np.array_equal(got, want)))
