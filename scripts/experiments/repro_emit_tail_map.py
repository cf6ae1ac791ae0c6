"""Which arenas ending in empty NALs does the pinned arena-tile emit path get wrong?  (round 6's soak finding; run with the library
before the fix: HBS_LIB=build/variants/before_fix/libhbs.so)"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import hevcbitstream_amd as hbs
from tests import _orc
from tests.test_gpu_emit import fake_index

orc = _orc.oracle()
ctx = hbs.Context(0)
ctx.set_emit_path(2)
rng = np.random.RandomState(1)
tile = 192 * 1024
for total in (16, 32, 48, 64, 1008, 1024, 1040, 2048, 4096, 44784, 52992, 65536, tile - 16, tile, tile + 16, 2 * tile, 2 * tile + 16, 3 * tile + 4096, 5 * tile + 1024 * 7 + 16):
    for last_len in (1, 5, 16, 100, 2000):
        for n_before in (1, 3, 300):
            if last_len + n_before > total:
                continue
            body = total - last_len
            cuts = np.sort(rng.choice(np.arange(1, body), size=min(body - 1, n_before - 1), replace=False)) if n_before > 1 and body > 1 else np.zeros(0, int)
            lens = [int(x) for x in np.diff(np.concatenate(([0], cuts, [body])))] + [last_len, 0]
            lens = [x for x in lens]
            if sum(lens) != total or min(lens[:-1]) < 0:
                continue
            gaps = [3] * len(lens)
            arena = rng.randint(1, 256, size=total).astype(np.uint8)
            idx = fake_index(lens, gaps)
            got, _ = ctx.emit_annexb(torch.from_numpy(arena).cuda(), idx)
            want = orc.emit_annexb(arena, idx)
            by_tiles = ctx.lib.hbs_ctx_last_emit_by_tiles(ctx.h)
            ok = len(got) == len(want) and bool(np.array_equal(got, want))
            print("total %8d (%%1024 = %4d) last_len %5d nals %4d tiles %d : %s" % (total, total % 1024, last_len, len(lens), by_tiles, "ok" if ok else "WRONG %d/%d" % (len(got), len(want))))
