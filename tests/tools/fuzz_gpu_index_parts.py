"""Randomised parity run for the index-only scan's tiles that are walked by rows (round 5: walked again by the emit pass in parts
of 32 rows, by helper wavefronts): streams of 0.2-12 MiB with random stretches of padding / zeros / tiny NALs / error patterns,
random cuts, at the tile height the call picks and at pinned ones (HBS5_TILE_ROWS is read once per process: a child per height).
usage: python3 tests/tools/fuzz_gpu_index_parts.py [streams per height] [seed]"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

WORKER = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import torch, hevcbitstream_amd as hbs
from tests import _orc
orc = _orc.oracle()
count, seed = int(sys.argv[1]), int(sys.argv[2])
ctxs = []
for k in (5, 0):
    c = hbs.Context(0); c.set_kernel(k); ctxs.append(c)
pats = [b"\x00\x00\x03", b"\x00", b"\x00\x00\x01\x42\x55", b"\x00\x00\x03\x00\x00\x02\x01", b"\x00\x00\x03\x04", b"\x00\x00\x00\x01\x26", b"\x00\x00\x03\x00"]
bad = 0
for it in range(count):
    rng = np.random.default_rng(seed * 7919 + it)
    n = int(rng.choice([200_000, 1_000_000, 3_000_000, 7_000_000, 12_000_000])) + int(rng.integers(0, 5000))
    s = rng.integers(1, 256, size=n, dtype=np.uint8)
    at = 0
    step = int(rng.choice([600, 3000, 20000]))
    while at + 8 < n:
        s[at:at + 4] = (0, 0, 1, 0x40)
        at += int(rng.integers(step // 2, step * 2))
    for _ in range(int(rng.integers(1, 6))):
        a = int(rng.integers(0, n - 10))
        b = min(n, a + int(rng.choice([3_000, 20_000, 70_000, 300_000, 1_500_000])))
        p = np.frombuffer(pats[int(rng.integers(0, len(pats)))], dtype=np.uint8)
        s[a:b] = np.tile(p, (b - a) // len(p) + 1)[:b - a]
    want, _, why = orc.index_extract(s)
    d = torch.from_numpy(s).cuda()
    for c in ctxs:
        got, arena, sm = c.index_extract(d, want_rbsp=False)
        ok = int(sm["error"]) == 0 and int(sm["stop_reason"]) == why and len(got) == len(want) and all(
            np.array_equal(got[f], want[f]) for f in ("start", "end", "rbsp_off", "rbsp_len", "status"))
        if not ok:
            bad += 1
            print("MISMATCH seed", seed, "iter", it, "len", n, "kernel", c.kernel(), flush=True)
print("streams", count, "mismatches", bad)
"""


def main():
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    total_bad = 0
    for rows in (0, 64, 128, 256, 352, 512):
        env = dict(os.environ)
        if rows:
            env["HBS5_TILE_ROWS"] = str(rows)
        r = subprocess.run([sys.executable, "-c", WORKER % HERE, str(count), str(seed + rows)], env=env, capture_output=True, text=True, timeout=1800)
        last = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "(no output) " + r.stderr[-500:]
        print("tile rows %s: %s" % (rows or "as the call picks", last), flush=True)
        if r.returncode != 0 or "mismatches 0" not in last:
            total_bad += 1
            print(r.stdout[-2000:], r.stderr[-2000:])
    print("heights with mismatches or errors:", total_bad)


if __name__ == "__main__":
    main()
