"""Randomised soak at the sizes where the AUTOMATIC paths change (dev aid, round 6): the arena-tile emit from 192 MiB, the streaming
index-only kernel from 1 GiB, its workgroup-per-tile schedule from 1.5 GiB, the count-ahead passes from 3 GiB.  tests/tools/soak_gpu.py's
streams are at most 20 MB, the full-size tests' streams are regular; here streams of 0.2-3.5 GiB are tiled from a few irregular
pages (random bytes, zero-heavy stretches, runs of zeros, tiny NALs, patterns), cut at sizes that are whole tiles of the kernels
(or a few bytes either side) half of the time, and go through hbs_index_extract (with and without the arena) and hbs_emit_annexb on
the automatic paths, against the oracle.  usage: python3 tests/tools/soak_big.py [seconds] [seed] [max GiB]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import hevcbitstream_amd as hbs
from tests import _orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
max_gib = float(sys.argv[3]) if len(sys.argv) > 3 else 3.5
orc = _orc.oracle()
ctx = hbs.Context(0)
ALPHA = np.array([0, 0, 0, 0, 1, 1, 2, 3, 3, 4, 0x40, 0x80, 0xFF], dtype=np.uint8)
MIB = 1 << 20


def page(rng, kind, n):
    if kind == 0:
        p = rng.integers(0, 256, size=n, dtype=np.uint8)                           # coded-video-like
    elif kind == 1:
        p = rng.integers(1, 256, size=n, dtype=np.uint8); p[rng.random(n) < 0.03] = 0
    elif kind == 2:
        p = ALPHA[rng.integers(0, len(ALPHA), size=n)]                              # dense in patterns
    elif kind == 3:
        p = np.zeros(n, dtype=np.uint8)                                             # a run of zeros
    else:
        p = np.tile(np.array([0, 0, 3, 0, 0, 1, 9], dtype=np.uint8), n // 7 + 1)[:n]
    return p


def make_stream(rng, size):
    # mostly coded-video-like pages with NALs of a few KiB, a few pages of another character
    base = page(rng, 0, 8 * MIB)
    mean = int(rng.choice([700, 3000, 10000, 60000]))
    pos = 0
    while pos + 8 < len(base):
        base[pos:pos + 4] = (0, 0, 0, 1) if rng.random() < 0.3 else (0x55, 0, 0, 1)
        base[pos + 4] = 0x42
        pos += int(rng.integers(mean // 2, mean * 3 // 2))
    s = np.tile(base, size // len(base) + 1)[:size].copy()
    for _ in range(int(rng.integers(0, 6))):                                         # stretches of another character
        kind = int(rng.choice([1, 2, 4, 1, 2, 4, 3]))
        n = int(rng.integers(1, 3 * MIB)) if kind != 3 else int(rng.integers(1, 65536))
        at = int(rng.integers(0, max(1, size - n)))
        s[at:at + n] = page(rng, kind, min(n, size - at))
    # start codes near multiples of the tile sizes, and one near the end
    for edge in (98304, 196608, 262144):
        step = edge * int(rng.integers(50, 400))
        for m in range(step, size - 8, step):
            o = m + int(rng.integers(-5, 3))
            s[o:o + 4] = (0, 0, 1, int(rng.integers(1, 255)))
    if rng.random() < 0.5 and size > 64:
        o = size - int(rng.integers(4, 40))
        s[o:o + 3] = (0, 0, 1)
    return s


t_end = time.time() + budget
it = bad = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed0 * 70001 + it)
    size = int(rng.integers(192 * MIB, int(max_gib * 1024) * MIB))
    if rng.random() < 0.5:
        unit = int(rng.choice([98304, 196608, 262144, MIB, 16]))
        size = max(unit, size // unit * unit) + int(rng.choice([0, 0, 1, -1, 15, 16, 17, -16]))
    t0 = time.time()
    s = make_stream(rng, size)
    want_idx, want_arena, why = orc.index_extract(s)
    n = len(want_idx)
    tot = int(want_idx["rbsp_off"][-1] + want_idx["rbsp_len"][-1]) if n else 0
    d = torch.from_numpy(s).cuda()
    got_idx, got_arena, sm = ctx.index_extract(d)
    ok = (int(sm["error"]) == 0 and int(sm["stop_reason"]) == why and len(got_idx) == n
          and all(np.array_equal(got_idx[f], want_idx[f]) for f in ("start", "end", "rbsp_off", "rbsp_len", "status"))
          and np.array_equal(got_arena[:tot], want_arena[:tot]))
    k_scan = ctx.last_kernel()
    if not ok:
        bad += 1
        print("SCAN MISMATCH iter", it, "size", size, "kernel", k_scan, flush=True)
    del got_arena
    got_idx, _, sm = ctx.index_extract(d, want_rbsp=False)
    ok = (int(sm["error"]) == 0 and len(got_idx) == n and all(np.array_equal(got_idx[f], want_idx[f]) for f in ("start", "end", "rbsp_off", "rbsp_len", "status")))
    k_idx = ctx.last_kernel()
    if not ok:
        bad += 1
        print("INDEX-ONLY MISMATCH iter", it, "size", size, "kernel", k_idx, flush=True)
    del d
    by_tiles = -1
    # every NAL of the index, the rejected ones too (their bytes are in the arena; rbsp_to_nal takes any bytes): back to back in
    # the arena, which is what the arena tiles need -- every third stream only the accepted ones (gaps in the arena: by NALs)
    keep = want_idx.copy()
    if n and it % 3 == 2:
        keep = want_idx[(want_idx["status"] & 1) == 0]
    else:
        keep["status"] = 0
    if len(keep):
        want_stream = orc.emit_annexb(want_arena, keep)
        got, _ = ctx.emit_annexb(torch.from_numpy(want_arena).cuda(), keep)
        by_tiles = ctx.lib.hbs_ctx_last_emit_by_tiles(ctx.h)
        if len(got) != len(want_stream) or not np.array_equal(got, want_stream):
            bad += 1
            print("EMIT MISMATCH iter", it, "arena", len(want_arena), "nals", len(keep), "by tiles", by_tiles, flush=True)
        del got, want_stream
    print("iter %d: %.3f GiB (size %% 196608 = %d), %d NALs, stop %d, kernels %d / %d, emit by tiles %d, %.1f s" %
          (it, size / 2**30, size % 196608, n, why, k_scan, k_idx, by_tiles, time.time() - t0), flush=True)
    del s, want_arena, want_idx
    torch.cuda.empty_cache()
    it += 1
print("iterations", it, "mismatches", bad)
