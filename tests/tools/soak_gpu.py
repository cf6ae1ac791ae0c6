"""Randomised soak of the byte kernels against the oracle (dev aid): scan + extract (automatic kernel choice and each
pinned kernel) and RBSP -> Annex-B on streams with mixed zero density, NAL sizes from a few bytes to MiBs, tile-edge
alignments.  usage: python3 tests/tools/soak_gpu.py [seconds] [seed]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import hevcbitstream_amd as hbs
from tests import _orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
orc = _orc.oracle()
ctxs = {v: hbs.Context(0) for v in (0, 2, 4, 6)}       # 6: the event-sparse kernel's 24-row geometry (round 6)
for v, c in ctxs.items():
    c.set_kernel(v)
    c.set_count_ahead(2)              # round 5: the event-sparse kernel's dense tiles counted ahead at every size (default: from 3 GiB)
ctx5 = hbs.Context(0)
ctx5.set_kernel(5)
ctx_tiles = hbs.Context(0)
ctx_tiles.set_emit_path(2)            # the arena-tile emit kernel whenever the index allows it, whatever the size
ALPHA = np.array([0, 0, 0, 0, 1, 1, 2, 3, 3, 4, 0x40, 0x80, 0xFF], dtype=np.uint8)


def make_stream(rng):
    """regions of different character glued together"""
    parts = []
    total = int(rng.choice([3_000, 70_000, 200_000, 1_000_000, 5_000_000, 20_000_000]))
    while sum(len(p) for p in parts) < total:
        kind = rng.integers(0, 6)
        n = int(rng.integers(1, max(2, total // 3)))
        if kind == 0:
            p = rng.integers(1, 256, size=n, dtype=np.uint8)                      # no zeros at all
        elif kind == 1:
            p = rng.integers(0, 256, size=n, dtype=np.uint8)                      # coded-video-like
        elif kind == 2:
            p = ALPHA[rng.integers(0, len(ALPHA), size=n)]                         # dense in patterns
        elif kind == 3:
            p = np.zeros(n, dtype=np.uint8)                                        # a run of zeros
        elif kind == 4:
            p = rng.integers(0, 256, size=n, dtype=np.uint8); p[rng.random(n) < 0.03] = 0
        else:
            p = np.tile(np.array([0, 0, 3, 0, 0, 1, 9], dtype=np.uint8), n // 7 + 1)[:n]
        parts.append(p)
        if rng.random() < 0.7:
            parts.append(np.array([0, 0, 0, 1] if rng.random() < 0.5 else [0, 0, 1], dtype=np.uint8))
    s = np.concatenate(parts)
    # start codes near multiples of the tile sizes
    for edge in (65536, 196608):
        for m in range(edge, len(s) - 8, edge * int(rng.integers(1, 4))):
            o = m + int(rng.integers(-5, 3))
            s[o:o + 4] = (0, 0, 1, int(rng.integers(1, 255)))
    return s


t_end = time.time() + budget
it = bad = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed0 * 100003 + it)
    s = make_stream(rng)
    want_idx, want_arena, why = orc.index_extract(s)
    d = torch.from_numpy(s).cuda()
    tot = int(want_idx["rbsp_off"][-1] + want_idx["rbsp_len"][-1]) if len(want_idx) else 0
    for v, c in ctxs.items():
        got_idx, got_arena, sm = c.index_extract(d)
        ok = (int(sm["error"]) == 0 and int(sm["stop_reason"]) == why and len(got_idx) == len(want_idx)
              and all(np.array_equal(got_idx[f], want_idx[f]) for f in ("start", "end", "rbsp_off", "rbsp_len", "status"))
              and np.array_equal(got_arena[:tot], want_arena[:tot]))
        if not ok:
            bad += 1
            print("SCAN MISMATCH kernel", v, "iter", it, "len", len(s))
    # the scan alone (no arena): the streaming kernel, pinned and through the automatic choice
    for v in (5, 0):
        c = ctx5 if v == 5 else ctxs[0]
        got_idx, _, sm = c.index_extract(d, want_rbsp=False)
        ok = (int(sm["error"]) == 0 and int(sm["stop_reason"]) == why and len(got_idx) == len(want_idx)
              and all(np.array_equal(got_idx[f], want_idx[f]) for f in ("start", "end", "rbsp_off", "rbsp_len", "status")))
        if not ok:
            bad += 1
            print("INDEX-ONLY MISMATCH kernel", v, "iter", it, "len", len(s))
    # emit what was extracted (accepted NALs only) and compare with the oracle's rbsp_to_nal loop
    if len(want_idx):
        keep = want_idx[(want_idx["status"] & 1) == 0]
        if len(keep):
            got, ent = ctxs[0].emit_annexb(torch.from_numpy(want_arena.copy()).cuda(), keep)
            want_stream = orc.emit_annexb(want_arena, keep)
            if not np.array_equal(got, want_stream):
                bad += 1
                print("EMIT MISMATCH iter", it, "nals", len(keep))
            got_t, ent_t = ctx_tiles.emit_annexb(torch.from_numpy(want_arena.copy()).cuda(), keep)
            if not np.array_equal(got_t, want_stream) or not np.array_equal(ent_t, ent):
                bad += 1
                print("EMIT (arena tiles) MISMATCH iter", it, "nals", len(keep))
            # ... and with every NAL kept (back to back in the arena: what the tile kernel is for), synthetic 3- / 4-byte start codes
            full = want_idx.copy()
            full["status"] = 0
            a, ea = ctxs[0].emit_annexb(torch.from_numpy(want_arena.copy()).cuda(), full, gap_mode=1)
            b, eb = ctx_tiles.emit_annexb(torch.from_numpy(want_arena.copy()).cuda(), full, gap_mode=1)
            if not np.array_equal(a, b) or not np.array_equal(ea, eb):
                bad += 1
                print("EMIT (arena tiles, all NALs) MISMATCH iter", it, "nals", len(full))
    it += 1
print("iterations", it, "mismatches", bad)
