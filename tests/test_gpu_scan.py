"""Parity tests proper for K12 (scan + index + RBSP extraction): the HIP path,
called through the C ABI, against the oracle and the golden fixtures."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ALPHA = np.array([0, 0, 0, 0, 1, 1, 2, 3, 3, 4, 0x40, 0x80, 0xFF], dtype=np.uint8)


@pytest.fixture(scope="module", params=[0, 2, 4, 5, 6],
                ids=["automatic", "lds-image-kernel", "sparse-kernel", "index-only-passes", "sparse-kernel-24-rows"])
def ctx(request):
    """the implementations of the fused kernel (hbs_scan.hip 2, hbs_scan4.hip 4 and its 24-row geometry hbs_scan4_r24.hip 6),
    the index-only streaming kernel (hbs_scan5.hip) and the automatic choice between them (the default)"""
    import torch
    import hevcbitstream_amd as hbs
    assert torch.cuda.is_available()
    c = hbs.Context(0)
    c.set_kernel(request.param)
    yield c
    c.close()


def run(ctx, stream, **kw):
    import torch
    stream = np.ascontiguousarray(stream, dtype=np.uint8)
    d = torch.from_numpy(stream).cuda() if len(stream) else torch.empty(0, dtype=torch.uint8, device="cuda")
    return ctx.index_extract(d, **kw)


def check(ctx, orc, stream):
    want_idx, want_arena, why = orc.index_extract(stream)
    got_idx, got_arena, s = run(ctx, stream)
    hx = bytes(np.asarray(stream[:64], dtype=np.uint8)).hex()
    assert int(s["error"]) == 0, s
    assert int(s["stop_reason"]) == why, hx
    assert len(got_idx) == len(want_idx), hx
    for f in ("start", "end", "rbsp_off", "rbsp_len", "status"):
        assert np.array_equal(got_idx[f], want_idx[f]), (f, hx)
    tot = int(want_idx["rbsp_off"][-1] + want_idx["rbsp_len"][-1]) if len(want_idx) else 0
    assert np.array_equal(got_arena[:tot], want_arena[:tot]), hx


def test_ten_nal_golden(ctx):
    stream = np.fromfile(os.path.join(HERE, "golden", "ten_nal.hevc"), dtype=np.uint8)
    want = json.load(open(os.path.join(HERE, "golden", "ten_nal.index.json")))
    idx, arena, s = run(ctx, stream)
    assert [[int(a), int(b)] for a, b in zip(idx["start"], idx["end"])] == [w[:2] for w in want]
    assert int(s["stop_reason"]) == -1


def test_find_golden_first_nal(ctx):
    """first NAL of every golden find_nal_unit vector (reference answers)."""
    gold = json.load(open(os.path.join(HERE, "golden", "l2_vectors.json")))
    for hx, (r, st, en) in gold["find"][:120]:
        idx, _, s = run(ctx, np.frombuffer(bytes.fromhex(hx), dtype=np.uint8))
        if r > 0 or r == -1:
            assert len(idx) >= 1 and int(idx["start"][0]) == st and int(idx["end"][0]) == en, hx
        elif st == 0 and en == 0:
            assert len(idx) == 0 and int(s["stop_reason"]) == 0, hx
        else:   # empty NAL: loop stops there
            assert len(idx) == 0 and int(s["stop_reason"]) == 1, hx


def test_empty_and_tiny(ctx, orc):
    for n in range(0, 9):
        check(ctx, orc, np.zeros(n, dtype=np.uint8))
        check(ctx, orc, np.array(([0, 0, 1] * 3)[:n], dtype=np.uint8))


def test_fuzz_small(ctx, orc):
    rng = np.random.RandomState(5)
    for _ in range(150):
        n = rng.randint(0, 400)
        s = ALPHA[rng.randint(0, len(ALPHA), size=n)].copy()
        s[rng.rand(n) < 0.5] = 0x77
        check(ctx, orc, s)


def test_fuzz_multi_tile(ctx, orc):
    """many 64 KiB tiles: exercises the look-back chain under real concurrency."""
    rng = np.random.RandomState(6)
    for trial in range(6):
        n = rng.randint(4 << 20, 12 << 20)
        s = rng.randint(0, 256, size=n).astype(np.uint8)
        # sprinkle start codes / EPBs / zero runs / errors
        for pat, cnt in ((b"\x00\x00\x01", n // 5000), (b"\x00\x00\x00\x01", n // 9000), (b"\x00\x00\x03", n // 700),
                         (b"\x00\x00\x00", n // 40000), (b"\x00\x00\x02", n // 200000 + 1), (b"\x00" * 70, 3)):
            for at in rng.randint(0, n - 80, size=cnt):
                s[at:at + len(pat)] = np.frombuffer(pat, dtype=np.uint8)
        check(ctx, orc, s)


@pytest.mark.parametrize("mode", [0, 1])
def test_synthetic_64mib(ctx, orc, mode):
    stream, idx, arena = orc.gen_stream(0x1234 + mode, 6400, mode)
    got_idx, got_arena, s = run(ctx, stream)
    assert int(s["error"]) == 0 and int(s["stop_reason"]) == -1
    assert np.array_equal(got_idx, idx)
    assert np.array_equal(got_arena, arena)


def test_tile_edges(ctx, orc):
    rng = np.random.RandomState(8)
    pats = [bytes([0, 0, 1]), bytes([0, 0, 0, 1]), bytes([0, 0, 3]), bytes([0, 0, 3, 0, 0, 3]), bytes([0, 0, 0]),
            bytes([0, 0, 2]), bytes([0, 0, 3, 9]), bytes([0] * 9)]
    for trial in range(40):
        n = 65536 * 3 + rng.randint(0, 200)
        s = rng.randint(4, 256, size=n).astype(np.uint8)
        s[0:4] = [0, 0, 1, 0x40]
        for edge in (64, 128, 256, 4096, 16384, 65536, 65536 + 64, 65536 + 256, 131072, 196608):
            for _ in range(2):
                p = pats[rng.randint(len(pats))]
                at = edge - rng.randint(0, len(p) + 2)
                if at >= 4 and at + len(p) <= n:
                    s[at:at + len(p)] = np.frombuffer(p, dtype=np.uint8)
        check(ctx, orc, s)


def test_index_only_and_capacity(ctx, orc):
    stream, idx, arena = orc.gen_stream(7, 300, 1)
    got_idx, got_arena, s = run(ctx, stream, want_rbsp=False)
    assert got_arena is None and np.array_equal(got_idx, idx)
    got_idx, _, s = run(ctx, stream, index_cap=100)
    assert int(s["error"]) == -4 and int(s["nal_found"]) == 300 and len(got_idx) == 100
    assert np.array_equal(got_idx["start"], idx["start"][:100])


def test_index_only_streams(ctx, orc):
    """no RBSP arena asked for (find_nal_unit over a stream): on kernel 5 and in automatic mode the flags pass + element
    pass of hbs_scan5.hip answer -- several 1 MiB tiles, ends cut inside a chunk, regions dense in zero pairs (more than
    64 elements per tile), zeros and start codes across tile edges"""
    rng = np.random.RandomState(41)
    for rep, size in enumerate([1 << 20, (1 << 20) + 7, 3 * (1 << 20) + 1234567, 70001, 5_000_013]):
        s = rng.randint(0, 256, size=size).astype(np.uint8)
        if rep % 2 == 0:
            s[rng.rand(size) < 0.02] = 0                       # many elements per tile
        for m in range(1 << 20, size - 8, 1 << 20):           # patterns across the 1 MiB tile edges
            o = m + int(rng.randint(-4, 2))
            s[o:o + 5] = (0, 0, 0, 1, 0x42) if rep % 2 else (0, 0, 1, 0, 0)
        dense = ALPHA[rng.randint(0, len(ALPHA), size=min(size // 8, 200_000))]
        a = int(rng.randint(0, size - len(dense)))
        s[a:a + len(dense)] = dense                            # a region where nearly every chunk is an element
        want_idx, _, why = orc.index_extract(s)
        got_idx, got_arena, sm = run(ctx, s, want_rbsp=False)
        assert got_arena is None and int(sm["error"]) == 0 and int(sm["stop_reason"]) == why, (rep, size)
        assert len(got_idx) == len(want_idx), (rep, size, len(got_idx), len(want_idx))
        for f in ("start", "end", "rbsp_off", "rbsp_len", "status"):
            assert np.array_equal(got_idx[f], want_idx[f]), (rep, size, f)


_PARTS_WORKER = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import torch, hevcbitstream_amd as hbs
from tests import _orc
orc = _orc.oracle()
ctx = hbs.Context(0)
ctx.set_kernel(5)
rng = np.random.RandomState(5)
n = 9 * (1 << 20) + 4321
base = rng.randint(1, 256, size=n).astype(np.uint8)
at = 0
while at + 8 < n:
    base[at:at + 4] = (0, 0, 1, 0x40)
    at += int(rng.randint(2000, 30000))
pats = [b"\x00\x00\x03", b"\x00", b"\x00\x00\x01\x42\x55", b"\x00\x00\x03\x00\x00\x02\x01", b"\x00\x00\x03\x04"]
for k, (a, b) in enumerate([(300_000, 1_400_000), (2_000_000, 2_070_000), (3_100_000, 5_900_000), (7_000_000, 7_000_000 + 33_000), (n - 700_000, n)]):
    s = base.copy()
    p = np.frombuffer(pats[k %% len(pats)], dtype=np.uint8)
    s[a:b] = np.tile(p, (b - a) // len(p) + 1)[:b - a]
    want, _, why = orc.index_extract(s)
    for cut in (0, 5, 1 << 20):
        t = s[:len(s) - cut]
        if cut:
            want, _, why = orc.index_extract(t)
        got, arena, sm = ctx.index_extract(torch.from_numpy(np.ascontiguousarray(t)).cuda(), want_rbsp=False)
        assert int(sm["error"]) == 0 and int(sm["stop_reason"]) == why and len(got) == len(want), (k, cut, len(got), len(want))
        for f in ("start", "end", "rbsp_off", "rbsp_len", "status"):
            assert np.array_equal(got[f], want[f]), (k, cut, f)
print("ok")
"""


@pytest.mark.parametrize("rows", [0, 64, 96, 256, 512], ids=["height-of-the-call", "64", "96", "256", "512"])
def test_index_only_tiles_walked_by_rows_in_parts(rows):
    """round 5: a tile the index-only scan walks by rows (a stretch of padding, zeros, tiny NALs, patterns that mark errors) is
    walked again by the emit pass in parts of 32 rows, by helper wavefronts, from the aggregates the stream pass left in front of
    every part -- at the tile height the call picks and at pinned ones (HBS5_TILE_ROWS is read once: a process each); stretches
    that begin and end inside tiles, cover whole tiles, and run to the stream's end; ends cut inside a part."""
    import subprocess
    import sys
    env = dict(os.environ)
    if rows:
        env["HBS5_TILE_ROWS"] = str(rows)
    r = subprocess.run([sys.executable, "-c", _PARTS_WORKER % os.path.dirname(HERE)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-2000:], r.stderr[-4000:])


def test_index_only_ring_of_flagged_chunks(ctx, orc):
    """round 4: the index-only scan leaves the flagged chunks' bytes in a ring of 192 entries in LDS and walks them 64 at a time
    inside its streaming loop.  Streams that fill, wrap and overflow the ring: NALs of 512 bytes and of 1 KiB (several batches
    per 256 KiB tile, more elements than a tile used to record), and tiles that are sparse but for ONE 16 KiB span with
    ~150 start codes in it while 40-70 elements still wait (the span does not fit: the tile's elements are redone from its
    flag words) -- at several offsets of that span in the tile.  Without an arena (kernel 5 and the automatic mode take the
    index-only kernels; the others their own) and with one."""
    rng = np.random.RandomState(58)
    T = 256 * 1024

    def put_nals(s, lo, hi, step_lo, step_hi):
        at = lo
        while at + 8 < hi:
            sc = (0, 0, 1) if rng.randint(3) else (0, 0, 0, 1)
            s[at:at + len(sc)] = sc
            s[at + len(sc)] = 0x40
            at += int(rng.randint(step_lo, step_hi))

    def both(s):
        want_idx, _, why = orc.index_extract(s)
        got_idx, got_arena, sm = run(ctx, s, want_rbsp=False)
        assert got_arena is None and int(sm["error"]) == 0 and int(sm["stop_reason"]) == why
        assert len(got_idx) == len(want_idx), (len(got_idx), len(want_idx))
        for f in ("start", "end", "rbsp_off", "rbsp_len", "status"):
            assert np.array_equal(got_idx[f], want_idx[f]), f
        check(ctx, orc, s)

    for mean in (512, 1024):
        n = 5 * T + 12345
        s = rng.randint(1, 256, size=n).astype(np.uint8)
        put_nals(s, 0, n, mean * 3 // 4, mean * 5 // 4)
        both(s)
    for waiting in (40, 50, 57, 63, 70):
        for span in (1, 7, 15):
            n = 3 * T + 999
            s = rng.randint(1, 256, size=n).astype(np.uint8)
            s[0:4] = (0, 0, 1, 0x40)
            base = T                                              # the second tile
            step = max(64, (span * 16384) // waiting)
            put_nals(s, base + 100, base + span * 16384 - 64, step, step + 1)       # ~`waiting` elements in front of the span
            put_nals(s, base + span * 16384, base + (span + 1) * 16384, 100, 120)   # ~150 in the span
            put_nals(s, base + (span + 1) * 16384 + 3000, n, 20000, 40000)
            both(s)


def test_repeatable(ctx, orc):
    """same context, back-to-back calls (descriptor workspace is reused)."""
    stream, idx, arena = orc.gen_stream(99, 500, 0)
    for _ in range(3):
        got_idx, got_arena, s = run(ctx, stream)
        assert np.array_equal(got_idx, idx) and np.array_equal(got_arena, arena)


@pytest.mark.parametrize("pattern", [b"\x00", b"\x00\x00\x03", b"\x00\x00\x01\x42", b"\x00\x00\x00\x01\x26\x01\xaf"],
                         ids=["zeros", "epb-run", "start-codes", "tiny-nals"])
def test_dense_patterns(ctx, orc, pattern):
    """streams in which every 16-byte chunk is touched by a pattern: the event-sparse kernel takes the
    elements of a tile in batches (and finds no plain chunk at all), the others run as usual."""
    n = (1 << 20) + 12345
    s = np.tile(np.frombuffer(pattern, dtype=np.uint8), n // len(pattern) + 1)[:n].copy()
    check(ctx, orc, s)
    s[n // 2:n // 2 + 4000] = 0x55          # a plain stretch in the middle of the dense stream
    check(ctx, orc, s)


@pytest.mark.parametrize("shape", ["nals-500", "nals-1k-epb", "zeros-10-percent", "zeros-and-starts"])
def test_tiles_with_several_batches_of_elements(ctx, orc, shape):
    """65 to 512 elements per 192 KiB tile (round 3: exact flags, per-wavefront element lists, two element wavefronts, the batch
    aggregates and records in LDS): streams of small NALs, NALs full of emulation prevention bytes, zero-heavy bytes."""
    rng = np.random.RandomState({"nals-500": 11, "nals-1k-epb": 12, "zeros-10-percent": 13, "zeros-and-starts": 14}[shape])
    n = 3 * (1 << 20) + 4321
    if shape.startswith("nals"):
        s = rng.randint(1, 256, size=n).astype(np.uint8)
        lo, hi = (300, 700) if shape == "nals-500" else (700, 1400)
        at = 0
        while at + 8 < n:
            sc = (0, 0, 1) if rng.randint(3) else (0, 0, 0, 1)
            s[at:at + len(sc)] = sc
            s[at + len(sc)] = 0x40
            at += int(rng.randint(lo, hi))
        if shape == "nals-1k-epb":
            for p in rng.randint(8, n - 8, size=n // 500):
                s[p:p + 4] = (0, 0, 3, rng.randint(0, 4))
    else:
        s = rng.randint(1, 256, size=n).astype(np.uint8)
        s[rng.random_sample(n) < 0.10] = 0
        s[0:4] = (0, 0, 1, 0x40)
        if shape == "zeros-and-starts":
            for p in rng.randint(8, n - 8, size=n // 3000):
                s[p:p + 4] = (0, 0, 1, 0x42)
    check(ctx, orc, s)


@pytest.mark.parametrize("kernel", [0, 4], ids=["automatic", "sparse-kernel"])
def test_dense_tiles_counted_ahead(orc, kernel):
    """round 5: the event-sparse kernel's dense tiles counted ahead of it (hbs_ctx_set_count_ahead; by default only from 3 GiB up,
    here on every stream): stretches of padding / zeros / tiny NALs in a sparse stream -- inside a tile, across tiles, at the
    stream's start and in its last tile; a tile the sample marks and which is not dense (zero pairs exactly where the sample
    looks), one that is walked ahead and still is not (400 flagged chunks); a dense tile the sample misses (dense everywhere but there).  Modes 0 and 2 must give what the oracle gives."""
    import hevcbitstream_amd as hbs
    tile = 192 << 10
    rng = np.random.RandomState(77)
    n = 17 * tile + 7777
    base = rng.randint(1, 256, size=n).astype(np.uint8)
    at = 0
    while at + 8 < n:
        base[at:at + 4] = (0, 0, 1, 0x40)
        at += int(rng.randint(3000, 20000))
    pats = [b"\x00\x00\x03", b"\x00", b"\x00\x00\x01\x42\x55", b"\x00\x00\x03\x00\x00\x03\x01"]

    def fill(s, a, b, k):
        p = np.frombuffer(pats[k % len(pats)], dtype=np.uint8)
        s[a:b] = np.tile(p, (b - a) // len(p) + 1)[:b - a]

    streams = []
    s = base.copy(); fill(s, 2 * tile + 100, 3 * tile - 100, 0); streams.append(s)                      # inside one tile
    s = base.copy(); fill(s, 4 * tile + 50_000, 8 * tile + 20_000, 1); streams.append(s)                # across tiles, zeros
    s = base.copy(); fill(s, 0, tile + 999, 2); fill(s, 16 * tile, n, 3); streams.append(s)             # first tile(s), last tiles
    s = base.copy(); fill(s, 15 * tile - 5, 17 * tile + 5, 0); streams.append(s)                        # up to the last full tile and beyond
    s = base.copy(); fill(s, 12 * tile - 9000, 13 * tile + 9000, 0); streams.append(s)                  # 9 KB of the tiles on either side: dense, marked by their edge
    s = base.copy(); fill(s, 7 * tile + 70_000, 7 * tile + 130_000, 1); streams.append(s)               # 60 KB inside a tile: the first look may miss it
    s = base.copy()                                                                                      # marked, not dense
    for t in (3, 9):
        for lane in range(48):
            o = t * tile + lane * 4096 + 2048
            s[o:o + 16] = (0, 0, 3, 1) * 4
    streams.append(s)
    s = base.copy()                                                                                      # marked and walked ahead, not dense
    for t in (2, 10):
        for k in range(400):
            o = t * tile + k * 480
            s[o:o + 4] = (0, 0, 3, 1)
        for lane in range(0, 48, 3):
            o = t * tile + lane * 4096 + 2048
            s[o:o + 32] = (0, 0, 3, 1) * 8
    streams.append(s)
    s = base.copy()                                                                                      # dense, not marked
    for t in (5, 6):
        fill(s, t * tile, (t + 1) * tile, 0)
        for k in range(48):
            s[t * tile + k * 4096 + 2048 - 8:t * tile + k * 4096 + 2048 + 24] = 0x77
    streams.append(s)
    s = base.copy()                                                                                      # every tile dense
    fill(s, 0, n, 0); s[0:4] = (0, 0, 1, 0x40); streams.append(s)
    for mode in (2, 0):
        c = hbs.Context(0)
        c.set_kernel(kernel)
        c.set_count_ahead(mode)
        try:
            for s in streams:
                check(c, orc, s)
                check(c, orc, s[:len(s) - 3 * tile - 13])       # another tile count, another last tile
        finally:
            c.close()


def test_automatic_choice_follows_density(orc):
    """the default mode picks the event-sparse kernel for coded-video-like bytes, its 24-row geometry for streams of small NALs
    (one chunk in ~13 an element) and the LDS-image kernel for zero-heavy ones, on the device; the answer is the oracle's each time"""
    import hevcbitstream_amd as hbs
    c = hbs.Context(0)
    try:
        assert c.kernel() == 0
        rng = np.random.default_rng(77)
        n = 3 << 20
        sparse = rng.integers(1, 256, size=n, dtype=np.uint8)
        for p in range(5000, n - 8, 9973):
            sparse[p:p + 4] = (0, 0, 1, 0x40)
        small = rng.integers(1, 256, size=n, dtype=np.uint8)             # NALs of 200-300 bytes, 3- and 4-byte start codes
        p = 7
        while p < n - 8:
            small[p:p + 4] = (0, 0, 1, 0x40) if (p & 1) else (0, 0, 0, 1)
            p += int(rng.integers(200, 300))
        dense = ALPHA[rng.integers(0, len(ALPHA), size=n)]
        for stream, want in ((sparse, 4), (small, 6), (dense, 2), (sparse, 4), (small, 6)):
            check(c, orc, stream)
            assert c.last_kernel() == want
    finally:
        c.close()


@pytest.mark.parametrize("mean", [100, 128, 200, 330, 450])
def test_streams_of_small_nals_on_the_24_row_geometry(orc, mean):
    """round 6: 40 MiB of NALs of `mean` +- 25 % bytes (start codes of 3 and 4 bytes, some emulation prevention bytes, a few runs of
    zeros that make single tiles pass 1024 elements) through kernel 6 pinned, with and without an arena, and through the
    automatic mode: every entry and every RBSP byte against the oracle"""
    import hevcbitstream_amd as hbs
    rng = np.random.default_rng(1000 + mean)
    n = 40 << 20
    s = rng.integers(1, 256, size=n, dtype=np.uint8)
    s[rng.random(n) < 0.002] = 0                                   # stray zeros: pairs, EPB candidates
    p = 3
    while p < n - 8:
        if p & 2:
            s[p:p + 4] = (0, 0, 1, 0x42)
        else:
            s[p:p + 5] = (0, 0, 0, 1, 0x26)
        p += int(rng.integers(max(8, mean * 3 // 4), mean * 5 // 4 + 1))
    for q in rng.integers(0, n - 70000, size=6):                   # stretches that turn their tiles dense
        s[q:q + 20000:3] = 0
        s[q + 1:q + 20000:3] = 0
        s[q + 2:q + 20000:3] = 3
    want_idx, want_arena, why = orc.index_extract(s)
    tot = int(want_idx["rbsp_off"][-1] + want_idx["rbsp_len"][-1])
    c = hbs.Context(0)
    try:
        for kernel in (6, 0):
            c.set_kernel(kernel)
            got_idx, got_arena, sm = run(c, s, index_cap=len(want_idx) + 16)
            assert int(sm["error"]) == 0 and int(sm["stop_reason"]) == why and len(got_idx) == len(want_idx), (kernel, sm)
            for f in ("start", "end", "rbsp_off", "rbsp_len", "status"):
                assert np.array_equal(got_idx[f], want_idx[f]), (kernel, f)
            assert np.array_equal(got_arena[:tot], want_arena[:tot]), kernel
            got2, _, sm2 = run(c, s, index_cap=len(want_idx) + 16, want_rbsp=False)
            assert int(sm2["error"]) == 0 and len(got2) == len(want_idx)
            for f in ("start", "end", "rbsp_off", "rbsp_len", "status"):
                assert np.array_equal(got2[f], want_idx[f]), (kernel, "no arena", f)
    finally:
        c.close()


def test_last_kernel_reports_what_ran_without_an_arena():
    """arena-less calls in automatic mode: the streaming index-only kernel (5) from 1 GiB up, the register-tile kernel (4)
    below -- and hbs_ctx_last_kernel says so on both sides of the threshold (round 3's advice: it used its own, older
    threshold and reported 5 at 0.8 GiB where 4 had run).  Results of the two agree on the same bytes."""
    import torch
    import hevcbitstream_amd as hbs
    c = hbs.Context(0)
    try:
        g = c.synth_stream(0x51, 110_000)                                   # ~1.05 GiB of ~10 KiB NALs
        sb = g["stream_bytes"]
        assert sb > (1 << 30)
        cut = int(0.8 * (1 << 30)) & ~15
        got = {}
        for nbytes, want in ((cut, 4), (sb, 5), (cut, 4)):
            index, _, summary, cap = c.alloc_outputs(nbytes, index_cap=120_000, want_rbsp=False)
            c.index_extract_async(g["stream"][:nbytes], index, cap, None, summary)
            s = c.read_summary(summary)
            assert int(s["error"]) == 0
            assert c.last_kernel() == want, (nbytes, c.last_kernel())
            got[nbytes] = (int(s["nal_count"]), index[: int(s["nal_count"]) * 32].clone())
        # kernel 4 pinned on the whole stream = kernel 5's automatic answer
        c.set_kernel(4)
        index, _, summary, cap = c.alloc_outputs(sb, index_cap=120_000, want_rbsp=False)
        c.index_extract_async(g["stream"][:sb], index, cap, None, summary)
        n = int(c.read_summary(summary)["nal_count"])
        assert c.last_kernel() == 4 and n == got[sb][0] == 110_000
        a = index[: n * 32].view(torch.int64).view(n, 4)
        b = got[sb][1].view(torch.int64).view(n, 4)
        assert torch.equal(a[:, :3], b[:, :3])
    finally:
        c.close()


def test_calls_capture_into_a_hip_graph(orc):
    """hbs_index_extract only enqueues (no allocation after a warm-up call of the same size, no host wait, the
    kernel choice made on the device): captured once in a HIP graph, it is replayed on other bytes in the same
    buffers -- zero-heavy, then sparse again, so a replay takes the other kernel -- and stays exact."""
    import torch
    import hevcbitstream_amd as hbs
    from tests._orc import NAL_ENTRY
    from hevcbitstream_amd.api import SUMMARY
    ctx = hbs.Context(0)
    n = 3_000_000
    rng = np.random.RandomState(91)

    def make(zero_heavy):
        s = rng.randint(1, 256, size=n).astype(np.uint8)
        if zero_heavy:
            s[rng.rand(n) < 0.12] = 0
        for p in rng.randint(0, n - 8, size=300):
            s[p:p + 4] = (0, 0, 1, 0x42)
        return s

    first, second = make(False), make(True)
    d_stream = torch.from_numpy(first).cuda()
    index, rbsp, summary, cap = ctx.alloc_outputs(n)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        ctx.index_extract_async(d_stream, index, cap, rbsp, summary)        # warm-up: workspaces get their size
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            ctx.index_extract_async(d_stream, index, cap, rbsp, summary)
    for data in (second, first):
        d_stream.copy_(torch.from_numpy(data))
        g.replay()
        torch.cuda.synchronize()
        want_idx, want_arena, why = orc.index_extract(data)
        sm = np.frombuffer(summary.cpu().numpy().tobytes(), dtype=SUMMARY)[0]
        assert int(sm["error"]) == 0 and int(sm["nal_count"]) == len(want_idx) and int(sm["stop_reason"]) == why
        got = index[: len(want_idx) * 32].cpu().numpy().view(NAL_ENTRY)
        for f in ("start", "end", "rbsp_off", "rbsp_len", "status"):
            assert np.array_equal(got[f], want_idx[f]), f
        tot = int(want_idx["rbsp_off"][-1] + want_idx["rbsp_len"][-1]) if len(want_idx) else 0
        assert np.array_equal(rbsp[:tot].cpu().numpy(), want_arena[:tot])
    ctx.close()


def test_count_ahead_replayed_from_a_hip_graph(orc):
    """round 5: what the count-ahead keeps between calls (the call's number, the list of marked tiles, a word per tile) lives
    in device memory and is advanced by the call's last launch -- a captured call replayed on other bytes must not take the
    aggregates the previous replay left: stretches of padding that move from replay to replay, disappear, come back."""
    import torch
    import hevcbitstream_amd as hbs
    from tests._orc import NAL_ENTRY
    from hevcbitstream_amd.api import SUMMARY
    ctx = hbs.Context(0)
    ctx.set_kernel(4)
    ctx.set_count_ahead(2)
    tile = 192 << 10
    n = 14 * tile + 999
    rng = np.random.RandomState(92)
    base = rng.randint(1, 256, size=n).astype(np.uint8)
    at = 0
    while at + 8 < n:
        base[at:at + 4] = (0, 0, 1, 0x40)
        at += int(rng.randint(3000, 20000))

    def with_stretch(a, b, pat):
        s = base.copy()
        if b > a:
            p = np.frombuffer(pat, dtype=np.uint8)
            s[a:b] = np.tile(p, (b - a) // len(p) + 1)[:b - a]
        return s

    versions = [with_stretch(3 * tile + 5000, 6 * tile + 70_000, b"\x00\x00\x03"),
                with_stretch(3 * tile + 5000, 6 * tile + 70_000, b"\x00\x00\x03\x00\x00\x01\x41"),      # same tiles, other aggregates
                with_stretch(0, 0, b""),                                                                # nothing dense
                with_stretch(8 * tile - 30_000, 11 * tile + 10, b"\x00"),                             # elsewhere
                with_stretch(3 * tile + 5000, 6 * tile + 70_000, b"\x00\x00\x03")]
    d_stream = torch.from_numpy(versions[0]).cuda()
    index, rbsp, summary, cap = ctx.alloc_outputs(n)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        ctx.index_extract_async(d_stream, index, cap, rbsp, summary)        # warm-up: workspaces get their size
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            ctx.index_extract_async(d_stream, index, cap, rbsp, summary)
    for k, data in enumerate(versions[1:] + versions):
        d_stream.copy_(torch.from_numpy(data))
        g.replay()
        torch.cuda.synchronize()
        want_idx, want_arena, why = orc.index_extract(data)
        sm = np.frombuffer(summary.cpu().numpy().tobytes(), dtype=SUMMARY)[0]
        assert int(sm["error"]) == 0 and int(sm["nal_count"]) == len(want_idx) and int(sm["stop_reason"]) == why, k
        got = index[: len(want_idx) * 32].cpu().numpy().view(NAL_ENTRY)
        for f in ("start", "end", "rbsp_off", "rbsp_len", "status"):
            assert np.array_equal(got[f], want_idx[f]), (k, f)
        tot = int(want_idx["rbsp_off"][-1] + want_idx["rbsp_len"][-1]) if len(want_idx) else 0
        assert np.array_equal(rbsp[:tot].cpu().numpy(), want_arena[:tot]), k
    ctx.close()


def test_timing_ring_keeps_the_last_calls(ctx, orc):
    """hbs_ctx_kernel_ms_back: the event pairs of the last 64 timed calls stay readable (bench.py reads every step of its timed
    loop behind the loop's fence); calls further back, or before timing was enabled, are refused"""
    import torch
    import hevcbitstream_amd as hbs
    stream, _, _ = orc.gen_stream(0x55, 400, 0)
    d = torch.from_numpy(stream).cuda()
    index, rbsp, summary, cap = ctx.alloc_outputs(d.numel())
    ctx.enable_timing(True)
    try:
        with pytest.raises(hbs.HbsError):
            ctx.kernel_ms_back(0)                          # nothing timed yet
        for _ in range(5):
            ctx.index_extract_async(d, index, cap, rbsp, summary)
        torch.cuda.synchronize()
        ms = [ctx.kernel_ms_back(b) for b in range(5)]
        assert all(0.0 < x < 50.0 for x in ms), ms
        assert abs(ctx.kernel_ms() - ms[0]) < 1e-6         # the last call is slot 0
        with pytest.raises(hbs.HbsError):
            ctx.kernel_ms_back(5)                          # only five calls were timed
        for _ in range(70):
            ctx.index_extract_async(d, index, cap, rbsp, summary)
        torch.cuda.synchronize()
        assert all(0.0 < ctx.kernel_ms_back(b) < 50.0 for b in range(64))
        with pytest.raises(hbs.HbsError):
            ctx.kernel_ms_back(64)                         # the ring holds 64
    finally:
        ctx.enable_timing(False)


@pytest.mark.parametrize("exclusive", [0, 1], ids=["tickets", "device-exclusive"])
def test_two_contexts_scan_one_device_at_the_same_time(orc, exclusive):
    """round 5's advice: the persistent kernels of TWO contexts on ONE device, enqueued on two streams so that their workgroups
    share the CUs.  With every tile by ticket (the default) a workgroup only waits for tiles that a running workgroup has claimed,
    so both calls finish whatever the interleaving, exact; K3's tile kernel behind each scan the same.  hbs_ctx_set_device_exclusive
    is what a caller sets when it does NOT do this: one context at a time, first tiles by workgroup number, same results."""
    import torch
    import hevcbitstream_amd as hbs
    from hevcbitstream_amd.api import SUMMARY
    n_nals = 30_000                                          # ~300 MiB each: ~1 600 tiles over 512 workgroups, several rounds of tickets
    ctxs = [hbs.Context(0) for _ in range(2)]
    try:
        work = []
        for k, c in enumerate(ctxs):
            c.set_kernel(4)
            c.set_device_exclusive(exclusive)
            g = c.synth_stream(0x77 + k, n_nals, 0)
            sb, rb = g["stream_bytes"], g["rbsp_bytes"]
            index, rbsp, summary, cap = c.alloc_outputs(sb, index_cap=n_nals + 8)
            out = torch.empty(sb + 4096, dtype=torch.uint8, device="cuda")
            esum = torch.zeros(SUMMARY.itemsize, dtype=torch.uint8, device="cuda")
            work.append((c, g, sb, rb, index, rbsp, summary, cap, out, esum, torch.cuda.Stream()))
        torch.cuda.synchronize()
        for rep in range(3):
            if exclusive:
                # one context at a time (what the flag promises)
                for c, g, sb, rb, index, rbsp, summary, cap, out, esum, side in work:
                    c.index_extract_async(g["stream"][:sb], index, cap, rbsp, summary)
                    c.emit_annexb_async(rbsp, rb, index, n_nals, 1, out, None, esum)
                    torch.cuda.synchronize()
            else:
                for c, g, sb, rb, index, rbsp, summary, cap, out, esum, side in work:
                    with torch.cuda.stream(side):
                        c.index_extract_async(g["stream"][:sb], index, cap, rbsp, summary)
                        c.emit_annexb_async(rbsp, rb, index, n_nals, 1, out, None, esum)
                torch.cuda.synchronize()
            for c, g, sb, rb, index, rbsp, summary, cap, out, esum, side in work:
                s = c.read_summary(summary)
                assert int(s["error"]) == 0 and int(s["nal_count"]) == n_nals and int(s["rbsp_bytes"]) == rb, s
                assert torch.equal(rbsp[:rb], g["rbsp"][:rb])
                a = index[: n_nals * 32].view(torch.int64).view(n_nals, 4)
                b = g["index"][: n_nals * 32].view(torch.int64).view(n_nals, 4)
                assert torch.equal(a[:, :3], b[:, :3])
                es = c.read_summary(esum)
                assert int(es["error"]) == 0 and int(es["stream_bytes"]) == sb
                assert torch.equal(out[:sb], g["stream"][:sb])
        # the first stream through the oracle as well (the generator is the product's own)
        c, g, sb, rb, index, rbsp, summary, cap, out, esum, side = work[0]
        head = g["stream"][: 4 << 20].cpu().numpy()
        want_idx, want_arena, why = orc.index_extract(head)
        m = len(want_idx) - 1                                # (the last NAL of the cut is cut)
        got = index[: m * 32].cpu().numpy().view(hbs.NAL_ENTRY)
        for f in ("start", "end", "rbsp_off", "rbsp_len"):
            assert np.array_equal(got[f], want_idx[f][:m]), f
    finally:
        for c in ctxs:
            c.close()
