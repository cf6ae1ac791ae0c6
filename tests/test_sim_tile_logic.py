"""CPU single-step of the product's per-tile device logic (hbs_tile.h) against
the oracle: window rules, end-of-stream rules, scan algebra and the gather.
The same comparisons run on the real kernels in test_gpu_scan.py (-m gpu)."""
import numpy as np
import pytest

from tests import _sim

ALPHA = np.array([0, 0, 0, 0, 1, 1, 2, 3, 3, 4, 0x40, 0x80, 0xFF], dtype=np.uint8)


@pytest.fixture(autouse=True, params=[2, 3, 4], ids=["lds-image", "registers", "sparse"])
def kernel_variant(request):
    """both kernels' per-thread logic: hbs_tile.h (64-byte blocks over an LDS image) and
    hbs_chunk.h (16-byte chunks in registers), and hbs_sparse.h (flagged chunks as elements, gaps)"""
    old = _sim.VARIANT
    _sim.VARIANT = request.param
    yield
    _sim.VARIANT = old


def check(orc, stream):
    stream = np.ascontiguousarray(stream, dtype=np.uint8)
    want_idx, want_arena, why = orc.index_extract(stream)
    got_idx, got_arena, s = _sim.index_extract(stream)
    hx = bytes(stream[:200]).hex()
    assert int(s["stop_reason"]) == why, hx
    assert len(got_idx) == len(want_idx), hx
    for f in ("start", "end", "rbsp_off", "rbsp_len", "status"):
        assert np.array_equal(got_idx[f], want_idx[f]), (f, hx, got_idx[f][:8], want_idx[f][:8])
    # arena bytes of the NALs the reference loop visits (everything before a truncation point)
    tot = int(want_idx["rbsp_off"][-1] + want_idx["rbsp_len"][-1]) if len(want_idx) else 0
    assert np.array_equal(got_arena[:tot], want_arena[:tot]), hx
    assert int(s["error"]) == 0


def test_short_buffers_dense(orc):
    rng = np.random.RandomState(1)
    for _ in range(6000):
        n = rng.randint(0, 48)
        check(orc, ALPHA[rng.randint(0, len(ALPHA), size=n)])


def test_short_buffers_thin(orc):
    rng = np.random.RandomState(2)
    for _ in range(3000):
        n = rng.randint(1, 300)
        s = ALPHA[rng.randint(0, len(ALPHA), size=n)].copy()
        s[rng.rand(n) < 0.6] = 0x77
        check(orc, s)


def test_tail_patterns_exhaustive(orc):
    """every combination of {00,01,03,55} in the last 7 bytes behind a NAL."""
    vals = [0, 1, 3, 0x55]
    head = np.array([0, 0, 1, 0x40, 0x41, 0x42], dtype=np.uint8)
    for code in range(4 ** 7):
        tail = [vals[(code >> (2 * i)) & 3] for i in range(7)]
        check(orc, np.concatenate([head, np.array(tail, dtype=np.uint8)]))


@pytest.mark.parametrize("mode", [0, 1])
def test_synthetic_stream(orc, mode):
    stream, idx, arena = orc.gen_stream(0x1234, 60, mode)
    got_idx, got_arena, s = _sim.index_extract(stream)
    assert np.array_equal(got_idx, idx)
    assert np.array_equal(got_arena, arena)
    assert int(s["stop_reason"]) == -1 and int(s["nal_count"]) == 60


def test_tile_boundaries(orc):
    """patterns straddling 64-byte block, 256-byte thread and 64 KiB tile boundaries."""
    rng = np.random.RandomState(3)
    pats = [bytes([0, 0, 1]), bytes([0, 0, 0, 1]), bytes([0, 0, 3]), bytes([0, 0, 3, 0, 0, 3]), bytes([0, 0, 0]),
            bytes([0, 0, 2]), bytes([0, 0, 3, 9]), bytes([0] * 9)]
    for trial in range(80):
        n = 65536 * 2 + rng.randint(0, 200)
        s = rng.randint(4, 256, size=n).astype(np.uint8)
        s[0:4] = [0, 0, 1, 0x40]
        for edge in (64, 128, 256, 512, 16384, 65536, 65536 + 64, 65536 + 256, 131072):
            for _ in range(2):
                p = pats[rng.randint(len(pats))]
                at = edge - rng.randint(0, len(p) + 2)
                if at >= 4 and at + len(p) <= n:
                    s[at:at + len(p)] = np.frombuffer(p, dtype=np.uint8)
        check(orc, s)


def test_zero_runs_and_garbage(orc):
    rng = np.random.RandomState(4)
    for _ in range(100):
        parts = []
        for _k in range(rng.randint(1, 6)):
            parts.append(np.array([0] * rng.randint(2, 5) + [1], dtype=np.uint8))
            parts.append(rng.randint(4, 256, size=rng.randint(1, 3000)).astype(np.uint8))
            if rng.rand() < 0.5:
                parts.append(np.zeros(rng.randint(3, 400), dtype=np.uint8))
            if rng.rand() < 0.3:
                parts.append(rng.randint(4, 256, size=rng.randint(1, 50)).astype(np.uint8))  # garbage outside NALs
        check(orc, np.concatenate(parts))


def test_capacity_clip(orc):
    stream, idx, arena = orc.gen_stream(7, 12, 0)
    got_idx, _, s = _sim.index_extract(stream, index_cap=5)
    assert int(s["error"]) == -4 and int(s["nal_found"]) == 12 and len(got_idx) == 5
    assert np.array_equal(got_idx["start"], idx["start"][:5])
