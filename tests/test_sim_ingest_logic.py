"""Windowed ingest (hbs_ingest.h: windows, carry-over of the unfinished NAL, skipping what is found
again) run over the CPU single-stepper, against the oracle's walk over the WHOLE stream."""
import numpy as np
import pytest

from tests import _sim

ALPHA = np.array([0, 0, 0, 0, 1, 1, 2, 3, 3, 4, 0x40, 0x80, 0xFF], dtype=np.uint8)


def check(orc, stream, window):
    stream = np.ascontiguousarray(stream, dtype=np.uint8)
    want_idx, want_arena, why = orc.index_extract(stream)
    rc, got_idx, got_arena, s = _sim.index_extract_windowed(stream, window)
    assert rc == 0 and int(s["error"]) == 0, (rc, s, window)
    assert int(s["stop_reason"]) == why, (window, bytes(stream[:64]).hex())
    assert len(got_idx) == len(want_idx), (window, len(got_idx), len(want_idx))
    for f in ("start", "end", "rbsp_off", "rbsp_len", "status"):
        assert np.array_equal(got_idx[f], want_idx[f]), (f, window)
    tot = int(want_idx["rbsp_off"][-1] + want_idx["rbsp_len"][-1]) if len(want_idx) else 0
    assert np.array_equal(got_arena[:tot], want_arena[:tot]), window


def test_synthetic_stream_many_windows(orc):
    stream, idx, arena = orc.gen_stream(0x77, 60, 0)         # ~600 KB, 8-12 KiB NALs
    for window in (16384, 20000, 65536, 1 << 20):
        check(orc, stream, window)


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_small_nals(orc, seed):
    """short NALs, zero runs, empty NALs and patterns right at window edges"""
    rng = np.random.RandomState(100 + seed)
    n = 30000
    s = ALPHA[rng.randint(0, len(ALPHA), size=n)] if seed % 2 else rng.randint(0, 256, size=n).astype(np.uint8)
    for at in rng.randint(0, n - 8, size=120):
        s[at:at + 3] = (0, 0, 1)
    if seed % 2 == 0:
        # no empty NAL, so that the walk covers the whole stream
        for at in np.nonzero((s[:-6] == 0) & (s[1:-5] == 0) & (s[2:-4] == 1) & (s[3:-3] == 0) & (s[4:-2] == 0))[0]:
            s[at + 3] = 0x42
    for window in (4096, 4112, 8192):
        check(orc, s, window)


def test_nal_longer_than_window_is_reported(orc):
    stream, idx, arena = orc.gen_stream(0x78, 4, 0)          # NALs of 8-12 KiB
    rc, got_idx, got_arena, s = _sim.index_extract_windowed(stream, 4096)
    assert rc == 0 and int(s["error"]) == -4                 # HBS_E_CAPACITY


def test_edges(orc):
    check(orc, np.zeros(0, dtype=np.uint8), 4096)
    check(orc, np.array([0, 0, 1, 0x40, 1, 2, 3], dtype=np.uint8), 4096)
    s = np.full(10000, 0x55, dtype=np.uint8)                  # no start code at all
    check(orc, s, 4096)
    for at in range(700, 10000, 1500):                       # NALs shorter than the window
        s[at:at + 3] = (0, 0, 1)
    s[4094:4097] = (0, 0, 1)                                  # a start code across the first window edge
    check(orc, s, 4096)
    s[8190:8194] = (0, 0, 0, 1)                               # a 4-byte one across the second
    check(orc, s, 4096)
    s[4090:4096] = (0, 0, 1, 0, 0, 1)                         # an empty NAL at a window edge ends the walk
    check(orc, s, 4096)
