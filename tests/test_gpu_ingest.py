"""hbs_index_extract_host: a host stream scanned window by window on the GPU (uploads, scans and
downloads overlapped) must give the whole-stream result of the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch
    import hevcbitstream_amd as hbs
    assert torch.cuda.is_available()
    c = hbs.Context(0)
    yield c
    c.close()


def check(ctx, orc, stream, window, pinned=True):
    want_idx, want_arena, why = orc.index_extract(stream)
    got_idx, got_arena, s = ctx.index_extract_host(stream, window_bytes=window, pinned=pinned)
    assert int(s["error"]) == 0 and int(s["stop_reason"]) == why, (s, window)
    assert len(got_idx) == len(want_idx), (window, len(got_idx), len(want_idx))
    for f in ("start", "end", "rbsp_off", "rbsp_len", "status"):
        assert np.array_equal(got_idx[f], want_idx[f]), (f, window)
    tot = int(want_idx["rbsp_off"][-1] + want_idx["rbsp_len"][-1]) if len(want_idx) else 0
    assert np.array_equal(got_arena[:tot], want_arena[:tot]), window


def test_synthetic_stream_windows(ctx, orc):
    stream, idx, arena = orc.gen_stream(0x91, 1500, 0)       # ~15 MB
    for window in (1 << 16, (1 << 20) + 4096, 4 << 20, 64 << 20):
        check(ctx, orc, stream, window)
    check(ctx, orc, stream, 1 << 20, pinned=False)


def test_zero_heavy_and_patterns(ctx, orc):
    stream, idx, arena = orc.gen_stream(0x92, 400, 1)
    check(ctx, orc, stream, 1 << 18)
    rng = np.random.RandomState(5)
    n = 3 << 20
    s = rng.randint(0, 256, size=n).astype(np.uint8)
    for pat, cnt in ((b"\x00\x00\x01", n // 3000), (b"\x00\x00\x00\x01", n // 9000), (b"\x00\x00\x03", n // 700)):
        for at in rng.randint(0, n - 80, size=cnt):
            s[at:at + len(pat)] = np.frombuffer(pat, dtype=np.uint8)
    # empty NALs would end the walk early: break them up
    for at in np.nonzero((s[:-6] == 0) & (s[1:-5] == 0) & (s[2:-4] == 1) & (s[3:-3] == 0) & (s[4:-2] == 0))[0]:
        s[at + 3] = 0x42
    for window in (1 << 16, 1 << 19):
        check(ctx, orc, s, window)


def test_limits(ctx, orc):
    stream, idx, arena = orc.gen_stream(0x93, 8, 0)          # NALs of 8-12 KiB, window of 4 KiB
    got_idx, got_arena, s = ctx.index_extract_host(stream, window_bytes=4096)
    assert int(s["error"]) == -4
    check(ctx, orc, np.zeros(0, dtype=np.uint8), 1 << 16)
