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
    ctx.set_ingest_window_max(4096)                          # no growth allowed: round 4's behaviour
    got_idx, got_arena, s = ctx.index_extract_host(stream, window_bytes=4096)
    assert int(s["error"]) == -4 and int(s["reserved"][2]) == 1
    # ... everything in front of the NAL that did not fit was delivered, and the summary says where that NAL is
    k = int(s["nal_count"])
    assert np.array_equal(got_idx[:k]["start"], idx[:k]["start"]) and np.array_equal(got_idx[:k]["end"], idx[:k]["end"])
    assert int(s["reserved"][1]) == (int(idx["end"][k - 1]) if k else 0)
    ctx.set_ingest_window_max(0)
    check(ctx, orc, np.zeros(0, dtype=np.uint8), 1 << 16)


def test_a_nal_longer_than_the_window_grows_the_window(ctx, orc):
    """Round 5 (the reference's fixed 32 MiB reader cuts such a NAL short, hevc_analyze.c:126,190-209): NALs of 8-12 KiB through a
    4 KiB window -- the window doubles until they fit (16 KiB) and the result is the whole stream's; then a stream with one
    3 MiB NAL in the middle of small ones through 64 KiB windows; then the ceiling in the way."""
    stream, idx, arena = orc.gen_stream(0x93, 40, 0)
    got_idx, got_arena, s = ctx.index_extract_host(stream, window_bytes=4096)
    assert int(s["error"]) == 0 and int(s["nal_count"]) == len(idx) and 8192 <= int(s["reserved"][0]) <= 32768
    for f in ("start", "end", "rbsp_off", "rbsp_len", "status"):
        assert np.array_equal(got_idx[f], idx[f]), f
    assert np.array_equal(got_arena, arena)
    rng = np.random.RandomState(5)
    small, sidx, _ = orc.gen_stream(0x94, 60, 0)
    big = np.concatenate([np.array([0, 0, 1, 0x26, 0x01], dtype=np.uint8), rng.randint(4, 256, size=3 << 20).astype(np.uint8)])
    cut = int(sidx["end"][29])
    mixed = np.concatenate([small[:cut], big, small[cut:]])
    got_idx, got_arena, s = ctx.index_extract_host(mixed, window_bytes=65536)
    import torch
    ref_idx, ref_arena, _ = ctx.index_extract(torch.from_numpy(mixed).cuda())
    assert int(s["error"]) == 0 and int(s["reserved"][0]) == 2 << 20          # (a window buffer holds what is scanned again + a window: a NAL fits when 2 windows hold it)
    for f in ("start", "end", "rbsp_off", "rbsp_len", "status"):
        assert np.array_equal(got_idx[f], ref_idx[f]), f
    assert np.array_equal(got_arena, ref_arena)
    assert len(got_idx) == len(sidx) + 1
    ctx.set_ingest_window_max(1 << 20)
    got_idx, got_arena, s = ctx.index_extract_host(mixed, window_bytes=65536)
    assert int(s["error"]) == -4 and int(s["reserved"][2]) == 1 and int(s["nal_count"]) == 30 and int(s["reserved"][1]) == cut
    ctx.set_ingest_window_max(0)
