"""hbs_index_parse (BASELINE config 3 without an RBSP arena: index-only scan, then the header parse on windows stripped
straight from the stream) against hbs_index_extract + hbs_parse_headers -- which the other tests pin on the oracle -- record
for record and byte for byte, and against the oracle directly on a sequence of a few thousand NALs."""
import numpy as np
import pytest

from tests._parsecmp import compare, oracle_pass
from tests.hevc_synth import Synth, annexb, stream_4k30
from tests.test_sim_parse_logic import broken, sequence

pytestmark = pytest.mark.gpu
INT_MIN = -(1 << 31)


@pytest.fixture(scope="module")
def ctx():
    import hevcbitstream_amd as hbs
    c = hbs.Context(0)
    yield c
    c.close()


def both_ways(ctx, stream_bytes, window=0):
    """(index, parsed, structs, arena) by the arena path; (index, parsed, structs, payload_off, parse summary) by hbs_index_parse"""
    import torch
    import hevcbitstream_amd as hbs
    from hevcbitstream_amd.api import PARSED, SUMMARY
    s = np.frombuffer(stream_bytes, dtype=np.uint8).copy()
    d = torch.from_numpy(s).cuda()
    index, rbsp, summary, cap = ctx.alloc_outputs(d.numel())
    ctx.index_extract_async(d, index, cap, rbsp, summary)
    sm = ctx.read_summary(summary)
    n = int(sm["nal_count"])
    parsed_a, structs_a = ctx.parse_headers(rbsp, index, n, poison=0xA5)
    idx_a = index[: n * 32].cpu().numpy().view(hbs.NAL_ENTRY).copy()
    arena = rbsp[: int(sm["rbsp_bytes"])].cpu().numpy()

    index2 = torch.zeros_like(index)
    parsed2 = torch.empty(max(n, 1) * PARSED.itemsize, dtype=torch.uint8, device="cuda")
    structs2 = torch.full_like(structs_a, 0xA5)          # the same poison: what neither path writes is equal too
    pay = torch.zeros(max(n, 1), dtype=torch.int64, device="cuda")
    s1 = torch.zeros(SUMMARY.itemsize, dtype=torch.uint8, device="cuda")
    s2 = torch.zeros(SUMMARY.itemsize, dtype=torch.uint8, device="cuda")
    n2 = ctx.index_parse_async(d, index2, cap, parsed2, structs2, s1, s2, window=window, payload_off=pay)
    ps = ctx.read_summary(s2)
    assert n2 == n
    idx_b = index2[: n * 32].cpu().numpy().view(hbs.NAL_ENTRY).copy()
    parsed_b = parsed2[: n * PARSED.itemsize].cpu().numpy().view(PARSED).copy()
    return (s, idx_a, parsed_a, structs_a.cpu().numpy(), arena), (idx_b, parsed_b, structs2.cpu().numpy(), pay.cpu().numpy()[:n], ps)


def same(a, b):
    s, idx_a, parsed_a, structs_a, arena = a
    idx_b, parsed_b, structs_b, pay, ps = b
    assert int(ps["error"]) == 0, ps
    for f in ("start", "end", "rbsp_off", "rbsp_len", "status"):
        assert np.array_equal(idx_a[f], idx_b[f]), f
    for f in parsed_a.dtype.names:
        assert np.array_equal(parsed_a[f], parsed_b[f]), f
    assert np.array_equal(structs_a, structs_b), "struct arenas differ"
    # the payload's place in the STREAM: the byte there is the RBSP byte at slice_data_off, for every parsed slice with a payload
    sl = (parsed_a["nal_unit_type"] >= 0) & (parsed_a["nal_unit_type"] < 32) & (parsed_a["slice_data_off"] > 0) & (parsed_a["struct_off"] != np.uint64(0xFFFFFFFFFFFFFFFF))
    for k in np.nonzero(sl)[0]:
        off, ln = int(parsed_a["slice_data_off"][k]), int(idx_a["rbsp_len"][k])
        p = int(pay[k])
        if off < ln:
            assert int(idx_a["start"][k]) <= p < int(idx_a["end"][k]), k
            assert s[p] == arena[int(idx_a["rbsp_off"][k]) + off], k
            # and nothing but kept bytes in front of it: the RBSP of the bytes [start, p) is the first `off` bytes of the NAL's RBSP
        else:
            assert p == int(idx_a["end"][k]) or p <= int(idx_a["end"][k]), k
    assert np.all(pay[~sl] == -1)


def test_sequences_and_broken_streams(ctx):
    for seed in range(24):
        same(*both_ways(ctx, annexb(sequence(seed))))
    for seed in range(16):
        nals = broken(sequence(seed), np.random.RandomState(1000 + seed), lambda t: True)
        same(*both_ways(ctx, annexb(nals)))


def test_slices_longer_than_their_window(ctx, orc):
    """4K30 sequences whose slice payloads (KiBs, with emulation prevention bytes in them) lie far outside the 512-byte windows;
    3000 NALs of one also against the oracle's parser, field for field"""
    stream, count = stream_4k30(5, n_pictures=375, slices_per_picture=8, idr_every=30, payload_bytes=(3000, 9000), rich=True)
    a, b = both_ways(ctx, stream)
    same(a, b)
    assert len(a[1]) == count >= 3000
    nals = [bytes(a[0][int(x):int(y)]) for x, y in zip(a[1]["start"][:3000], a[1]["end"][:3000])]
    compare(b[1][:3000], b[2], a[4], a[1][:3000], oracle_pass(nals))     # hbs_index_parse's records and structs against the oracle
    same(a, both_ways(ctx, stream, window=4096)[1])


def test_a_window_that_is_too_small_is_reported(ctx):
    """64-byte windows do not hold every slice header of a rich sequence (entry points, weight tables): HBS_E_CAPACITY in the
    parse summary, rc = INT32_MIN for exactly the NALs whose header does not end 8 bytes inside its window, every other record as
    from the arena path -- and a larger window gives the whole answer"""
    hits = 0
    for seed in (5, 6, 7):
        stream, _ = stream_4k30(seed, n_pictures=40, slices_per_picture=6, idr_every=10, payload_bytes=(300, 900), rich=True)
        a, b = both_ways(ctx, stream, window=64)
        idx_b, parsed_b, structs_b, pay, ps = b
        sl = (a[2]["nal_unit_type"] >= 0) & (a[2]["nal_unit_type"] < 32) & (a[2]["struct_off"] != np.uint64(0xFFFFFFFFFFFFFFFF))
        too_long = sl & (a[1]["rbsp_len"] > 64) & (a[2]["slice_data_off"] + 8 > 64)
        # (a header cut short inside the window can also look long to the cut parse: such a NAL is reported as well, never guessed)
        reported = parsed_b["rc"] == INT_MIN
        assert np.all(reported[too_long]), seed
        assert int(ps["error"]) == (-4 if reported.any() else 0)
        ok = ~reported
        for f in a[2].dtype.names:
            assert np.array_equal(a[2][f][ok], parsed_b[f][ok]), (seed, f)
        hits += int(too_long.sum())
        same(a, both_ways(ctx, stream, window=1024)[1])
    assert hits > 0, "no header of the test streams was longer than 56 bytes"


def test_streams_without_nals_and_small_index(ctx):
    """no start code at all, an empty stream, and an index that is too small (the scan's HBS_E_CAPACITY comes back as the call's error)"""
    import torch
    import hevcbitstream_amd as hbs
    from hevcbitstream_amd.api import PARSED, SUMMARY
    for data in (np.full(5000, 0x55, dtype=np.uint8), np.zeros(0, dtype=np.uint8)):
        d = torch.from_numpy(data).cuda()
        index = torch.zeros(64 * 32, dtype=torch.uint8, device="cuda")
        parsed = torch.zeros(64 * PARSED.itemsize, dtype=torch.uint8, device="cuda")
        s1 = torch.zeros(SUMMARY.itemsize, dtype=torch.uint8, device="cuda")
        s2 = torch.zeros(SUMMARY.itemsize, dtype=torch.uint8, device="cuda")
        assert ctx.index_parse_async(d, index, 64, parsed, None, s1, s2) == 0
        assert int(ctx.read_summary(s1)["nal_count"]) == 0 and int(ctx.read_summary(s2)["error"]) == 0
    stream = np.frombuffer(annexb(sequence(2)), dtype=np.uint8).copy()
    d = torch.from_numpy(stream).cuda()
    index = torch.zeros(2 * 32, dtype=torch.uint8, device="cuda")
    parsed = torch.zeros(64 * PARSED.itemsize, dtype=torch.uint8, device="cuda")
    s1 = torch.zeros(SUMMARY.itemsize, dtype=torch.uint8, device="cuda")
    s2 = torch.zeros(SUMMARY.itemsize, dtype=torch.uint8, device="cuda")
    with pytest.raises(hbs.HbsError):
        ctx.index_parse_async(d, index, 2, parsed, None, s1, s2)


def test_workspace_is_sized_by_the_nals_found_not_by_the_index_capacity():
    """round 3's advice: with the default index capacity (stream_bytes / 3 entries on a small stream) the header windows
    used to be sized by it -- 11 GB for a 64 MiB stream.  They are sized by the NALs the scan found."""
    import torch
    import hevcbitstream_amd as hbs
    from hevcbitstream_amd.api import PARSED, SUMMARY
    c = hbs.Context(0)
    try:
        stream, _ = stream_4k30(5, n_pictures=400, slices_per_picture=4, idr_every=30, payload_bytes=(20000, 40000))
        s = np.frombuffer(stream, dtype=np.uint8).copy()
        d = torch.from_numpy(s).cuda()
        cap = c.default_index_cap(d.numel())
        assert cap > 1_000_000                                    # the capacity is far above the ~1 700 NALs there are
        index = torch.zeros(cap * 32, dtype=torch.uint8, device="cuda")
        n_guess = 4000
        parsed = torch.empty(n_guess * PARSED.itemsize, dtype=torch.uint8, device="cuda")
        structs = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
        s1 = torch.zeros(SUMMARY.itemsize, dtype=torch.uint8, device="cuda")
        s2 = torch.zeros(SUMMARY.itemsize, dtype=torch.uint8, device="cuda")
        n = c.index_parse_async(d, index, cap, parsed, structs, s1, s2)
        assert 1000 < n < n_guess and int(c.read_summary(s2)["error"]) == 0
        held = c.device_bytes()
        assert held < 64 << 20, "context scratch %d bytes for a %d-byte stream of %d NALs" % (held, d.numel(), n)
        # and the answer is the arena path's
        a, b = both_ways(c, stream)
        same(a, b)
    finally:
        c.close()
