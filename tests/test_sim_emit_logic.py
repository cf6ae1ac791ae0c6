"""CPU single-step of K3's per-segment logic (hbs_emit.h) and of the synthetic
generator against the oracle."""
import numpy as np
import pytest

from tests import _sim
from tests._orc import NAL_ENTRY

ALPHA = np.array([0, 0, 0, 0, 1, 1, 2, 3, 3, 4, 0x40, 0x80, 0xFF], dtype=np.uint8)


def fake_index(lens, gaps):
    idx = np.zeros(len(lens), dtype=NAL_ENTRY)
    off = pos = 0
    for k, (n, g) in enumerate(zip(lens, gaps)):
        idx["start"][k] = pos + g
        idx["end"][k] = pos + g + n          # lengths here are placeholders: only start-prev_end is used
        idx["rbsp_off"][k] = off
        idx["rbsp_len"][k] = n
        pos += g + n
        off += n
    return idx


def test_emit_random_rbsp(orc):
    rng = np.random.RandomState(21)
    for _ in range(400):
        nn = rng.randint(1, 6)
        lens = [int(rng.randint(1, 900)) for _ in range(nn)]
        gaps = [int(rng.randint(3, 7)) for _ in range(nn)]
        arena = ALPHA[rng.randint(0, len(ALPHA), size=sum(lens))].copy()
        if rng.rand() < 0.5:
            arena[rng.rand(len(arena)) < 0.6] = 0x55
        idx = fake_index(lens, gaps)
        want = orc.emit_annexb(arena, idx)
        got, idx_out = _sim.emit_annexb(arena, idx)
        assert np.array_equal(got, want), bytes(arena[:64]).hex()


def test_emit_long_zero_runs(orc):
    for z in (2, 3, 4, 5, 254, 255, 256, 257, 258, 511, 512, 513, 700):
        for tail in ([], [1], [3], [4], [0, 0, 1]):
            for lead in (0, 1, 255, 256, 257):
                arena = np.array([7] * lead + [0] * z + tail, dtype=np.uint8)
                idx = fake_index([len(arena)], [3])
                assert np.array_equal(_sim.emit_annexb(arena, idx)[0], orc.emit_annexb(arena, idx)), (z, tail, lead)


@pytest.mark.parametrize("mode", [0, 1])
def test_synth_matches_oracle_generator(orc, mode):
    stream, idx, arena = orc.gen_stream(0x1234, 30, mode)
    a2, i2 = _sim.synth_rbsp(0x1234, 30, mode)
    assert np.array_equal(a2, arena)
    assert np.array_equal(i2["rbsp_off"], idx["rbsp_off"]) and np.array_equal(i2["rbsp_len"], idx["rbsp_len"])
    s2, i3 = _sim.emit_annexb(a2, i2, gap_mode=1)
    assert np.array_equal(s2, stream)
    assert np.array_equal(i3["start"], idx["start"]) and np.array_equal(i3["end"], idx["end"])


def test_roundtrip_through_both(orc):
    """extract (K12 logic) then emit (K3 logic) reproduces the stream."""
    for mode in (0, 1):
        stream, idx, arena = orc.gen_stream(77, 25, mode)
        got_idx, got_arena, s = _sim.index_extract(stream)
        back, _ = _sim.emit_annexb(got_arena, got_idx)
        assert np.array_equal(back, stream)
