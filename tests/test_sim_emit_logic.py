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


def _byte_walk(chunks, start_at, gaps, count):
    """rbsp_to_nal's loop (h264_nal.c:110-116) over the bytes, entered with `count`; a NAL that begins puts its start code in
    and sets the count to 0"""
    ins = 0
    for c in range(len(chunks) // 16):
        for b in range(16):
            if start_at[c] == b:
                ins += int(gaps[c])
                count = 0
            v = int(chunks[16 * c + b])
            if count == 2 and v <= 3:
                ins += 1
                count = 0
            count = count + 1 if v == 0 else 0
    return ins, count


def test_dense_tile_chunk_algebra_equals_the_byte_walk():
    """k3_dense_tile (round 4) does not walk bytes: a chunk is classified once for the three counts it may be entered with (what
    goes in, the count behind it, whether it is all zeros -- dz_fast4), a chunk in which a NAL begins as two half chunks
    (dz_one_start4), and the chunks are combined in order.  The same functions stepped on the CPU against the byte walk:
    zero-heavy bytes, runs of zeros of every length and phase, 00 00 03 padding, NAL starts at every byte of a chunk."""
    from tests import _sim
    rng = np.random.RandomState(7)
    alpha = np.array([0, 0, 0, 0, 0, 1, 2, 3, 3, 4, 0x80, 0xFF], dtype=np.uint8)
    cases = []
    for rep in range(60):
        n = int(rng.randint(1, 400))
        kind = rep % 4
        if kind == 0:
            b = alpha[rng.randint(0, len(alpha), size=16 * n)]
        elif kind == 1:                                  # runs of zeros of random lengths between single other bytes
            b = np.zeros(16 * n, dtype=np.uint8)
            at = int(rng.randint(0, 40))
            while at < len(b):
                b[at] = alpha[rng.randint(5, len(alpha))]
                at += int(rng.randint(1, 70))
        elif kind == 2:                                  # padding, at every phase
            b = np.tile(np.array([0, 0, 3], dtype=np.uint8), 16 * n // 3 + 2)[rep % 3: rep % 3 + 16 * n].copy()
        else:
            b = rng.randint(0, 256, size=16 * n).astype(np.uint8)
            b[rng.rand(16 * n) < 0.3] = 0
        start_at = np.full(n, -1, dtype=np.int8)
        gaps = np.zeros(n, dtype=np.uint32)
        for c in rng.randint(0, n, size=max(1, n // 12)):
            start_at[c] = rng.randint(0, 16)
            gaps[c] = rng.randint(3, 6)
        cases.append((b, start_at, gaps))
    # every start offset on the same all-zero and padded neighbourhoods
    for s in range(16):
        for fill in (0, 1):
            b = np.zeros(48, dtype=np.uint8) if fill == 0 else np.tile(np.array([0, 0, 3], dtype=np.uint8), 16)
            start_at = np.array([-1, s, -1], dtype=np.int8)
            cases.append((b, start_at, np.array([0, 4, 0], dtype=np.uint32)))
    for b, start_at, gaps in cases:
        tot, st = _sim.dz_walk(b, start_at, gaps)
        for h in range(3):
            want_ins, want_count = _byte_walk(b, start_at, gaps, h)
            assert tot[h] == want_ins, (h, len(b), tot, want_ins)
            assert st[h] == want_count, (h, len(b), st, want_count)
