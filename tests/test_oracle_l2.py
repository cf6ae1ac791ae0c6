"""Pin the oracle's byte layer: against golden vectors generated from the real
reference (tests/golden/l2_vectors.json, made by make_golden_l2.py) and, when
oracle/_ref is present (dev container), against the reference itself on fuzz."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "l2_vectors.json")))


def test_find_nal_unit_golden(orc):
    for hx, want in GOLD["find"]:
        assert list(orc.find_nal_unit(bytes.fromhex(hx))) == want, hx


def test_nal_to_rbsp_golden(orc):
    for hx, (r, ns, rs, data) in GOLD["n2r"]:
        got = orc.nal_to_rbsp(bytes.fromhex(hx))
        assert got[0] == r, hx
        if r >= 0:  # sizes are untouched (== input) on error in both
            assert got[1] == ns and got[2] == rs and got[3].hex() == data, hx
        else:
            assert got[1] == ns and got[2] == rs, hx


def test_rbsp_to_nal_golden(orc):
    for hx, (r, data) in GOLD["r2n"]:
        got = orc.rbsp_to_nal(bytes.fromhex(hx))
        assert got[0] == r and got[2].hex() == data, hx


def test_ten_nal_index_golden(orc):
    stream = np.fromfile(os.path.join(HERE, "golden", "ten_nal.hevc"), dtype=np.uint8)
    want = json.load(open(os.path.join(HERE, "golden", "ten_nal.index.json")))
    idx, why = orc.index_stream(stream)
    assert why == -1
    assert [[int(a), int(b)] for a, b in zip(idx["start"], idx["end"])] == [w[:2] for w in want]
    assert idx["status"][-1] & 4 and not idx["status"][:-1].any()


ALPHA = np.array([0, 0, 0, 0, 1, 1, 2, 3, 3, 4, 0x40, 0x80, 0xFF], dtype=np.uint8)


@pytest.mark.ref
def test_fuzz_vs_reference(orc, ref):
    rng = np.random.RandomState(7)
    for _ in range(4000):
        b = bytes(ALPHA[rng.randint(0, len(ALPHA), size=rng.randint(3, 64))])
        assert orc.find_nal_unit(b) == ref.find_nal_unit(b), b.hex()
        a, r = orc.nal_to_rbsp(b), ref.nal_to_rbsp(b)
        assert a == r, b.hex()
        assert orc.rbsp_to_nal(b) == ref.rbsp_to_nal(b), b.hex()


@pytest.mark.ref
def test_stream_loop_vs_reference(orc, ref):
    """orc_index_stream == the hevc_analyze.c:135-205 loop driven on the reference."""
    rng = np.random.RandomState(11)
    for _ in range(300):
        n = rng.randint(4, 400)
        s = ALPHA[rng.randint(0, len(ALPHA), size=n)].copy()
        s[rng.rand(n) < 0.5] = 0x55          # thin out so NALs have some length
        data = bytes(s)
        want, p = [], 0
        while True:
            r, st, en = ref.find_nal_unit(data[p:])
            if r <= 0:
                if r == -1:
                    want.append((p + st, p + en))
                break
            want.append((p + st, p + en))
            p += en
        idx, why = orc.index_stream(s)
        assert [(int(a), int(b)) for a, b in zip(idx["start"], idx["end"])] == want, data.hex()


@pytest.mark.ref
def test_reference_baseline_driver(orc, ref):
    """oracle/ref_driver.c (what bench.py times as the "reference" CPU baseline) walks a stream like the oracle does:
    same NAL starts, same RBSP bytes"""
    import ctypes as C
    import os
    drv = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libref_driver.so")
    if not os.path.exists(drv):
        pytest.skip("oracle/_ref/libref_driver.so not built")
    lib = C.CDLL(drv)
    u8p = C.POINTER(C.c_uint8)
    lib.ref_walk.restype = C.c_int64
    lib.ref_walk.argtypes = [u8p, C.c_int64, u8p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_uint64), C.c_int64]
    for mode in (0, 1):
        stream, idx, arena = orc.gen_stream(0x77 + mode, 150, mode)
        buf = stream.copy()
        out = np.zeros(len(buf) + 64, dtype=np.uint8)
        starts = np.zeros(len(idx) + 8, dtype=np.uint64)
        tot = C.c_int64(0)
        n = lib.ref_walk(buf.ctypes.data_as(u8p), len(buf), out.ctypes.data_as(u8p), len(out), C.byref(tot),
                         starts.ctypes.data_as(C.POINTER(C.c_uint64)), len(starts))
        assert n == len(idx) and np.array_equal(starts[:n], idx["start"])
        assert tot.value == len(arena) and np.array_equal(out[:tot.value], arena)


def test_roundtrip_property(orc):
    """rbsp_to_nal(nal_to_rbsp(x)) == x for accepted x not ending in 00 00 03
    (SURVEY.md App. B); and the synthetic generator's arena/index are what the
    oracle extracts from its stream."""
    for mode in (0, 1):
        s, idx, arena = orc.gen_stream(0x1234, 40, mode)
        i2, a2, why = orc.index_extract(s)
        assert why == -1
        assert np.array_equal(i2, idx) and np.array_equal(a2, arena)
        assert np.array_equal(orc.emit_annexb(a2, i2), s)
