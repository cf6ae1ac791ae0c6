"""CPU single-step of the product's header readers (hbs_parse.h, the code K4
runs per lane) against the oracle parser, on streams from tests/hevc_synth."""
import json
import os

import numpy as np
import pytest

from tests import _sim
from tests._parsecmp import compare, oracle_pass
from tests.hevc_synth import Synth, annexb, stream_4k30

HERE = os.path.dirname(os.path.abspath(__file__))


def run(nals):
    stream = np.frombuffer(annexb(nals) , dtype=np.uint8)
    idx, arena, s = _sim.index_extract(stream)
    assert len(idx) == len(nals)
    parsed, structs = _sim.parse_headers(arena, idx)
    compare(parsed, structs, arena, idx, oracle_pass(nals))


def sequence(seed, rich=True):
    g = Synth(seed, rich=rich)
    rng = np.random.RandomState(seed)
    seq = [g.vps(), g.sps_nal(int(rng.randint(64, 4096)), int(rng.randint(64, 2304))), g.pps_nal()]
    for k in range(10):
        t = int(rng.choice([0, 1, 8, 9, 16, 19, 20, 21]))
        seq.append(g.slice_nal(t, first=bool(rng.randint(0, 2)),
                               payload=rng.randint(0, 256, size=rng.randint(1, 80)).astype(np.uint8).tobytes(),
                               address=int(rng.randint(0, 100))))
        if k == 3 and rng.rand() < 0.5:
            seq.append(g.pps_nal())
        if k == 6 and rng.rand() < 0.5:
            seq.append(g.sps_nal(int(rng.randint(64, 4096)), int(rng.randint(64, 2304))))
            seq.append(g.pps_nal())
    seq.append(bytes([35 << 1, 1, 0x50]))              # AUD: not dispatched by the reference (rc -1)
    seq.append(bytes([39 << 1, 1, 1, 2, 3, 0x80]))     # SEI: idem
    return seq


def test_rich_sequences():
    for seed in range(150):
        run(sequence(seed))


def test_slices_before_any_parameter_set():
    g = Synth(5, rich=True)
    g.sps_nal(640, 480)
    g.pps_nal()
    sl = [g.slice_nal(1, first=True, payload=b"\x11\x22\x33"), g.slice_nal(19, first=False, payload=b"\x44" * 9, address=3)]
    # the stream itself carries no parameter sets: the reference parses against its zeroed structs
    run(sl)


def broken(seq, rng, which):
    """cut NALs short (zeros past the end, overrun) or plant 00 00 02 (nal_to_rbsp rejects)"""
    out = []
    for nal in seq:
        t = (nal[0] >> 1) & 0x3F
        r = rng.rand()
        if which(t) and r < 0.3 and len(nal) > 4:
            nal = nal[: rng.randint(2, len(nal))]
            if nal[-1] == 0:
                nal = nal + b"\x80"
        elif which(t) and r < 0.4:
            nal = nal[:3] + b"\x00\x00\x02" + nal[3:]
        out.append(nal)
    return out


def test_truncated_and_broken_slices():
    """Slices and VPS damaged, SPS/PPS intact: every NAL must still agree with the reference
    semantics (a slice parsed against damaged parameter sets is outside the envelope where the
    reference is deterministic: it reads RPS table rows left behind by earlier slices)."""
    for seed in range(60):
        run(broken(sequence(seed), np.random.RandomState(1000 + seed), lambda t: t not in (33, 34)))


def test_truncated_parameter_sets():
    for seed in range(60):
        g = Synth(seed, rich=True)
        rng = np.random.RandomState(2000 + seed)
        seq = [g.vps(), g.sps_nal(int(rng.randint(64, 4096)), int(rng.randint(64, 2304))), g.pps_nal(), g.vps()]
        run(broken(seq, rng, lambda t: True))


def test_ten_nal_parameter_sets_and_idr():
    data = open(os.path.join(HERE, "golden", "ten_nal.hevc"), "rb").read()
    idx = json.load(open(os.path.join(HERE, "golden", "ten_nal.index.json")))
    run([data[s:e] for s, e, _ in idx[:4]])


def test_config3_downsized():
    stream, n = stream_4k30(3, n_pictures=10, slices_per_picture=8, idr_every=4, payload_bytes=(200, 400))
    s = np.frombuffer(stream, dtype=np.uint8)
    idx, arena, summ = _sim.index_extract(s)
    assert len(idx) == n
    nals = [bytes(s[int(a):int(b)]) for a, b in zip(idx["start"], idx["end"])]
    parsed, structs = _sim.parse_headers(arena, idx)
    compare(parsed, structs, arena, idx, oracle_pass(nals))
    assert (parsed["rc"] >= 0).all()


def test_bit_io_fast_paths_equal_the_bit_loops():
    """hbs_bitfast.h (whole fields at once) against bs.h's one bit at a time: value, cursor and written bytes agree for
    every width and every cursor position, the end of the buffer and beyond included"""
    import ctypes as C
    L = _sim.lib()
    L.sim_bitio_check.argtypes = [C.c_uint64, C.c_int64]
    L.sim_bitio_check.restype = C.c_int64
    assert L.sim_bitio_check(12345, 400000) == 0


def test_exact_rewalk_of_slices_that_depend_on_earlier_nals():
    """hbs_parse_fix.h single-stepped: on streams the spec forbids -- parameter sets and slices damaged, sequences glued
    together, an IDR coded as a P slice every hundred slices -- the batch parse plus the re-walk of the slices whose RPS
    rows have another last writer than their SPS equals the oracle's sequential parse everywhere; the batch parse alone
    does not (counted), and no chain of writers is deeper than the re-walk follows."""
    def one(nals, fix, stats):
        stream = np.frombuffer(annexb(nals), dtype=np.uint8)
        idx, arena, s = _sim.index_extract(stream)
        assert len(idx) == len(nals)
        parsed, structs = _sim.parse_headers(arena, idx, fix=fix, stats=stats)
        return parsed, structs, arena, idx

    rewalked = wrong_without = 0
    streams = []
    for seed in range(6000, 6120):
        streams.append(broken(sequence(seed), np.random.RandomState(7 * seed + 2), lambda t: True))
    for seed in (300, 1036, 1320, 5224):
        nals = []
        for s2 in range(seed, seed + 8):
            nals += broken(sequence(s2), np.random.RandomState(7 * s2 + 2), lambda t: True) if s2 % 2 else sequence(s2)
        streams.append(nals)
    for nals in streams:
        exp = oracle_pass(nals)
        st = []
        compare(*one(nals, 1, st), exp)
        assert st[2] == 0
        rewalked += st[1]
        if st[1]:
            try:
                compare(*one(nals, 0, []), exp)
            except AssertionError:
                wrong_without += 1
    assert rewalked > 10 and wrong_without > 3, (rewalked, wrong_without)
    # the 4K30-style batch with one out-of-spec slice in a hundred
    stream, n = stream_4k30(21, n_pictures=400, slices_per_picture=8, idr_every=60, payload_bytes=(60, 120), forbidden_every=100)
    s = np.frombuffer(stream, dtype=np.uint8)
    idx, arena, summ = _sim.index_extract(s)
    nals = [bytes(s[int(a):int(b)]) for a, b in zip(idx["start"], idx["end"])]
    exp = oracle_pass(nals)
    st = []
    parsed, structs = _sim.parse_headers(arena, idx, fix=1, stats=st)
    compare(parsed, structs, arena, idx, exp)
    assert st[0] == 1 and 20 <= st[1] <= 40 and st[2] == 0, st


def test_table_state_behind_a_batch_equals_the_sequential_parsers():
    """k4_state's arithmetic single-stepped: the 32 rows of the derived tables behind the last NAL of a batch, each taken from
    the last NAL that wrote it (hbs_parse_fix.h), against the tables the oracle's sequential parser is left with -- on
    ordinary sequences, damaged ones, and sequences glued together (rows of earlier SPSs with more sets stay alive)."""
    from tests import _orc
    checked = 0
    for seed in range(7000, 7080):
        variants = [sequence(seed), broken(sequence(seed), np.random.RandomState(7 * seed + 2), lambda t: True)]
        if seed % 4 == 0:
            variants.append(sequence(seed) + broken(sequence(seed + 1), np.random.RandomState(seed), lambda t: True) + sequence(seed + 2))
        for nals in variants:
            stream = np.frombuffer(annexb(nals), dtype=np.uint8)
            idx, arena, s = _sim.index_extract(stream)
            got, ok = _sim.parse_state(arena, idx)
            if not ok:
                continue
            o = _orc.OracleHevc()
            for nal in nals:
                o.read(nal)
            want = o.tables()
            o.close()
            # counts exactly; values below the counts (what lies beyond them in a row is never read -- hevc_stream.c:35-59, :1043-1075
            # loop to the counts -- and keeps whatever an earlier, longer set left there in the reference)
            g3, w3 = got[:96].reshape(3, 32), want[:96].reshape(3, 32)
            assert np.array_equal(g3, w3), (seed, g3, w3)
            g4, w4 = got[96:].reshape(4, 32, 32), want[96:].reshape(4, 32, 32)
            for r in range(32):
                nn, npos = int(w3[1][r]), int(w3[2][r])
                for tbl, cnt in ((0, nn), (1, nn), (2, npos), (3, npos)):
                    c = max(0, min(cnt, 32))
                    assert np.array_equal(g4[tbl][r][:c], w4[tbl][r][:c]), (seed, r, tbl)
            checked += 1
    assert checked > 150
