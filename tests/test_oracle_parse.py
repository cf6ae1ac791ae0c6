"""Pin the oracle's HEVC header parser (oracle/hbs_oracle_parse.c): against the
golden struct dumps generated from the real reference (parse_vectors.json) and,
when oracle/_ref is present, against the reference itself on fresh sequences."""
import hashlib
import json
import os

import numpy as np
import pytest

from tests import _orc
from tests.hevc_synth import Synth

HERE = os.path.dirname(os.path.abspath(__file__))


def test_abi_layout_matches_reference_header():
    """include/hevc_stream.h vs sizeof/offsetof taken from the reference header."""
    want = json.load(open(os.path.join(HERE, "golden", "abi_layout.json")))
    lay = _orc.layout()
    for k, v in want.items():
        if lay.get(k):
            assert lay[k]["size"] == v, k
    for k, v in want.items():
        if "." in k:
            t, f = k.split(".")
            if lay.get(t):
                offs = {x[0]: x[1] for x in lay[t]["fields"] if x}
                assert offs[f] == v, k


def test_parse_golden():
    gold = json.load(open(os.path.join(HERE, "golden", "parse_vectors.json")))
    for seq in gold:
        o = _orc.OracleHevc()
        for step in seq["steps"]:
            nal = bytes.fromhex(step["nal"])
            rc = o.read(nal)
            assert rc == step["rc"], (seq["seed"], step["nal"][:8])
            snap = o.snapshot()
            for k, pairs in step["structs"].items():
                want = np.zeros_like(snap[k])
                for i, v in pairs:
                    want[i] = v
                assert np.array_equal(snap[k], want), (seq["seed"], k, step["nal"][:8])
            if "slice_data" in step:
                size, data = o.slice_data()
                assert size == step["slice_data"][0]
                if data is not None:
                    assert hashlib.md5(data).hexdigest() == step["slice_data"][1]
        o.close()


def test_ten_nal_known_answers():
    """SURVEY.md App. E known answers for the x265-style parameter sets + IDR slice."""
    data = open(os.path.join(HERE, "golden", "ten_nal.hevc"), "rb").read()
    idx = json.load(open(os.path.join(HERE, "golden", "ten_nal.index.json")))
    o = _orc.OracleHevc()
    rcs = [o.read(data[s:e]) for s, e, _ in idx[:4]]
    assert rcs == [24, 42, 7, 206]
    f = {n: (i, c) for n, i, c in _orc.flat_fields("hevc_sps_t")}
    sps = o.v["sps"]
    assert sps[f["pic_width_in_luma_samples"][0]] == 1920 and sps[f["pic_height_in_luma_samples"][0]] == 1080
    assert sps[f["vui.vui_time_scale"][0]] == 30 and sps[f["chroma_format_idc"][0]] == 1
    fs = {n: (i, c) for n, i, c in _orc.flat_fields("hevc_slice_header_t")}
    sh = o.v["sh"]
    assert sh[fs["slice_type"][0]] == 2 and sh[fs["slice_qp_delta"][0]] == 8 and sh[fs["num_entry_point_offsets"][0]] == 9
    size, sd = o.slice_data()
    assert size == 197 and sd[0] == 0xC4


@pytest.mark.ref
def test_parse_fuzz_vs_reference(ref):
    o, r = _orc.OracleHevc(), _orc.ReferenceHevc()
    for seed in range(120):
        g = Synth(seed, rich=True)
        rng = np.random.RandomState(seed)
        seq = [g.vps(), g.sps_nal(int(rng.randint(64, 4096)), int(rng.randint(64, 2304))), g.pps_nal()]
        for k in range(6):
            t = int(rng.choice([0, 1, 8, 9, 16, 19, 20, 21]))
            seq.append(g.slice_nal(t, first=bool(rng.randint(0, 2)),
                                   payload=rng.randint(0, 256, size=rng.randint(1, 60)).astype(np.uint8).tobytes(),
                                   address=int(rng.randint(0, 100))))
        seq.append(bytes([40 << 1, 1, 5, 0x80]))
        for nal in seq:
            a, b = o.read(nal), r.read(nal)
            assert a == b
            so, sr = o.snapshot(), r.snapshot()
            for k in so:
                assert np.array_equal(so[k], sr[k]), (seed, k)
            if b >= 0 and ((nal[0] >> 1) <= 9 or 16 <= (nal[0] >> 1) <= 21):
                assert o.slice_data() == r.slice_data()
