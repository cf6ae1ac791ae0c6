"""The syntax writers (write_hevc_nal_unit, hevc_stream.c:1249-1327) single-stepped on the CPU against
golden outputs of the reference's writer on parsed and on edited structs (tests/golden/make_golden_write.py)."""
import gzip
import json
import os

import numpy as np

from tests import _orc, _sim
from tests._parsecmp import which_struct
from tests.hevc_synth import annexb

HERE = os.path.dirname(os.path.abspath(__file__))
SPS_SLOT = None


def slot_bytes(kind):
    size = _orc.layout()[_orc.STRUCT_TYPES[kind]]["size"]
    if kind == "sps":
        return ((size + 15) // 16) * 16 + 4 * (3 * 32 + 4 * 32 * 32)      # struct + derived RPS tables
    return ((size + 15) // 16) * 16


def field_index(kind, name):
    for n, i, c in _orc.flat_fields(_orc.STRUCT_TYPES[kind]):
        if n == name:
            return i
    raise KeyError(name)


def run_sequence(orc, steps):
    nals = [bytes.fromhex(s["nal"]) for s in steps]
    stream = np.frombuffer(annexb(nals), dtype=np.uint8)
    idx, arena, _ = _sim.index_extract(stream)
    assert len(idx) == len(nals)
    parsed, structs = _sim.parse_headers(arena, idx)
    last = {"sps": None, "pps": None}
    for k, st in enumerate(steps):
        t = int(parsed["nal_unit_type"][k])
        kind = which_struct(t)
        assert int(parsed["rc"][k]) == st["read_rc"], k
        if "write_rc" in st:
            off = int(parsed["struct_off"][k])
            slot = structs[off:off + slot_bytes(kind)].copy()
            view = slot.view(np.int32)
            for name, value in st["edits"]:
                view[field_index(kind, name)] = value
            cap = st["size"] * 3 // 4
            res, rbsp = _sim.write_nal(t, int(parsed["nal_layer_id"][k]), int(parsed["nal_temporal_id_plus1"][k]), slot,
                                       last["sps"], last["pps"], cap)
            if st["write_rc"] < 0:
                assert int(res["rc"]) < 0, k
            else:
                assert int(res["rc"]) == 0, (k, kind)
                rc, _, out = orc.rbsp_to_nal(bytes(rbsp))
                assert rc == st["write_rc"] and out.hex() == st["out"], (k, kind, st["edits"], out.hex()[:80], st["out"][:80])
                if kind == "sh":
                    assert int(res["slice_data_size"]) == st["slice_data_size"], k
        # parameter sets in force for later NALs: the stream's own (the golden script undoes its edits)
        if kind in ("sps", "pps") and int(parsed["rc"][k]) >= 0:
            off = int(parsed["struct_off"][k])
            last[kind] = structs[off:off + slot_bytes(kind)].copy()


def test_writers_match_reference(orc):
    vectors = json.load(gzip.open(os.path.join(HERE, "golden", "write_vectors.json.gz"), "rt"))
    assert len(vectors) >= 8
    for v in vectors:
        run_sequence(orc, v["steps"])
