"""Shared comparison of a batch header parse (K4: on the GPU or single-stepped
on the CPU) with the oracle parser run NAL by NAL.  Test infrastructure."""
import numpy as np

from tests import _orc

PARSED = np.dtype([("rc", "<i4"), ("nal_unit_type", "<i4"), ("nal_layer_id", "<i4"), ("nal_temporal_id_plus1", "<i4"),
                   ("struct_off", "<u8"), ("slice_data_size", "<i4"), ("slice_data_off", "<u4")])
NONE = np.uint64(0xFFFFFFFFFFFFFFFF)


def which_struct(t):
    if 0 <= t <= 9 or 16 <= t <= 21:
        return "sh"
    return {32: "vps", 33: "sps", 34: "pps"}.get(t)


def oracle_pass(nals, parser=None, timing=None):
    """Feed the NALs, in order, to the oracle's read_hevc_nal_unit restatement (or to `parser`, e.g. _orc.ReferenceHevc():
    the compiled reference itself).  timing: a list that receives the seconds spent inside read() alone."""
    import time
    o = parser if parser is not None else _orc.OracleHevc()
    exp = []
    spent = 0.0
    for nal in nals:
        t0 = time.perf_counter()
        rc = o.read(nal)
        spent += time.perf_counter() - t0
        nalhdr = o.v["nal"].copy()
        t = int(nalhdr[1])
        rec = {"rc": rc, "nal": nalhdr}
        k = which_struct(t)
        if k is not None:
            rec["kind"] = k
            rec["struct"] = o.v[k].copy()
            if k == "sh" and rc >= 0:
                rec["slice_data"] = o.slice_data()
        exp.append(rec)
    o.close()
    if timing is not None:
        timing.append(spent)
    return exp


def compare(parsed, structs, rbsp, idx, exp):
    """parsed: ndarray[PARSED]; structs, rbsp: uint8 arrays; idx: NAL entries; exp: oracle_pass()."""
    assert len(parsed) == len(exp)
    prev_nal = None
    for k, (p, e) in enumerate(zip(parsed, exp)):
        assert int(p["rc"]) == e["rc"], (k, int(p["rc"]), e["rc"])
        failed_early = bool(idx["status"][k] & 1)
        if not failed_early:
            assert [0, int(p["nal_unit_type"]), int(p["nal_layer_id"]), int(p["nal_temporal_id_plus1"])] == list(e["nal"]), k
        if "kind" in e and not failed_early:
            size = _orc.layout()[_orc.STRUCT_TYPES[e["kind"]]]["size"]
            off = int(p["struct_off"])
            assert p["struct_off"] != NONE, k
            got = structs[off:off + size].view(np.int32)
            if not np.array_equal(got, e["struct"]):
                names = _orc.flat_fields(_orc.STRUCT_TYPES[e["kind"]])
                for name, i, c in names:
                    if not np.array_equal(got[i:i + c], e["struct"][i:i + c]):
                        raise AssertionError("NAL %d %s.%s: got %s want %s" % (k, e["kind"], name, got[i:i + c][:8], e["struct"][i:i + c][:8]))
            if "slice_data" in e:
                size_want, data = e["slice_data"]
                assert int(p["slice_data_size"]) == size_want, k
                if data is not None:
                    a = int(idx["rbsp_off"][k]) + int(p["slice_data_off"])
                    assert bytes(rbsp[a:a + size_want]) == data, k
