"""The opt-in extension (NAL types 35..40: AUD, EOS, EOB, filler data, SEI).
  - the oracle's restatement (oracle/hbs_oracle_parse.c: orc_read_extended_nal) against the golden vectors the REAL
    reference's never-dispatched readers produced (tests/golden/ext_vectors.json, made by make_golden_ext.py);
  - the product's byte-level reader (hbs_parse_ext.h, single-stepped on the CPU) against the oracle on the same NALs
    and on random ones;
  - `-m ref`: the oracle against the reference driver itself on random NALs (dev container only);
  - `-m gpu`: hbs_parse_extended through the C ABI, a stream of mixed NALs, against the oracle per NAL."""
import ctypes as C
import json
import os
import random

import numpy as np
import pytest

from tests import _orc, _sim

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
EXT = np.dtype([("num_sei_messages", "<i4"), ("primary_pic_type", "<i4"), ("filler_bytes", "<u4"), ("reserved", "<u4"),
                ("sei", [("payloadType", "<i4"), ("payloadSize", "<i4"), ("payload_off", "<u4"), ("reserved", "<u4")], (6,))])


def oracle_ext(orc, nal):
    out = np.zeros(1, dtype=EXT)
    t = C.c_int(0)
    orc.lib.orc_read_extended_nal.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
    rc = orc.lib.orc_read_extended_nal(bytes(nal), len(nal), out.ctypes.data, C.byref(t))
    return rc, t.value, out[0]


def as_tuple(rc, rec):
    n = int(rec["num_sei_messages"])
    return (rc, int(rec["primary_pic_type"]), int(rec["filler_bytes"]), n,
            [(int(m["payloadType"]), int(m["payloadSize"]), int(m["payload_off"])) for m in rec["sei"][: min(n, 6)]])


def random_nals(seed, count):
    import tests.golden.make_golden_ext as g
    rng = random.Random(seed)
    out = []
    for _ in range(count):
        t = rng.choice((35, 36, 37, 38, 39, 40))
        body = bytes(rng.choice((0, 0, 0xff, 0xff, 0x80, 1, 2, 3, rng.randrange(256))) for _ in range(rng.randrange(0, 60)))
        out.append(g.to_nal(g.header(t, rng.randrange(64), rng.randrange(8)) + body))
    return out


def test_oracle_against_the_reference_vectors(orc):
    vec = json.load(open(os.path.join(HERE, "golden", "ext_vectors.json")))["vectors"]
    assert len(vec) > 200
    for v in vec:
        nal = bytes.fromhex(v["nal"])
        rc, t, rec = oracle_ext(orc, nal)
        assert t == v["type"], v
        assert rc == v["rc"], v
        if rc == -2 or t < 0:
            continue
        want = (v["rc"], v["primary_pic_type"], v["filler_bytes"], v["num_sei_messages"], [tuple(m) for m in v["sei"]])
        assert as_tuple(rc, rec) == want, (v, as_tuple(rc, rec))


def product_reader_on_cpu(nal):
    """nal_to_rbsp by the oracle, then the product's reader (sim) on the RBSP"""
    orc = _orc.oracle()
    r, consumed, rbsp_size, data = orc.nal_to_rbsp(nal)
    if r < 0:
        return -1, None
    rbsp = np.frombuffer(data + b"\xee" * 8, dtype=np.uint8).copy()
    out = np.zeros(1, dtype=EXT)
    t = C.c_int(0)
    lib = _sim.lib()
    lib.sim_read_extended_nal.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
    rc = lib.sim_read_extended_nal(rbsp.ctypes.data, rbsp_size, consumed, out.ctypes.data, C.byref(t))
    return rc, out[0]


def test_product_reader_single_stepped_against_the_oracle(orc):
    vec = [bytes.fromhex(v["nal"]) for v in json.load(open(os.path.join(HERE, "golden", "ext_vectors.json")))["vectors"]]
    for nal in vec + random_nals(7, 3000):
        rc, t, rec = oracle_ext(orc, nal)
        got_rc, got = product_reader_on_cpu(nal)
        if t < 0:                      # nal_to_rbsp rejected the NAL
            assert got_rc == -1
            continue
        assert got_rc == rc, (nal.hex(), got_rc, rc)
        if rc != -2:
            assert as_tuple(got_rc, got) == as_tuple(rc, rec), nal.hex()


@pytest.mark.ref
def test_oracle_against_the_reference_driver_fuzz(orc):
    drv = os.path.join(ROOT, "oracle", "_ref", "libref_ext_driver.so")
    if not os.path.exists(drv):
        pytest.skip("reference build not present")
    lib = C.CDLL(drv)
    lib.ref_read_extended_nal.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.POINTER(C.c_int)]
    for nal in random_nals(11, 6000):
        out = np.zeros(1, dtype=EXT)
        t = C.c_int(0)
        rc = lib.ref_read_extended_nal(bytes(nal), len(nal), out.ctypes.data, C.byref(t))
        orc_rc, orc_t, rec = oracle_ext(orc, nal)
        assert (rc, t.value) == (orc_rc, orc_t), nal.hex()
        if t.value >= 0 and rc != -2:
            assert as_tuple(rc, out[0]) == as_tuple(orc_rc, rec), nal.hex()


@pytest.mark.gpu
def test_gpu_parse_extended_against_the_oracle(orc):
    import torch
    import hevcbitstream_amd as hbs
    from tests.hevc_synth import annexb
    from tests.test_sim_parse_logic import sequence
    vec = [bytes.fromhex(v["nal"]) for v in json.load(open(os.path.join(HERE, "golden", "ext_vectors.json")))["vectors"]]
    rng = random.Random(5)
    nals = []
    for nal in vec + random_nals(13, 4000):
        # a NAL must survive as one NAL inside an Annex-B stream: no trailing zero bytes (they would belong to the next start code),
        # no 00 00 0x inside (to_nal() took care of the generated ones; some hand-made vectors are for the single-NAL path only)
        if len(nal) < 2 or nal[-1] == 0 or b"\x00\x00\x00" in nal or b"\x00\x00\x01" in nal or b"\x00\x00\x02" in nal:
            continue
        nals.append(nal)
        if rng.random() < 0.1:
            nals += sequence(rng.randrange(50))[:4]            # parameter sets and slices in between
    stream = np.frombuffer(annexb(nals), dtype=np.uint8).copy()
    ctx = hbs.Context(0)
    index, rbsp, summary, cap = ctx.alloc_outputs(len(stream))
    ctx.index_extract_async(torch.from_numpy(stream).cuda(), index, cap, rbsp, summary)
    n = int(ctx.read_summary(summary)["nal_count"])
    assert n == len(nals)
    parsed_dev = torch.empty(n * hbs.PARSED.itemsize, dtype=torch.uint8, device="cuda")
    psum = torch.zeros(hbs.SUMMARY.itemsize, dtype=torch.uint8, device="cuda")
    ctx.parse_headers_async(rbsp, index, n, parsed_dev, None, psum)          # plan: types; the default leaves rc = -1 for 35..40
    before = parsed_dev.cpu().numpy().view(hbs.PARSED).copy()
    ext = ctx.parse_extended(rbsp, index, n, parsed_dev)
    after = parsed_dev.cpu().numpy().view(hbs.PARSED)
    checked = 0
    for k, nal in enumerate(nals):
        rc, t, rec = oracle_ext(orc, nal)
        assert int(after["nal_unit_type"][k]) == t or t < 0
        if rc == -2 or t < 0:
            assert int(after["rc"][k]) == int(before["rc"][k])             # not an extended type: left alone
            assert int(ext["num_sei_messages"][k]) == 0
            continue
        assert int(before["rc"][k]) == -1                                     # hevc_stream.c:221-222
        assert as_tuple(int(after["rc"][k]), ext[k]) == as_tuple(rc, rec), (k, nal.hex())
        checked += 1
    assert checked > 3000
    ctx.close()
