"""The per-field trace of the header parser (read_debug_* of the reference, hevc_stream.c:2343-3434),
single-stepped on the CPU: records + hbs_trace_names.h formatted like the reference CLI's stdout and
compared with golden outputs of that CLI (tests/golden/make_trace_names.py)."""
import gzip
import json
import os
import re

import numpy as np

from tests import _sim

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def names():
    out = {}
    for line in open(os.path.join(ROOT, "hevcbitstream_amd", "csrc", "hbs_trace_names.h")):
        m = re.match(r'\s*\{ (\d+), "(.*)" \},', line)
        if m:
            out[int(m.group(1))] = m.group(2).replace('\\"', '"').replace("\\\\", "\\")
    return out


def field_lines(text):
    """the '<byte>.<left>: ...' lines of a CLI stdout, per NAL"""
    nals, cur = [], None
    for line in text.splitlines():
        if line.startswith("!! Found NAL"):
            cur = []
            nals.append(cur)
        elif cur is not None and re.match(r"\d+\.\d: ", line):
            cur.append(line)
    return nals


def our_lines(stream):
    s = np.frombuffer(bytes.fromhex(stream), dtype=np.uint8)
    idx, arena, _ = _sim.index_extract(s)
    parsed, structs, recs = _sim.parse_trace(arena, idx)
    nm = names()
    out = []
    for k, rl in enumerate(recs):
        b0, b1 = int(arena[int(idx["rbsp_off"][k])]), int(arena[int(idx["rbsp_off"][k]) + 1])
        lines = ["0.8: forbidden_zero_bit: %d " % (b0 >> 7), "0.7: nal->nal_unit_type: %d " % ((b0 >> 1) & 63),
                 "0.1: nal->nal_layer_id: %d " % (((b0 & 1) << 5) | (b1 >> 3)), "1.3: nal->nal_temporal_id_plus1: %d " % (b1 & 7)]
        pending = ""
        for r in rl:
            head = "%d.%d: " % (int(r["pos"]) >> 3, 8 - (int(r["pos"]) & 7))
            name = nm[int(r["site"])]
            if name == "":
                pending += head                 # a cursor printed without a value glues itself to the next line
            else:
                lines.append(pending + head + "%s: %d " % (name, int(r["value"])))
                pending = ""
        out.append(lines)
    return out


def test_trace_matches_reference_cli():
    vectors = json.load(gzip.open(os.path.join(HERE, "golden", "trace_vectors.json.gz"), "rt"))
    assert len(vectors) >= 5
    for v in vectors:
        want = field_lines(v["stdout"])
        got = our_lines(v["stream"])
        assert len(got) == len(want), v["tag"]
        for k, (g, w) in enumerate(zip(got, want)):
            assert g == w, (v["tag"], k, [(a, b) for a, b in zip(g, w) if a != b][:3], len(g), len(w))


def test_ten_nal_fixture_trace():
    stream = open(os.path.join(HERE, "golden", "ten_nal.hevc"), "rb").read()
    want = field_lines(open(os.path.join(HERE, "golden", "ten_nal.analyze.txt")).read())
    got = our_lines(stream.hex())
    # the fixture's TRAIL_R slices name pps_id 1, which the reference resolves out of bounds: first four NALs only
    assert got[:4] == want[:4]


def test_every_read_site_is_named():
    src = open(os.path.join(ROOT, "hevcbitstream_amd", "csrc", "hbs_parse.h")).read().split("\n")
    nm = names()
    missing = []
    for ln, line in enumerate(src, 1):
        if "define HBS_SITE" in line:
            continue
        for m in re.finditer(r"HBS_SITE\((\d+)\)", line):
            site = ln * 8 + int(m.group(1))
            # the plain reader's 8-bit sub_layer_level_idc read is never traced (the debug reader takes one bit)
            if site not in nm and not ("b.u8(HBS_SITE(1), 0)" in line and int(m.group(1)) == 1):
                missing.append((ln, line.strip()[:80]))
    assert not missing, ("run tests/golden/make_trace_names.py", missing[:5])
