#!/usr/bin/env python3
"""Learn the names the reference's read_debug_* readers print for every read site of
hevcbitstream_amd/csrc/hbs_parse.h, and write hevcbitstream_amd/csrc/hbs_trace_names.h.

The device parser logs (site, bit position, value) per syntax element; the reference's CLI
(oracle/_ref/hevc_analyze_ref, built from /root/reference by oracle/Makefile) prints
"<byte>.<bits left>: <name>: <value>" per element, in the same order because both read the same
bits.  Aligning the two on a set of training streams gives site -> name.  Run in the dev container
(needs oracle/_ref); re-run whenever lines of hbs_parse.h move.  Also writes the golden traces
tests/golden/trace_*.txt that tests/test_sim_trace.py compares with.

usage: python tests/golden/make_trace_names.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests import _sim                                             # noqa: E402
from tests.hevc_synth import annexb, stream_4k30                   # noqa: E402
from tests.test_sim_parse_logic import sequence                    # noqa: E402

EXE = os.path.join(ROOT, "oracle", "_ref", "hevc_analyze_ref")
VERBOSE = "-v" in sys.argv


def reference_trace(stream):
    """[[(byte, bits_left, name, value), ...] per NAL] as printed by the reference CLI"""
    with tempfile.NamedTemporaryFile(suffix=".hevc", delete=False) as f:
        f.write(bytes(stream))
        path = f.name
    try:
        # unbuffered: the reference CLI can crash on streams outside its envelope (App. D); keep what it printed
        txt = subprocess.run(["stdbuf", "-o0", EXE, path], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode("latin-1")
    finally:
        os.unlink(path)
    nals, cur = [], None
    for line in txt.splitlines():
        if line.startswith("!! Found NAL"):
            cur = []
            nals.append(cur)
            continue
        if cur is None or ": " not in line:
            continue
        head, rest = line.split(": ", 1)
        if "." not in head or not head.replace(".", "").isdigit():
            continue
        if ": " not in rest:
            continue
        name, value = rest.rsplit(": ", 1)
        byte, left = head.split(".")
        try:
            # a cursor printed without a value (hevc_stream.c:3147) glues itself to the next line
            while ": " in name and name.split(": ", 1)[0].replace(".", "").isdigit() and "." in name.split(": ", 1)[0]:
                cur.append((int(byte), int(left), "", 0))
                h2, name = name.split(": ", 1)
                byte, left = h2.split(".")
            cur.append((int(byte), int(left), name, int(value.strip())))
        except ValueError:
            pass
    return nals, txt


def our_trace(stream):
    s = np.frombuffer(bytes(stream), dtype=np.uint8)
    idx, arena, _ = _sim.index_extract(s)
    parsed, structs, recs = _sim.parse_trace(arena, idx)
    return idx, parsed, recs


def training_streams():
    for seed in range(150):
        yield "seq%d" % seed, annexb(sequence(seed))
    s, _ = stream_4k30(3, n_pictures=40, slices_per_picture=4, idr_every=10, payload_bytes=(20, 60), rich=True)
    yield "4k30", s


def main():
    votes = {}                 # site -> {name: count}
    problems = 0
    nal_total = nal_ok = 0
    for tag, stream in training_streams():
        ref, _ = reference_trace(stream)
        idx, parsed, recs = our_trace(stream)
        if len(ref) != len(recs):
            ref = ref[:-1]                                       # the CLI died in its last NAL: trust the ones before
            print(tag, "reference stopped after", len(ref), "of", len(recs), "NALs")
        for k, (rl, ol) in enumerate(zip(ref, recs)):
            rl = rl[4:]                                          # the four NAL-header lines are printed by the wrapper
            nal_total += 1
            good = len(rl) == len(ol)
            for (byte, left, name, value), r in zip(rl, ol):
                pos = int(r["pos"])
                if (pos >> 3, 8 - (pos & 7)) != (byte, left) or int(r["value"]) != value:
                    good = False                                 # outside the envelope from here on
                    break
                v = votes.setdefault(int(r["site"]), {})
                v[name] = v.get(name, 0) + 1
            nal_ok += good
            if not good and VERBOSE:
                print(tag, "NAL", k, "type", int(parsed["nal_unit_type"][k]), "diverges")
    names = {}
    for site, v in votes.items():
        best = max(v, key=v.get)
        names[site] = best
        if len(v) > 1:
            print("site", site, "line", site // 8, "votes", v)
            problems += sum(c for n, c in v.items() if n != best) * 10 > v[best]
    print("NALs whose whole trace matches:", nal_ok, "of", nal_total)
    print("sites named:", len(names), "problems:", problems)
    out = os.path.join(ROOT, "hevcbitstream_amd", "csrc", "hbs_trace_names.h")
    with open(out, "w") as f:
        f.write("/* hbs_trace_names.h -- GENERATED by tests/golden/make_trace_names.py: what the reference's read_debug_*\n"
                " * readers (hevc_stream.c:2343-3434) print for each read site of hbs_parse.h (site = line * 8 + ordinal on\n"
                " * the line), learnt by aligning the parser's trace with the reference CLI's output on the training\n"
                " * streams of that script.  Re-run it whenever lines of hbs_parse.h move. */\n"
                "#ifndef HBS_TRACE_NAMES_H\n#define HBS_TRACE_NAMES_H\n\n"
                "static const struct { unsigned site; const char* name; } hbs_trace_names[] = {\n")
        for site in sorted(names):
            f.write('    { %d, "%s" },\n' % (site, names[site].replace("\\", "\\\\").replace('"', '\\"')))
        f.write("};\n\n#endif\n")
    print("wrote", out)

    # golden CLI outputs for the tests: streams inside the envelope (the reference CLI survives them and every
    # NAL's trace lines up), kept whole: "!! Found NAL" lines, hex dumps and field lines
    import gzip
    import json
    vectors = []
    chosen = [("seq%d" % sd, annexb(sequence(sd))) for sd in (0, 1, 2, 3, 5, 8, 13, 21, 34, 55)]
    chosen.append(("4k30-plain", stream_4k30(5, n_pictures=6, slices_per_picture=3, idr_every=3, payload_bytes=(20, 60))[0]))
    for tag, stream in chosen:
        ref, txt = reference_trace(stream)
        idx, parsed, recs = our_trace(stream)
        ok = len(ref) == len(recs)
        for rl, ol in zip(ref, recs):
            rl = rl[4:]
            ok = ok and len(rl) == len(ol) and all((int(r["pos"]) >> 3, 8 - (int(r["pos"]) & 7), int(r["value"])) == (b, l, v) for (b, l, n, v), r in zip(rl, ol))
        if ok:
            vectors.append({"tag": tag, "stream": bytes(stream).hex(), "stdout": txt})
        else:
            print("not used as a golden (outside the envelope):", tag)
    with gzip.open(os.path.join(HERE, "trace_vectors.json.gz"), "wt") as f:
        json.dump(vectors, f)
    print("golden CLI outputs:", [v["tag"] for v in vectors])
    return problems


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
