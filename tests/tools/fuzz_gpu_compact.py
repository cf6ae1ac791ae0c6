"""Extended differential run of the compact header parse (hbs_parse_headers_compact / hbs_parse_materialize) against the full parse
(which tests/tools/fuzz_gpu_parse.py runs against the oracle): random rich sequences, damaged slices, damaged parameter sets
(out-of-spec streams: the exact re-walk into lane-owned slots), concatenations; every third stream with a random list of NALs
to materialise (dev aid; the committed tests run a subset).  usage: python3 tests/tools/fuzz_gpu_compact.py [first_seed] [count]"""
import sys
import numpy as np
sys.path.insert(0, ".")
import hevcbitstream_amd as hbs
from tests.hevc_synth import annexb
from tests.test_gpu_compact import both, check
from tests.test_sim_parse_logic import broken, sequence

first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
ctx = hbs.Context(0)
bad = streams = deep = 0


def run(nals, tag, want=None):
    global bad, streams, deep
    streams += 1
    try:
        _, _, _, n, fp, fs, cp, cc, cs, w, _ = both(ctx, annexb(nals), want=want)
        assert n == len(nals)
        check(fp, fs, cp, cc, cs, want=None if w is None else sorted({k for k in w if k < n}))
    except hbs.HbsError as e:
        if getattr(e, "code", 0) == -6:          # HBS_E_DEPTH: reported, by design (the full parse answers such a stream in order)
            deep += 1
            return
        bad += 1
        print("ERROR", tag, e)
    except AssertionError as e:
        bad += 1
        print("MISMATCH", tag, str(e)[:300])


for seed in range(first, first + count):
    rng = np.random.RandomState(seed)
    pick = (lambda fp: rng.randint(0, len(fp) + 3, size=rng.randint(0, 12)).tolist()) if seed % 3 == 0 else None
    run(sequence(seed), ("sequence", seed), pick)
    run(broken(sequence(seed), np.random.RandomState(7 * seed + 1), lambda t: t not in (33, 34)), ("broken slices", seed), pick)
    if seed % 2 == 0:
        run(broken(sequence(seed), np.random.RandomState(7 * seed + 2), lambda t: True), ("broken anything", seed), pick)
    if seed % 10 == 0:
        nals = []
        for s2 in range(seed, seed + 12):
            nals += broken(sequence(s2), np.random.RandomState(s2), lambda t: True) if s2 % 2 else sequence(s2)
        run(nals, ("concatenated", seed), pick)
print("seeds", first, "..", first + count - 1, "streams", streams, "mismatches", bad, "reported too deep for the compact parse", deep)
