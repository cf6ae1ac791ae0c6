"""Differential run of hbs_index_parse (no RBSP arena) against hbs_index_extract + hbs_parse_headers on the generators of
fuzz_gpu_parse.py, with payloads appended so that NALs are longer than their windows, windows of several sizes (dev aid).
usage: python3 tests/tools/fuzz_gpu_index_parse.py [first_seed] [count]"""
import sys
import numpy as np
sys.path.insert(0, ".")
import hevcbitstream_amd as hbs
from tests.hevc_synth import annexb
from tests.test_sim_parse_logic import broken, sequence
from tests.test_gpu_index_parse import both_ways, same, INT_MIN

first = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
ctx = hbs.Context(0)
bad = reported = 0
for seed in range(first, first + count):
    rng = np.random.RandomState(seed)
    seq = sequence(seed)
    if seed % 3 == 0:
        seq = broken(seq, np.random.RandomState(7 * seed + 1), lambda t: True)
    # payload behind every NAL (zero-heavy now and then: emulation prevention bytes right behind the headers after re-escaping
    # would change the NALs, so plain bytes above 3 are used where the NAL must stay what the generator made)
    fat = []
    for nal in seq:
        extra = rng.randint(0, 3000)
        fat.append(bytes(nal) + bytes(rng.randint(4, 256, size=extra).astype(np.uint8)))
    stream = annexb(fat)
    for window in (0, 64, 128, 2048):
        try:
            a, b = both_ways(ctx, stream, window=window)
            if int(b[4]["error"]) == -4:
                rep = b[1]["rc"] == INT_MIN
                assert rep.any()
                ok = ~rep
                for f in a[2].dtype.names:
                    assert np.array_equal(a[2][f][ok], b[1][f][ok]), f
                reported += 1
            else:
                same(a, b)
        except AssertionError as e:
            bad += 1
            print("MISMATCH seed", seed, "window", window, str(e)[:200])
print("seeds", first, "..", first + count - 1, "mismatches", bad, "runs with a window reported too small", reported)
