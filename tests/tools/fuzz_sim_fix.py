"""The exact re-walk of hbs_parse_fix.h single-stepped on the CPU (tests/sim) against the oracle's sequential parser, on the
families of corrupted sequences that tests/tools/fuzz_gpu_parse.py runs on the GPU: sequences, slices damaged, everything
damaged, many sequences in one stream, and a 4K30-style stream in which some slices are made to read rows their SPS does
not have.  usage: python3 tests/tools/fuzz_sim_fix.py [first_seed] [count] [fix_mode]
fix_mode 0: the batch parse alone (mismatches EXPECTED on forbidden streams: the teeth of the test), 1: as the library
runs it (re-walk when a slice raised the flag), 2: re-walk also when none did (is the flag complete?)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests import _sim
from tests._parsecmp import compare, oracle_pass
from tests.hevc_synth import annexb
from tests.test_sim_parse_logic import broken, sequence

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 1
tot = {"streams": 0, "flagged": 0, "rewalked": 0, "deep": 0, "bad": 0}


def run(nals, tag):
    stream = np.frombuffer(annexb(nals), dtype=np.uint8)
    idx, arena, s = _sim.index_extract(stream)
    assert len(idx) == len(nals), (tag, len(idx), len(nals))
    st = []
    parsed, structs = _sim.parse_headers(arena, idx, fix=mode, stats=st)
    tot["streams"] += 1
    tot["flagged"] += 1 if st[0] else 0
    tot["rewalked"] += st[1]
    tot["deep"] += st[2]
    if st[2]:
        return           # a chain deeper than kFixDepth: the library walks such a batch in order (k4_seq), not tested here
    try:
        compare(parsed, structs, arena, idx, oracle_pass(nals))
    except AssertionError as e:
        tot["bad"] += 1
        if tot["bad"] <= 12:
            print("MISMATCH", tag, st, str(e)[:200])


for seed in range(first, first + count):
    run(sequence(seed), ("sequence", seed))
    run(broken(sequence(seed), np.random.RandomState(7 * seed + 1), lambda t: t not in (33, 34)), ("broken slices", seed))
    if seed % 2 == 0:
        run(broken(sequence(seed), np.random.RandomState(7 * seed + 2), lambda t: True), ("broken anything", seed))
    if seed % 5 == 0:
        nals = []
        for s2 in range(seed, seed + 12):
            nals += broken(sequence(s2), np.random.RandomState(7 * s2 + 2), lambda t: True) if s2 % 3 == 0 else sequence(s2)
        run(nals, ("concatenated", seed))
print("seeds", first, "..", first + count - 1, "fix mode", mode, tot)
