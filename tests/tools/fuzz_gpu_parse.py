"""Extended differential run of the GPU header parse against the oracle (dev aid; the committed tests run a
subset of the same generators).  usage: python3 tests/tools/fuzz_gpu_parse.py [first_seed] [count]"""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
import hevcbitstream_amd as hbs
from tests._parsecmp import compare, oracle_pass
from tests.hevc_synth import Synth, annexb
from tests.test_sim_parse_logic import broken, sequence

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 600
ctx = hbs.Context(0)
import os
if os.environ.get("HBS_FUZZ_SEQ"):                       # the opt-in sequential walk: expected to match everywhere
    ctx.set_sequential_parse(True)


def gpu_parse(stream_bytes):
    s = np.frombuffer(stream_bytes, dtype=np.uint8).copy()
    d = torch.from_numpy(s).cuda()
    index, rbsp, summary, cap = ctx.alloc_outputs(d.numel())
    ctx.index_extract_async(d, index, cap, rbsp, summary)
    sm = ctx.read_summary(summary)
    n = int(sm["nal_count"])
    parsed, structs = ctx.parse_headers(rbsp, index, n, poison=0x5A)
    idx = index[: n * 32].cpu().numpy().view(hbs.NAL_ENTRY)
    return s, idx, rbsp[: int(sm["rbsp_bytes"])].cpu().numpy(), parsed, structs.cpu().numpy()


def run(nals, tag):
    s, idx, arena, parsed, structs = gpu_parse(annexb(nals))
    assert len(idx) == len(nals), (tag, len(idx), len(nals))
    try:
        compare(parsed, structs, arena, idx, oracle_pass(nals))
    except AssertionError as e:
        print("MISMATCH", tag, str(e)[:300])
        return 1
    return 0


bad = 0
longest = 0
for seed in range(first, first + count):
    seq = sequence(seed)
    longest = max(longest, max(len(x) for x in seq))
    bad += run(seq, ("sequence", seed))
    bad += run(broken(sequence(seed), np.random.RandomState(7 * seed + 1), lambda t: t not in (33, 34)), ("broken slices", seed))
    if seed % 4 == 0:
        bad += run(broken(sequence(seed), np.random.RandomState(7 * seed + 2), lambda t: True), ("broken anything", seed))
    if seed % 10 == 0:   # many sequences in one stream: 64 NALs per wavefront with mixed types, contexts changing under way
        nals = []
        for s2 in range(seed, seed + 12):
            nals += sequence(s2)
        bad += run(nals, ("concatenated", seed))
print("seeds", first, "..", first + count - 1, "mismatches", bad, "longest NAL", longest)
