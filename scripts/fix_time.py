"""round 4: the header parse of a 100 k-NAL 4K30-style batch in which ~1 % of the slices are out of spec in the way that makes
their header depend on NALs in front of their SPS (an IDR coded as a P slice asks for the RPS row the last slice with its own
set left behind).  Until round 4 one such slice sent the whole batch through k4_seq (81 k NAL/s); now only those slices are
walked again (hbs_parse_fix.h).  Every NAL is compared with the oracle's sequential parser.  usage: fix_time.py [pictures] [every]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import hevcbitstream_amd as hbs
from hevcbitstream_amd.api import PARSED, SUMMARY
from hevcbitstream_amd.hevc_synth import stream_4k30

pics = int(sys.argv[1]) if len(sys.argv) > 1 else 12500
every = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ctx = hbs.Context(0)
for ev in (0, every):
    stream, n = stream_4k30(11, n_pictures=pics, slices_per_picture=8, idr_every=60, payload_bytes=(60, 120), forbidden_every=ev)
    s8 = np.frombuffer(stream, dtype=np.uint8).copy()
    d = torch.from_numpy(s8).cuda()
    index, rbsp, summ, cap = ctx.alloc_outputs(d.numel())
    ctx.index_extract_async(d, index, cap, rbsp, summ)
    m = int(ctx.read_summary(summ)["nal_count"])
    assert m == n
    parsed, structs = ctx.parse_headers(rbsp, index, m)
    pt = torch.empty(m * PARSED.itemsize, dtype=torch.uint8, device="cuda")
    sm = torch.zeros(SUMMARY.itemsize, dtype=torch.uint8, device="cuda")
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    for i in range(6):
        ctx.parse_headers_async(rbsp, index, m, pt, structs, sm)
        evs[i].record()
    torch.cuda.synchronize()
    ms = min(evs[i].elapsed_time(evs[i + 1]) for i in range(5))
    line = "forbidden every %d: %d NALs, parse %.3f ms = %.1f M NAL/s" % (ev, m, ms, m / ms / 1e3)
    if ev:
        ctx.set_sequential_parse(True)
        t0 = time.perf_counter()
        ctx.parse_headers_async(rbsp, index, m, pt, structs, sm)
        torch.cuda.synchronize()
        line += "; the whole batch in order (k4_seq): %.1f ms" % ((time.perf_counter() - t0) * 1e3)
        ctx.set_sequential_parse(False)
        parsed, structs = ctx.parse_headers(rbsp, index, m)
    if os.environ.get("HBS_FIX_CHECK", "1") != "0":
        from tests._parsecmp import compare, oracle_pass
        idx = index[: m * 32].cpu().numpy().view(hbs.NAL_ENTRY)
        nals = [bytes(s8[int(a):int(b)]) for a, b in zip(idx["start"], idx["end"])]
        arena = rbsp[: int(idx["rbsp_off"][-1]) + int(idx["rbsp_len"][-1])].cpu().numpy()
        compare(parsed, structs.cpu().numpy(), arena, idx, oracle_pass(nals))
        line += "; all %d NALs equal to the oracle's sequential parse" % m
    print(line, flush=True)
