#!/usr/bin/env python3
"""BASELINE config 3 by itself (dev aid; bench.py's other_kernels.config3_end_to_end without the comparisons): the ~100 k NALs of
the synthetic 4K30 sequence with slice payloads of 16-28 KiB (2.1 GiB), timed as hbs_index_parse and hbs_index_parse_compact
(host wall time per call), and the header parses alone.  Under `rocprofv3 --kernel-trace` + scripts/trace_call.py: one call's kernels."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    import torch
    import hevcbitstream_amd as hbs
    from hevcbitstream_amd.api import COMPACT, NAL_ENTRY, PARSED, SUMMARY
    from hevcbitstream_amd.hevc_synth import stream_4k30
    ctx = hbs.Context(0)
    stream, _ = stream_4k30(11, n_pictures=12500, slices_per_picture=8, idr_every=60, payload_bytes=(60, 120))
    d = torch.from_numpy(np.frombuffer(stream, dtype=np.uint8).copy()).cuda()
    index, rbsp, summ, cap = ctx.alloc_outputs(d.numel())
    ctx.index_extract_async(d, index, cap, rbsp, summ)
    m = int(ctx.read_summary(summ)["nal_count"])
    parsed, structs = ctx.parse_headers(rbsp, index, m)
    ent = index[: m * 32].cpu().numpy().view(NAL_ENTRY)
    old_off, old_len = ent["rbsp_off"].astype(np.int64), ent["rbsp_len"].astype(np.int64)
    is_slice = parsed["nal_unit_type"] < 32
    rng = np.random.RandomState(3)
    new_len = old_len + np.where(is_slice, rng.randint(16 << 10, 28 << 10, size=m), 0)
    new_off = np.concatenate([[0], np.cumsum(new_len)[:-1]])
    total = int(new_len.sum())
    big = torch.randint(0, 256, (total + 64,), dtype=torch.uint8, device="cuda")
    shift = torch.from_numpy(new_off - old_off).cuda()
    old_total = int(old_off[-1] + old_len[-1])
    pos = torch.repeat_interleave(shift, torch.from_numpy(old_len).cuda()) + torch.arange(old_total, device="cuda")
    big[pos] = rbsp[:old_total]
    big[torch.from_numpy(new_off + new_len - 1).cuda()[torch.from_numpy(is_slice).cuda()]] = 0x80
    ent2 = np.zeros(m, dtype=NAL_ENTRY)
    ent2["rbsp_off"], ent2["rbsp_len"] = new_off, new_len
    idx2 = torch.from_numpy(ent2.view(np.uint8).copy()).cuda()
    stream2 = torch.empty(int(ctx.lib.hbs_annexb_bound(total, m)), dtype=torch.uint8, device="cuda")
    es = torch.zeros(SUMMARY.itemsize, dtype=torch.uint8, device="cuda")
    ctx.emit_annexb_async(big, total, idx2, m, 1, stream2, None, es)
    sb2 = int(ctx.read_summary(es)["stream_bytes"])
    del pos, shift, big
    cap2 = m + 8
    index3 = torch.empty(cap2 * 32, dtype=torch.uint8, device="cuda")
    parsed3 = torch.empty(m * PARSED.itemsize, dtype=torch.uint8, device="cuda")
    cc3 = torch.empty(m * COMPACT.itemsize, dtype=torch.uint8, device="cuda")
    structs3 = torch.empty_like(structs)
    ssum, psum = torch.zeros(SUMMARY.itemsize, dtype=torch.uint8, device="cuda"), torch.zeros(SUMMARY.itemsize, dtype=torch.uint8, device="cuda")
    res = {"stream_bytes": sb2, "nals": m}
    for name, fn in (("index_parse", lambda: ctx.index_parse_async(stream2[:sb2], index3, cap2, parsed3, structs3, ssum, psum)),
                     ("index_parse_compact", lambda: ctx.index_parse_compact_async(stream2[:sb2], index3, cap2, parsed3, cc3, structs3, ssum, psum))):
        ts = []
        for i in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            got = fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        assert got == m and int(ctx.read_summary(psum)["error"]) == 0
        res[name] = {"ms_min": round(min(ts[1:]), 4), "ms_all": [round(t, 4) for t in ts]}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
