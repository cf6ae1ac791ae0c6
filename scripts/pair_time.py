"""round 4: K12 and K3 on the 16 GiB bench stream with the outputs from torch (hipMalloc) and from hbs_pair_alloc, in ONE process:
the placement effect and what the measuring allocator makes of it.  usage: pair_time.py [nals]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hevcbitstream_amd as hbs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_677_000
ctx = hbs.Context(0)
ctx.enable_timing(True)
g = ctx.synth_stream(0x1234, n)
sb, rb = g["stream_bytes"], g["rbsp_bytes"]
stream = g["stream"][:sb]
algo = sb + rb + 32 * n


def k12(rbsp, index, cap, summary):
    ks = []
    for i in range(5):
        ctx.index_extract_async(stream, index, cap, rbsp, summary)
        if i:
            ks.append(ctx.kernel_ms())
    s = ctx.read_summary(summary)
    assert int(s["error"]) == 0 and int(s["nal_count"]) == n and int(s["rbsp_bytes"]) == rb
    ks.sort()
    return ks[len(ks) // 2]


index, rbsp, summary, cap = ctx.alloc_outputs(sb, index_cap=n + 8)
t_plain = k12(rbsp, index, cap, summary)
assert torch.equal(rbsp[:rb], g["rbsp"][:rb])
del rbsp
torch.cuda.synchronize()
t0 = time.perf_counter()
index2, rbsp2, summary2, cap2 = ctx.alloc_outputs(sb, index_cap=n + 8, peer=stream)
torch.cuda.synchronize()
t_alloc = time.perf_counter() - t0
t_pair = k12(rbsp2, index2, cap2, summary2)
assert torch.equal(rbsp2[:rb], g["rbsp"][:rb]), "paired arena: wrong bytes"
print("K12 %d NALs: torch arena %.3f ms (%.4f)  paired arena %.3f ms (%.4f)  alloc %.2f s  report %s" % (
    n, t_plain, algo / t_plain / 8e9, t_pair, algo / t_pair / 8e9, t_alloc, ctx.last_pair_report), flush=True)

# the way back: K3 reads the (generator's) arena, writes a stream
ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
res = []
for kind in ("torch", "paired"):
    if kind == "torch":
        out = torch.empty(sb + 4096, dtype=torch.uint8, device="cuda")
    else:
        out, rep = ctx.pair_alloc(g["rbsp"], sb + 4096)
    idx_out = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
    for i in range(5):
        ev[i].record()
        if i < 4:
            ctx.emit_annexb_async(g["rbsp"], rb, g["index"], n, 1, out, idx_out, summary)
    torch.cuda.synchronize()
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(1, 4))[1]
    assert torch.equal(out[:sb], stream)
    res.append("%s %.3f ms (%.4f)" % (kind, ms, (rb + sb) / ms / 8e9))
    del out
print("K3 emit: " + "  ".join(res) + ("  report %s" % rep), flush=True)
