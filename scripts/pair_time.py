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

# round 5: the pool.  The arena above is given back and asked for again (its chunks come off the free list: no new chunk, the
# stream's sixteen pieces are probed again), then K3's output against THAT arena (a pool buffer: classes by lookup)
del rbsp2
torch.cuda.synchronize()
t0 = time.perf_counter()
index3, rbsp3, summary3, cap3 = ctx.alloc_outputs(sb, index_cap=n + 8, peer=stream)
torch.cuda.synchronize()
t_alloc2 = time.perf_counter() - t0
rep3 = dict(ctx.last_pair_report)
t_pair2 = k12(rbsp3, index3, cap3, summary3)
assert torch.equal(rbsp3[:rb], g["rbsp"][:rb]), "second paired arena: wrong bytes"
t0 = time.perf_counter()
out4, rep4 = ctx.pair_alloc(rbsp3, sb + 4096)
torch.cuda.synchronize()
t_alloc3 = time.perf_counter() - t0
idx_out = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
for i in range(5):
    ev[i].record()
    if i < 4:
        ctx.emit_annexb_async(rbsp3, rb, index3, n, 1, out4, idx_out, summary3)
torch.cuda.synchronize()
ms4 = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(1, 4))[1]
assert torch.equal(out4[:sb], stream)
print("pool: second arena alloc %.4f s %s K12 %.3f ms (%.4f); K3 output against it alloc %.4f s %s K3 %.3f ms (%.4f)" % (
    t_alloc2, rep3, t_pair2, algo / t_pair2 / 8e9, t_alloc3, rep4, ms4, (rb + sb) / ms4 / 8e9), flush=True)
