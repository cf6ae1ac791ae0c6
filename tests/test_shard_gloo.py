"""N > 1 path on CPU: two processes, gloo backend.  Each rank owns an independent
stream shard; the gathered + rebased index must equal the index of the
concatenated stream.  (On the GPU box the same code runs over RCCL; the shard's
index then comes from hbs_index_extract instead of the oracle.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import _orc
    from hevcbitstream_amd import shard
    orc = _orc.oracle()
    n = 7 + rank                                           # ragged: shards of different NAL counts
    stream, idx, arena = orc.gen_stream(shard.shard_seed(0x1234, rank), n, rank % 2)
    cap = 16
    local = np.zeros(cap, dtype=_orc.NAL_ENTRY)
    local[:n] = idx
    t = torch.from_numpy(local.view(np.uint8).copy())
    all_index, meta = shard.gather_index(torch, dist, t, n, len(stream), len(arena), cap)
    glob = shard.global_entries(all_index, meta)
    # the pipelined form of the same exchange: three steps in flight over two slots, each with its own local buffer
    gat = shard.IndexGatherer(torch, dist, cap, torch.device("cpu"), depth=2)
    bufs, slots = [], []
    for step in range(3):
        b = t.clone()
        if step == 1:
            b[:32] = 0xEE                                  # a different payload per step: results must not mix
        bufs.append(b)
        slots.append(gat.submit(b, n, len(stream), len(arena)))
        if step == 1:
            got1, meta1 = gat.result(slots[1])
            mine = got1[rank]
            assert bytes(mine[:32].tolist()) == b"\xEE" * 32 and torch.equal(mine[32:], t[32:])
    got2, meta2 = gat.result(slots[2])
    gat.drain()
    assert torch.equal(got2, all_index) and torch.equal(meta2, meta)
    # every rank got the same thing; rank 0 checks it against the concatenated stream
    streams = [None] * world
    dist.all_gather_object(streams, stream.tobytes())
    if rank == 0:
        cat = np.frombuffer(b"".join(streams), dtype=np.uint8)
        want, want_arena, why = orc.index_extract(cat)
        ok = len(glob) == len(want) and all(np.array_equal(glob[f], want[f]) for f in ("start", "end", "rbsp_off", "rbsp_len"))
        ret.put(bool(ok) and int(meta[:, 0].sum()) == len(want))
    dist.barrier()
    dist.destroy_process_group()


def _worker_parts(rank, world, port, ret):
    """ONE stream over two ranks: each takes its part (cut at a start code, hbs_find_cut_host), scans it with the halo behind it
    (the oracle stands in for the GPU scan here), drops the NAL the halo opens, and the parts' entries meet with their cut
    offsets added -- counts first, then exactly count x 32 bytes per rank."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import _orc
    from hevcbitstream_amd import shard
    orc = _orc.oracle()
    ok = True
    for seed, mode in ((0x600, 0), (0x601, 1), (0x602, 0)):
        stream, idx, arena = orc.gen_stream(seed, 60, mode)                 # the same bytes on both ranks (a shared file)
        lo, hi, hh = shard.part_ranges(stream, world)[rank]
        e, a, why = orc.index_extract(stream[lo:hh])
        e = shard.trim_part(e, hi - lo, is_last=(rank == world - 1))
        rbsp_mine = int(e["rbsp_off"][-1] + e["rbsp_len"][-1]) if len(e) else 0
        # counts (and the RBSP bytes of the parts in front) first
        meta = torch.zeros(world * 2, dtype=torch.int64)
        dist.all_gather_into_tensor(meta, torch.tensor([len(e), rbsp_mine], dtype=torch.int64))
        meta = meta.view(world, 2)
        rbsp_base = int(meta[:rank, 1].sum())
        e = e.copy()
        e["start"] += np.uint64(lo); e["end"] += np.uint64(lo); e["rbsp_off"] += np.uint64(rbsp_base)
        # then exactly count x 32 bytes per rank
        pieces = []
        for r in range(world):
            buf = torch.from_numpy(e.view(np.uint8).copy()) if r == rank else torch.empty(int(meta[r, 0]) * 32, dtype=torch.uint8)
            dist.broadcast(buf, src=r)
            pieces.append(buf.numpy().view(_orc.NAL_ENTRY))
        glob = np.concatenate(pieces)
        want, want_arena, w = orc.index_extract(stream)
        ok = ok and len(glob) == len(want) and all(np.array_equal(glob[f], want[f]) for f in ("start", "end", "rbsp_off", "rbsp_len"))
    if rank == 0:
        ret.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_parts_of_one_stream():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_worker_parts, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(timeout=5) is True


def test_two_rank_index_gather():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(timeout=5) is True


_RANK_SCRIPT = '''
import json, os, sys
import torch, torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["LOCAL_RANK"]) == rank
if "--die" in sys.argv and rank == 1:
    sys.exit(7)
dist.init_process_group("gloo")
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t)
print("rank %d chatter" % rank)
if rank == 0:
    print(json.dumps({"metric": "m", "n_gpus": world, "sum": t.item(), "argv": sys.argv[1:]}))
dist.barrier()
dist.destroy_process_group()
'''


def test_bench_launcher_starts_the_ranks_itself(tmp_path):
    """`python3 bench.py --gpus N` without WORLD_SIZE (the driver's command): bench.launch_ranks starts one child per rank with
    the torch.distributed.run environment on 127.0.0.1, relays rank 0's JSON line as the LAST and ONLY stdout line, sends the
    rest to stderr, and exits non-zero when a rank fails (the others are ended, nobody hangs).  The rank script here is a
    gloo stand-in: bench.py's own ranks need a GPU (tests/test_gpu_shard.py runs the real thing through the same launcher)."""
    import json
    import subprocess
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT)
    drv = ("import sys; sys.path.insert(0, %r); import bench; "
           "sys.exit(bench.launch_ranks(2, sys.argv[1:], script=%r))" % (ROOT, str(script)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, "-c", drv, "--gpus", "2", "--steps", "3"], env=env, capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = p.stdout.decode().strip().splitlines()
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["sum"] == 3.0 and line["argv"] == ["--gpus", "2", "--steps", "3"]
    assert "rank 0 chatter" in p.stderr.decode() and "rank 1 chatter" in p.stderr.decode()
    t0 = __import__("time").time()
    p = subprocess.run([sys.executable, "-c", drv, "--die"], env=env, capture_output=True, timeout=300)
    assert p.returncode == 7 and p.stdout.decode().strip() == "", (p.returncode, p.stdout)
    assert __import__("time").time() - t0 < 120


def test_bench_gpus_1_does_not_launch():
    """--gpus 1 (and a run under torch.distributed.run, WORLD_SIZE set) never goes through the launcher"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'if args.gpus > 1 and "WORLD_SIZE" not in os.environ:' in src
    head = src[: src.index("def launch_ranks")]
    assert "import torch" not in head, "nothing may touch torch / HIP before the launcher decision"
