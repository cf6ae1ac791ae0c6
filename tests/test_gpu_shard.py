"""The index exchange of the N > 1 path on real RCCL: this pool has one GPU per box, so a one-rank `nccl` group
is the most that runs here -- it still puts the collectives on RCCL's stream next to the scan on the context's
stream, which is what the pipelined gatherer has to get right (the 2-rank semantics are covered by the gloo test)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_pipelined_gather_on_rccl_one_rank():
    import torch
    import torch.distributed as dist
    import hevcbitstream_amd as hbs
    from hevcbitstream_amd import shard
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        ctx = hbs.Context(0)
        n = 4000
        g = ctx.synth_stream(0x4321, n, 0)
        sb, rb = g["stream_bytes"], g["rbsp_bytes"]
        stream = g["stream"][:sb]
        index, rbsp, summary, cap = ctx.alloc_outputs(sb, index_cap=n + 8)
        indexes = [index, torch.empty_like(index)]
        gat = shard.IndexGatherer(torch, dist, cap, index.device, depth=2)
        slots = []
        for step in range(5):                                  # scans and gathers in flight together
            buf = indexes[step % 2]
            gat.release(step % 2)                              # the gather of step - 2 is done with this buffer
            ctx.index_extract_async(stream, buf, cap, rbsp, summary)
            slots.append(gat.submit(buf, n, sb, rb))
        all_index, meta = gat.result(slots[-1])
        gat.drain()
        torch.cuda.synchronize()
        assert meta.cpu().tolist() == [[n, sb, rb]]
        got = all_index[0, : n * 32].view(torch.int64).view(n, 4)
        want = g["index"][: n * 32].view(torch.int64).view(n, 4)
        assert torch.equal(got[:, :3], want[:, :3])               # start, end, rbsp_off as the generator laid them out
        glob = shard.global_entries(all_index, meta)
        assert len(glob) == n and int(glob["end"][-1]) == sb
        ctx.close()
    finally:
        dist.destroy_process_group()


def test_library_gather_and_parts_of_one_stream(orc):
    """The C ABI's exchange (hbs_comm_*, hbs_gather_index: RCCL looked up by the library) on a one-rank communicator -- to all and
    to a root -- and ONE stream cut in three parts, each scanned with its halo, trimmed (hbs_trim_part) and gathered with its cut
    offset as base: the concatenation must be the whole stream's index (oracle = reference semantics)."""
    import ctypes as C
    import torch
    import hevcbitstream_amd as hbs
    from hevcbitstream_amd import shard
    ctx = hbs.Context(0)
    comm = shard.LibraryComm(ctx, None, 0, 1)
    try:
        for seed, mode in ((0x77, 0), (0x78, 1)):
            stream, idx, arena = orc.gen_stream(seed, 300, mode)
            want, want_arena, why = orc.index_extract(stream)
            parts = shard.part_ranges(stream, 3)
            assert parts[0][0] == 0 and parts[-1][1] == len(stream) and all(p[0] < p[1] for p in parts)
            pieces, rbsp_base = [], 0
            lib = ctx.lib
            lib.hbs_trim_part.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
            for r, (lo, hi, hh) in enumerate(parts):
                d = torch.from_numpy(stream[lo:hh].copy()).cuda()
                index, rbsp, summary, cap = ctx.alloc_outputs(d.numel())
                ctx.index_extract_async(d, index, cap, rbsp, summary)
                s = ctx.read_summary(summary)
                kept, rkept = C.c_uint64(0), C.c_uint64(0)
                n_found, r_found = int(s["nal_count"]), int(s["rbsp_bytes"])
                if hh > hi:                                      # a halo behind the part: the NAL it opens is the next part's
                    assert lib.hbs_trim_part(ctx.h, C.c_void_p(index.data_ptr()), n_found, r_found, hi - lo, C.byref(kept), C.byref(rkept)) == 0
                    assert kept.value == n_found - 1
                else:
                    kept.value, rkept.value = n_found, r_found
                for root in (-1, 0):
                    all_index = torch.zeros((kept.value + 4) * 32, dtype=torch.uint8, device="cuda")
                    counts = comm.gather_index(index, kept.value, all_index, stream_base=lo, rbsp_base=rbsp_base, root=root)
                    assert counts == [kept.value]
                    got = all_index[: kept.value * 32].cpu().numpy().view(hbs.NAL_ENTRY)
                pieces.append(got.copy())
                # this part's RBSP is the whole stream's, where it belongs
                assert np.array_equal(rbsp[: rkept.value].cpu().numpy(), want_arena[rbsp_base: rbsp_base + rkept.value])
                rbsp_base += rkept.value
            glob = np.concatenate(pieces)
            assert len(glob) == len(want)
            for f in ("start", "end", "rbsp_off", "rbsp_len"):
                assert np.array_equal(glob[f], want[f]), f
            assert np.array_equal(glob["status"][:-1], want["status"][:-1])
            assert rbsp_base == len(want_arena)
    finally:
        comm.close()
        ctx.close()


def test_two_ranks_share_one_gpu(orc, tmp_path):
    """hbs_gather_index / hbs_gather_parts with world = 2, 3 and 8 on ONE GPU (tests/tools/shard_worker.py, one process per rank):
    to all and to every root with ranks of different (and zero) counts; a receiver whose buffer is too small -- EVERY rank gets
    HBS_E_CAPACITY and the communicator stays usable (round 2's advice: the receiver used to return alone, the others hung);
    one stream in parts with an empty NAL in the first or the last part (the whole-stream walk ends there).  RCCL refuses two
    ranks on one device: the library is pointed at tests/sim/libfake_rccl.so, a shared-memory stand-in that turns what real
    RCCL would answer with a hang (a send nobody receives, a rank that left the protocol) into an error."""
    import ctypes as C
    import os
    import subprocess
    import sys
    from multiprocessing import shared_memory
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fake = os.path.join(root, "tests", "sim", "libfake_rccl.so")
    assert os.path.exists(fake), "make -C tests/sim"
    base, idx, _ = orc.gen_stream(0x77, 300, 0, want_arena=False)
    streams = []
    for frac in (None, 0.3, 0.8):
        s = base
        if frac is not None:
            k = int(len(idx) * frac)
            cut = int(idx["start"][k]) - 3                        # in front of NAL k's 00 00 01: an empty NAL in front of it
            s = np.concatenate([base[:cut], np.array([0, 0, 1], dtype=np.uint8), base[cut:]])
            assert len(orc.index_stream(s)[0]) == k
        p = str(tmp_path / ("stream_%s.npy" % frac))
        np.save(p, s)
        streams.append(p)
    lib = C.CDLL(fake)
    lib.fake_rccl_segment_bytes.restype = C.c_uint64
    lib.fake_rccl_segment_bytes.argtypes = [C.c_int, C.c_uint64]
    slot = 4 << 20
    for world in (2, 3, 8):       # 8: a full node -- grouped broadcasts / send-receive with seven peers, ranks with no entries
        shm = shared_memory.SharedMemory(create=True, size=int(lib.fake_rccl_segment_bytes(world, slot)))
        try:
            env = dict(os.environ, HBS_RCCL_LIB=fake, HBS_FAKE_RCCL_SHM="/" + shm.name.lstrip("/"), HBS_FAKE_RCCL_SLOT=str(slot))
            procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "tools", "shard_worker.py"), str(r), str(world)] + streams,
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
            outs = [p.communicate(timeout=300)[0].decode() for p in procs]
            for r, (p, o) in enumerate(zip(procs, outs)):
                assert p.returncode == 0 and ("rank %d ok" % r) in o, "world %d rank %d:\n%s" % (world, r, o[-3000:])
        finally:
            shm.close()
            shm.unlink()


@pytest.mark.parametrize("how,world", [("launcher", 2), ("env", 2), ("launcher", 8)], ids=["launcher", "env", "launcher-world-8"])
def test_bench_with_two_ranks_on_one_gpu(tmp_path, how, world):
    """bench.py's whole N > 1 path with world = 2 -- and, round 6, world = 8 with small shards: what `python3 bench.py --gpus 8`
    does on a node, rehearsed as eight child processes on this one GPU (their persistent kernels share the device: every tile by
    ticket, see hbs_ctx_set_device_exclusive) --: the pipelined C-ABI gather (over the stand-in for RCCL), the
    max-over-ranks timing, the per-rank lines and the checks of the gathered rows -- so that the line is right the first time a
    node with several GPUs runs it (torch.distributed over gloo here: RCCL refuses two ranks on one device)"""
    import ctypes as C
    import json
    import os
    import subprocess
    import sys
    from multiprocessing import shared_memory
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fake = os.path.join(root, "tests", "sim", "libfake_rccl.so")
    assert os.path.exists(fake), "make -C tests/sim"
    lib = C.CDLL(fake)
    lib.fake_rccl_segment_bytes.restype = C.c_uint64
    lib.fake_rccl_segment_bytes.argtypes = [C.c_int, C.c_uint64]
    nals, slot = (40000, 8 << 20) if world == 2 else (12000, 4 << 20)
    shm = shared_memory.SharedMemory(create=True, size=int(lib.fake_rccl_segment_bytes(world, slot)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    try:
        hooks = dict(HBS_RCCL_LIB=fake, HBS_FAKE_RCCL_SHM="/" + shm.name.lstrip("/"), HBS_FAKE_RCCL_SLOT=str(slot),
                     HBS_BENCH_ONE_DEVICE="0", HBS_BENCH_DIST_BACKEND="gloo")
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "4", "--warmup", "1", "--nals", str(nals)]
        if how == "launcher":
            # the driver's command as typed: no RANK / WORLD_SIZE in the environment, bench.py starts its ranks itself
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
            p = subprocess.Popen(cmd, env=dict(env, **hooks), stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=root)
            o, e = p.communicate(timeout=900)
            assert p.returncode == 0, e.decode()[-3000:]
            outs = [(o, e)] + [(b"", b"")] * (world - 1)
            assert o.decode().strip().count("\n") == 0, "the launcher's stdout is the JSON line and nothing else"
        else:
            procs = []
            for r in range(world):
                env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **hooks)
                procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=root))
            outs = [p.communicate(timeout=600) for p in procs]
            for r, (p, (o, e)) in enumerate(zip(procs, outs)):
                assert p.returncode == 0, "rank %d:\n%s" % (r, e.decode()[-3000:])
        line = json.loads(outs[0][0].decode().strip().splitlines()[-1])
        # what a SCALE record needs: the world the line claims, the world the communicator saw, every rank's kernel and gather time
        assert line["n_gpus"] == world and line["gather"]["rccl_world"] == world and len(line["per_rank"]) == world
        assert [r["rank"] for r in line["per_rank"]] == list(range(world))
        assert line["gather"]["bytes_received_per_rank_per_step"] == world * nals * 32
        assert line["gather"]["gather_ms"] > 0 and all(r["kernel_ms"] > 0 and r["gather_ms"] > 0 for r in line["per_rank"])
        assert line["gather"]["gather_ms_max_over_ranks"] >= max(r["gather_ms"] for r in line["per_rank"]) - 1e-3
        assert line["scaling"] == "weak" and line["steps"] == 4 and line["ms_per_step"] > 0
        assert line["value"] > 0 and line["config"]["nals_per_gpu"] == nals and line["config"]["device_exclusive"] == 0
        assert outs[1][0].decode().strip() == "" or "metric" not in outs[1][0].decode()      # one JSON line, from rank 0
    finally:
        shm.close()
        shm.unlink()
