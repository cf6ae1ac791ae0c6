"""hbs_pair_alloc / hbs_pair_free: output buffers placed against their input (include/hevcbitstream_amd.h).  What is tested
here is function -- the memory is ordinary device memory, results through it are the oracle's, the report adds up, freeing
works; the speed it buys is measured by scripts/pair_time.py and bench.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_small_buffers_are_plain_memory():
    import torch
    import hevcbitstream_amd as hbs
    from tests import _orc
    orc = _orc.oracle()
    c = hbs.Context(0)
    try:
        stream, want_idx, want_arena = orc.gen_stream(0x99, 300, 0)
        d = torch.from_numpy(stream).cuda()
        index, rbsp, summary, cap = c.alloc_outputs(d.numel(), peer=d)
        rep = c.last_pair_report
        assert rep["chunks"] == 1 and rep["probed"] == 0 and rep["rejected"] == 0       # far below the probing size
        assert rbsp.is_cuda and rbsp.dtype == torch.uint8 and rbsp.numel() == d.numel() + 16 and rbsp.data_ptr() % 256 == 0
        c.index_extract_async(d, index, cap, rbsp, summary)
        s = c.read_summary(summary)
        n = int(s["nal_count"])
        assert n == len(want_idx)
        assert np.array_equal(index[: n * 32].cpu().numpy().view(hbs.NAL_ENTRY), want_idx)
        assert np.array_equal(rbsp[: int(s["rbsp_bytes"])].cpu().numpy(), want_arena)
        # torch ops work on it like on any tensor; dropping the last reference gives the memory back
        rbsp.fill_(7)
        assert int(rbsp.sum().item()) == 7 * rbsp.numel()
        view = rbsp[100:200]
        del rbsp
        assert int(view.sum().item()) == 700                 # a view keeps the memory alive
        del view
        torch.cuda.synchronize()
        assert c.lib.hbs_pair_free(None, None) == 0 and c.lib.hbs_pair_free(None, 12345) == -3
    finally:
        c.close()


def test_probed_arena_gives_the_same_bytes_and_the_report_adds_up():
    """2.2 GiB stream: two whole GiB, placed by measurement, and a remainder.  The arena through the paired buffer equals the
    arena through a torch buffer byte for byte (and both equal the generator's)."""
    import torch
    import hevcbitstream_amd as hbs
    c = hbs.Context(0)
    try:
        g = c.synth_stream(0x77, 230_000)
        sb, rb, n = g["stream_bytes"], g["rbsp_bytes"], 230_000
        assert sb > (2 << 30)
        stream = g["stream"][:sb]
        index, rbsp, summary, cap = c.alloc_outputs(sb, index_cap=n + 8, peer=stream)
        rep = c.last_pair_report
        # two whole GiB (measured) and a remainder
        assert rep["chunks"] == 3 and rep["probed"] >= 2 and rep["accepted_fast"] + rep["unprobed_after_budget"] == 3
        assert rep["rejected"] <= 90 and rep["accepted_fast"] <= 3
        c.index_extract_async(stream, index, cap, rbsp, summary)
        s = c.read_summary(summary)
        assert int(s["error"]) == 0 and int(s["nal_count"]) == n and int(s["rbsp_bytes"]) == rb
        assert torch.equal(rbsp[:rb], g["rbsp"][:rb])
        # the other direction: the re-emitted stream into a buffer placed against the arena
        out, rep2 = c.pair_alloc(rbsp, sb + 4096)
        idx_out = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
        c.emit_annexb_async(rbsp, rb, index, n, 1, out, idx_out, summary)
        s = c.read_summary(summary)
        assert int(s["error"]) == 0 and int(s["stream_bytes"]) == sb
        assert torch.equal(out[:sb], stream)
        # host copies work (ordinary device memory)
        assert np.array_equal(out[:4096].cpu().numpy(), stream[:4096].cpu().numpy())
    finally:
        c.close()


def test_many_allocations_in_one_process_stay_correct():
    """several paired buffers allocated, used and freed in one process: every one holds what was written into it (the first
    version, built from remapped virtual-memory chunks, served stale physical memory in the second buffer of a process)"""
    import torch
    import hevcbitstream_amd as hbs
    c = hbs.Context(0)
    try:
        g = c.synth_stream(0x31, 120_000)
        sb, rb, n = g["stream_bytes"], g["rbsp_bytes"], 120_000
        stream = g["stream"][:sb]
        keep = []
        for round_ in range(4):
            index, rbsp, summary, cap = c.alloc_outputs(sb, index_cap=n + 8, peer=stream)
            c.index_extract_async(stream, index, cap, rbsp, summary)
            s = c.read_summary(summary)
            assert int(s["error"]) == 0 and int(s["rbsp_bytes"]) == rb
            assert torch.equal(rbsp[:rb], g["rbsp"][:rb]), round_
            out, _ = c.pair_alloc(rbsp, sb + 4096)
            idx_out = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
            c.emit_annexb_async(rbsp, rb, index, n, 1, out, idx_out, summary)
            assert int(c.read_summary(summary)["stream_bytes"]) == sb
            assert torch.equal(out[:sb], stream), round_
            if round_ % 2 == 0:
                keep.append((rbsp, out))          # some stay alive, some are freed: the allocator sees both
        for rbsp, out in keep:
            assert torch.equal(rbsp[:rb], g["rbsp"][:rb]) and torch.equal(out[:sb], stream)
    finally:
        c.close()


def test_the_pool_keeps_what_it_learned():
    """Round 5: chunks are classed once.  A first paired buffer of 3 GiB pays for its chunks' probes; after it is freed, a second
    one against the same peer takes its chunks off the pool's free list -- no new chunk, only the peer's pieces are probed --
    and a buffer paired against a POOL buffer needs no probe at all (its peer's classes are on record).  The pool's books add
    up, trimming gives memory back, and everything written through these buffers is what was written."""
    import ctypes as C
    import time
    import torch
    import hevcbitstream_amd as hbs
    c = hbs.Context(0)
    try:
        lib = c.lib
        lib.hbs_pair_pool_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        lib.hbs_pair_pool_trim.argtypes = [C.c_void_p, C.c_uint64]
        lib.hbs_pair_pool_trim.restype = C.c_uint64

        def stats():
            out = (C.c_uint64 * 4)()
            assert lib.hbs_pair_pool_stats(c.h, out) == 0
            return [int(x) for x in out]

        peer = torch.randint(0, 256, ((3 << 30) + 4096,), dtype=torch.uint8, device="cuda")
        nbytes = (3 << 30) - 4096
        t0 = time.perf_counter()
        a, rep_a = c.pair_alloc(peer, nbytes)
        t_first = time.perf_counter() - t0
        assert rep_a["chunks"] == 3 and rep_a["accepted_fast"] + rep_a["unprobed_after_budget"] == 3
        s1 = stats()
        assert s1[0] >= 4 and s1[1] >= 3                           # the reference chunk + at least the buffer's three, classed
        a.fill_(3)
        pattern = torch.arange(nbytes, device="cuda", dtype=torch.int64).remainder_(251).to(torch.uint8)
        a.copy_(pattern)
        del a
        torch.cuda.synchronize()
        s2 = stats()
        assert s2[2] + s2[3] >= 3, s2                              # its chunks are on the free list now
        created_before = s2[0]
        t0 = time.perf_counter()
        b, rep_b = c.pair_alloc(peer, nbytes)
        t_second = time.perf_counter() - t0
        s3 = stats()
        assert rep_b["from_pool"] == 3 and s3[0] == created_before, (rep_b, s3)     # no new chunk: all three off the free list
        assert rep_b["probed"] == 3                                # the peer's three pieces, nothing else
        assert rep_b["accepted_fast"] == rep_a["accepted_fast"] or rep_b["accepted_fast"] == 3
        assert t_second < 0.25, (t_first, t_second)
        # the second buffer is fresh address space over recycled physical memory: it holds what is written into it
        b.copy_(pattern)
        assert torch.equal(b, pattern)
        # a buffer against a pool buffer: classes by lookup
        d, rep_d = c.pair_alloc(b, nbytes)
        assert rep_d["from_table"] == 3 and rep_d["probed"] <= 2 * (rep_d["chunks"] - rep_d["from_pool"]), rep_d
        d.copy_(b)
        assert torch.equal(d, pattern)
        del b, d, pattern
        torch.cuda.synchronize()
        free_before = sum(stats()[2:])
        assert free_before >= 6
        released = int(lib.hbs_pair_pool_trim(c.h, 2 << 30))
        after = stats()
        assert released == (free_before - 2) << 30 and after[2] + after[3] == 2, (released, after)
        assert int(lib.hbs_pair_pool_trim(c.h, 0)) == 2 << 30 and sum(stats()[2:]) == 0
        # an unaligned peer is an argument error, not a misaligned probe
        rc = lib.hbs_pair_alloc(c.h, C.c_void_p(peer.data_ptr() + 4), peer.numel() - 4, nbytes, C.byref(C.c_void_p()), None)
        assert rc == -3
    finally:
        c.close()


def test_free_from_another_thread_leaves_the_device_alone():
    """hbs_pair_free runs wherever the last reference dies; it must not change the caller's current device (one GPU here: the call
    must at least come back with the device it found and the memory given back)"""
    import threading
    import torch
    import hevcbitstream_amd as hbs
    c = hbs.Context(0)
    try:
        peer = torch.zeros((1 << 30) + 4096, dtype=torch.uint8, device="cuda")
        buf, rep = c.pair_alloc(peer, 1 << 30)
        assert rep["chunks"] == 1
        box = [buf]
        del buf

        def drop():
            box.clear()

        t = threading.Thread(target=drop)
        t.start()
        t.join()
        assert torch.cuda.current_device() == 0
        again, rep2 = c.pair_alloc(peer, 1 << 30)
        assert rep2["from_pool"] == 1
    finally:
        c.close()
