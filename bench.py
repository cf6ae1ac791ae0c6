#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X Annex-B indexer / RBSP extractor.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: as typed -- this process then starts one child per GPU and relays rank 0's line -- or under
            python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...)

Metric (BASELINE.json): Annex-B GB/s scanned (+ NAL units/s) on a 16 GiB synthetic
stream per GPU.  One "step" = one pass of the hot path -- start-code scan + NAL
index + RBSP extraction (hbs_index_extract, the fused K12 kernel plus its
end-of-stream fix-up) -- over a stream that is already resident in HBM; with
N > 1 each rank owns an independent 16 GiB shard (weak scaling) and the step
ends with the one real exchange of the path, the gather of the NAL index to
every rank through the C ABI (hbs_gather_index, RCCL), reported by itself as
"gather".  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline      the fused kernel against the 8 TB/s HBM3E peak: algorithmic bytes
                per launch (stream read + RBSP written + 32 B/NAL index) divided by
                its mean duration, measured here with HIP events recorded around
                that kernel on the stream it runs on (hbs_ctx_enable_timing).
  cpu_baseline  the reference's loop (find_nal_unit + nal_to_rbsp per NAL) timed on ONE host
                core over a bounded prefix of rank 0's stream: the real reference library when
                its prebuilt copy is there (oracle/_ref, "reference"), else the oracle's
                byte-at-a-time restatement (oracle/hbs_oracle_nal.c, "port").
  other_kernels (N = 1 only, outside the timed region) the other rows of the path on the same
                GPU: the scan alone (index only, no arena), RBSP -> Annex-B (hbs_emit_annexb) over
                the 16 GiB arena, header parse and header writers (hbs_parse_headers /
                hbs_write_headers) on a 100 k-NAL 4K30 stream.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
# hbs_ctx_last_kernel -> the kernel's name in a rocprofv3 trace
KERNEL_NAMES = {2: "hbs::k_scan_extract", 4: "hbs::k_scan_extract4", 6: "hbs::k_scan_extract4_r24",
                5: "hbs::k_index5_stream (+ k_index5_chunks, k_index5_prefix, k_index5_emit)"}
KERNEL_GEOMETRY = {2: "512 threads, 64 KiB tiles (LDS image)",
                   4: "256 threads, 192 KiB tiles held in registers (48 rows a wavefront), tiles handed out by ticket",
                   6: "256 threads, 96 KiB tiles held in registers (24 rows a wavefront), tiles handed out by ticket"}
PLACED = False                 # --placed-arena: outputs through hbs_pair_alloc (main() sets it; the helper lines below follow it)
N_NALS_16GIB = 1_677_000       # S(seed, n) with ~10 KiB NALs: 16.0 GiB of Annex-B
SEED = 0x1234


def cpu_baseline(stream_dev, index_dev, rbsp_dev, n_nals, sample_nals):
    """The reference's own loop (find_nal_unit + nal_to_rbsp per NAL) on one host core over the first `sample_nals`
    NALs of the stream: through the REAL reference library when its prebuilt copy travelled with the tree
    (oracle/_ref, kind "reference"), else through the oracle's restatement (kind "port").
    index_dev / rbsp_dev are what the GPU produced in the timed loop: EVERY entry (start, end, rbsp_off, rbsp_len) of the
    sample and EVERY byte of its RBSP are compared with what the reference produced here -- the headline run is pinned on the
    reference, not only on the device generator."""
    import ctypes as C
    import numpy as np
    from tests import _orc
    ent = index_dev[: sample_nals * 32].cpu().numpy().view(_orc.NAL_ENTRY)
    nbytes = int(ent["end"][-1])
    # keep the following start code: the last sampled NAL is then terminated exactly as in the full stream
    host = stream_dev[: nbytes + 4].cpu().numpy()
    arena = np.zeros(nbytes + 64, dtype=np.uint8)          # pre-faulted output
    arena[:] = 1
    u8p = C.POINTER(C.c_uint8)
    drv = os.path.join(ROOT, "oracle", "_ref", "libref_driver.so")
    if os.path.exists(drv):
        lib = C.CDLL(drv)
        lib.ref_walk.restype = C.c_int64
        lib.ref_walk.argtypes = [u8p, C.c_int64, u8p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_uint64), C.c_int64]
        lib.ref_walk_index.restype = C.c_int64
        lib.ref_walk_index.argtypes = [u8p, C.c_int64, u8p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        ref_entry = np.dtype([("start", "<u8"), ("end", "<u8"), ("rbsp_off", "<u8"), ("rbsp_len", "<i4"), ("rc_rbsp", "<i4"),
                              ("rc_find", "<i4"), ("pad", "<i4")])
        rent = np.zeros(sample_nals + 8, dtype=ref_entry)   # pre-faulted like the arena
        tot = C.c_int64(0)
        t0 = time.perf_counter()
        # the walk that also writes down what it learns per NAL (40 bytes per ~10 KiB NAL: the same loop, the same time)
        n = lib.ref_walk_index(host.ctypes.data_as(u8p), len(host), arena.ctypes.data_as(u8p), len(arena), rent.ctypes.data, len(rent), C.byref(tot))
        dt = time.perf_counter() - t0
        # the 4 bytes kept behind the sample are a start code: the reference finds one more, empty-handed NAL there or stops
        assert n >= sample_nals
        r = rent[:sample_nals]
        for f in ("start", "end", "rbsp_off"):
            assert np.array_equal(r[f], ent[f]), "GPU index field %s differs from the reference's" % f
        assert np.array_equal(r["rbsp_len"].astype(np.int64), ent["rbsp_len"].astype(np.int64)), "GPU rbsp_len differs from the reference's"
        assert int((r["rc_rbsp"] < 0).sum()) == 0
        ref_rb = int(r["rbsp_off"][-1]) + int(r["rbsp_len"][-1])
        assert tot.value >= ref_rb
        kind, what = "reference", "the reference's find_nal_unit + nal_to_rbsp (oracle/_ref/libhevcref.so, gcc -O2) driven by oracle/ref_driver.c"
    else:
        orc = _orc.oracle()
        idx = np.zeros(sample_nals + 8, dtype=_orc.NAL_ENTRY)
        why = C.c_int(0)
        t0 = time.perf_counter()
        n = orc.lib.orc_index_stream(host.ctypes.data_as(u8p), len(host), idx.ctypes.data, len(idx), C.byref(why))
        tot = orc.lib.orc_extract_rbsp(host.ctypes.data_as(u8p), idx.ctypes.data, n, arena.ctypes.data_as(u8p), len(arena))
        dt = time.perf_counter() - t0
        assert n >= sample_nals and tot > 0
        for f in ("start", "end", "rbsp_off", "rbsp_len"):
            assert np.array_equal(idx[f][:sample_nals], ent[f]), "GPU index field %s differs from the oracle's" % f
        ref_rb = int(idx["rbsp_off"][sample_nals - 1]) + int(idx["rbsp_len"][sample_nals - 1])
        kind, what = "port", "find_nal_unit loop + nal_to_rbsp per NAL, oracle/hbs_oracle_nal.c, gcc -O2"
    # every RBSP byte of the sample: the GPU's arena against the CPU's, a piece at a time
    t1 = time.perf_counter()
    step = 1 << 29
    for lo in range(0, ref_rb, step):
        hi = min(ref_rb, lo + step)
        piece = rbsp_dev[lo:hi].cpu().numpy()
        if not np.array_equal(piece, arena[lo:hi]):
            bad = lo + int(np.flatnonzero(piece != arena[lo:hi])[0])
            raise AssertionError("GPU RBSP arena differs from the %s's at byte %d of %d" % (kind, bad, ref_rb))
    t_cmp = time.perf_counter() - t1
    checked = ("the GPU's index (start, end, rbsp_off, rbsp_len of all %d NALs) and all %d RBSP bytes of the sample compared "
               "with this walk's: equal (%.1f s)" % (sample_nals, ref_rb, t_cmp))
    cpu_model = "?"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    res = {"value": round(len(host) / dt / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": kind, "cpu": cpu_model,
           "nal_per_s": round(n / dt, 1),
           "sample": "first %d NALs (%.2f GiB) of rank 0's stream: %s, 1 thread, %.1f s; %s" % (sample_nals, len(host) / 2**30, what, dt, checked),
           "parity_checked_nals": sample_nals, "parity_checked_rbsp_bytes": ref_rb}
    # the same loop on every host core at once, one contiguous share of the sample per thread (the reference itself is
    # single-threaded; this is what a caller could get out of the host by sharding the file)
    import threading
    cores = os.cpu_count() or 1
    if cores > 1 and sample_nals >= 64 * cores:
        cuts = [int(ent["start"][sample_nals * t // cores]) - 4 if t else 0 for t in range(cores)] + [len(host)]
        done = [0] * cores

        def work(t):
            lo, hi = cuts[t], cuts[t + 1] + (4 if t + 1 < cores else 0)      # up to and including the next share's start code
            sub, out = host[lo:hi], arena[lo:hi]
            if kind == "reference":
                tt = C.c_int64(0)
                done[t] = lib.ref_walk(sub.ctypes.data_as(u8p), len(sub), out.ctypes.data_as(u8p), len(out), C.byref(tt), None, 0)
            else:
                ix = np.zeros(sample_nals // cores + 64, dtype=_orc.NAL_ENTRY)
                w = C.c_int(0)
                m = orc.lib.orc_index_stream(sub.ctypes.data_as(u8p), len(sub), ix.ctypes.data, len(ix), C.byref(w))
                orc.lib.orc_extract_rbsp(sub.ctypes.data_as(u8p), ix.ctypes.data, min(m, len(ix)), out.ctypes.data_as(u8p), len(out))
                done[t] = m
        th = [threading.Thread(target=work, args=(t,)) for t in range(cores)]
        t0 = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        dtm = time.perf_counter() - t0
        assert sum(done) >= sample_nals
        res["all_cores"] = {"value": round(len(host) / dtm / 1e9, 3), "unit": "GB/s", "cores": cores,
                            "note": "%d threads, one contiguous share of the same sample each, %.2f s" % (cores, dtm)}
    return res


def cpu_baseline_parse(stream_dev, index_dev, rbsp_dev, m, parsed, structs_dev, compact=None):
    """The CPU-baseline leg of BASELINE config 3: the reference's read_hevc_nal_unit (hevc_stream.c:155-240) fed the m NALs of
    the 4K30 sequence in stream order on one host core -- the compiled reference when its prebuilt library travelled with the
    tree, else the oracle's restatement -- timed, and EVERY NAL's rc, NAL header, struct (every member) and slice payload
    compared with what the GPU parse produced (tests/_parsecmp.compare)."""
    import numpy as np
    from tests import _orc
    from tests._parsecmp import compare, oracle_pass
    s = stream_dev.cpu().numpy()
    idx = index_dev[: m * 32].cpu().numpy().view(_orc.NAL_ENTRY)
    nals = [bytes(s[int(a):int(b)]) for a, b in zip(idx["start"], idx["end"])]
    tm = []
    if _orc.reference() is not None:
        exp = oracle_pass(nals, parser=_orc.ReferenceHevc(), timing=tm)
        kind, what = "reference", "the reference's read_hevc_nal_unit (oracle/_ref/libhevcref.so, gcc -O2), one call per NAL through ctypes"
    else:
        exp = oracle_pass(nals, timing=tm)
        kind, what = "port", "orc_read_hevc_nal_unit (oracle/hbs_oracle_parse.c, gcc -O2), one call per NAL through ctypes"
    arena = rbsp_dev[: int(idx["rbsp_off"][-1]) + int(idx["rbsp_len"][-1])].cpu().numpy()
    compare(parsed, structs_dev.cpu().numpy(), arena, idx, exp)
    n_compact = None
    if compact is not None:
        # the compact parse against the SAME walk, directly (round 5's verdict: it was compared with the full parse only): rc and the NAL
        # header of every NAL, the sixteen members of every slice's record against the struct the reference filled for that slice
        from hevcbitstream_amd.api import COMPACT, COMPACT_FIELDS, PARSED
        cp_h = compact[0].cpu().numpy().view(PARSED)
        cc_h = compact[1].cpu().numpy().view(COMPACT)
        where = {name: i for name, i, cnt in _orc.flat_fields("hevc_slice_header_t")}
        cols = np.array([where[f] for f in COMPACT_FIELDS])
        n_compact = 0
        for k, e in enumerate(exp):
            assert int(cp_h["rc"][k]) == e["rc"], "compact parse: rc of NAL %d differs from the %s's" % (k, kind)
            assert [int(cp_h["nal_unit_type"][k]), int(cp_h["nal_layer_id"][k]), int(cp_h["nal_temporal_id_plus1"][k])] == list(e["nal"])[1:], k
            if e.get("kind") == "sh" and e["rc"] >= 0:
                got = np.array([cc_h[f][k] for f in COMPACT_FIELDS])
                assert np.array_equal(got, e["struct"][cols]), "compact parse: slice record of NAL %d differs from the %s's struct" % (k, kind)
                assert int(cp_h["slice_data_size"][k]) == e["slice_data"][0], k
                n_compact += 1
    return {"value": round(m / tm[0], 1), "unit": "NAL/s", "cores": 1, "kind": kind, "compact_slices_checked": n_compact,
            "sample": "all %d NALs of the sequence: %s, %.2f s inside the calls; the GPU parse's rc, NAL header, every struct member and "
                      "every slice payload compared with this walk's: equal" % (m, what, tm[0]),
            "parity_checked_nals": m}


def cpu_baseline_1gib(stream_dev, sb, n, index_dev, rbsp_dev, rb, index_only_dev, emitted_dev, emitted_index_dev):
    """BASELINE configs 2 and 4 pinned on the reference at the size they name: the WHOLE 1 GiB stream through the reference's own
    loop on one host core (find_nal_unit + nal_to_rbsp per NAL, hevc_analyze.c:135-177 / h264_nal.c:38-200; timed), every entry
    of the GPU's index (with and without an arena) and every byte of its RBSP arena compared with what that walk produced; then
    the reference's rbsp_to_nal (h264_nal.c:92-132; timed) over the reference's arena, every byte and every output entry of the
    stream the GPU re-emitted compared with it -- and with the input bytes (the round trip config 4 asks for)."""
    import numpy as np
    from tests._refwalk import device_equals_host, reference_emit, reference_walk
    from tests import _orc
    host = stream_dev[:sb].cpu().numpy()
    t0 = time.perf_counter()
    ent, ref_arena, ref_rb, kind = reference_walk(host, n + 16)
    t_walk = time.perf_counter() - t0
    assert len(ent) == n and ref_rb == rb, (len(ent), n, ref_rb, rb)
    got = index_dev[: n * 32].cpu().numpy().view(_orc.NAL_ENTRY)
    got5 = index_only_dev[: n * 32].cpu().numpy().view(_orc.NAL_ENTRY)
    for f in ("start", "end", "rbsp_off"):
        assert np.array_equal(got[f], ent[f]), "config 2: index field %s differs from the %s's" % (f, kind)
    for f in ("start", "end"):
        assert np.array_equal(got5[f], ent[f]), "config 2, index only: field %s differs from the %s's" % (f, kind)
    assert np.array_equal(got["rbsp_len"].astype(np.int64), ent["rbsp_len"].astype(np.int64)), "config 2: rbsp_len"
    assert int((ent["rc_rbsp"] < 0).sum()) == 0
    device_equals_host(rbsp_dev, ref_arena, ref_rb, "config 2: RBSP arena")
    t0 = time.perf_counter()
    ref_stream, ref_sb = reference_emit(ref_arena, ent, sb + (1 << 16))
    t_emit = time.perf_counter() - t0
    assert ref_sb == sb and np.array_equal(ref_stream[:sb], host), "the reference's own round trip"
    device_equals_host(emitted_dev, ref_stream, sb, "config 4: re-emitted stream")
    eo = emitted_index_dev[: n * 32].cpu().numpy().view(_orc.NAL_ENTRY)
    for f in ("start", "end"):
        assert np.array_equal(eo[f], ent[f]), "config 4: output index field %s" % f
    what = {"reference": "oracle/_ref/libhevcref.so (the reference compiled here, gcc -O2) driven by oracle/ref_driver.c",
            "port": "oracle/hbs_oracle_nal.c (the restatement pinned to it), gcc -O2"}[kind]
    return {"config2": {"value": round(sb / t_walk / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": kind,
                        "sample": "the whole stream, %d NALs / %.3f GiB: find_nal_unit + nal_to_rbsp per NAL, %s, 1 thread, %.2f s; every entry of the GPU's index "
                                  "(start, end, rbsp_off, rbsp_len; start / end of the index-only call too) and all %d bytes of its RBSP arena compared with this walk's: equal"
                                  % (n, sb / 2**30, what, t_walk, ref_rb)},
            "config4": {"value": round(sb / t_emit / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": kind,
                        "sample": "rbsp_to_nal per NAL over the whole arena, %s, 1 thread, %.2f s; all %d bytes the GPU emitted and its output index compared with this: equal, "
                                  "and equal to the input stream (bit-exact round trip)" % (what, t_emit, sb)}}


def configs_1gib(torch, hbs, ctx, check=True, reps=12):
    """BASELINE.json configs[1] and configs[3] at the size they name: S(0x1234, 104 858 NALs) = 1.0003 GiB of ~10 KiB NALs,
    resident in HBM.  Config 2: start-code scan + NAL index + RBSP extraction (hbs_index_extract), and the scan alone (no arena);
    config 4: the arena re-emitted as Annex-B (hbs_emit_annexb).  Per call: kernel_ms = the library's HIP events around the
    dominant kernel(s) on their own stream (what the 16 GiB headline's roofline uses), call_ms = the whole call, every launch of
    it, from events around calls issued back to back; fractions of the 8 TB/s peak from the algorithmic bytes.  With `check`
    (the default bench run) the WHOLE stream goes through the reference on the host and everything is compared (cpu_baseline_1gib)."""
    n = 104_858
    g = ctx.synth_stream(SEED, n, 0)
    sb, rb = g["stream_bytes"], g["rbsp_bytes"]
    stream = g["stream"][:sb]
    index, rbsp, summary, cap = ctx.alloc_outputs(sb, index_cap=n + 8, peer=stream if PLACED else None)
    place_arena = dict(ctx.last_pair_report) if PLACED else None

    def timed(fn, with_kernel_ms):
        ks = []
        fn()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        for i in range(reps + 1):
            ev[i].record()
            if i < reps:
                fn()
                if with_kernel_ms:
                    ks.append(ctx.kernel_ms())           # (waits for that call: the next one is issued behind an idle stream -- the call_ms
        torch.cuda.synchronize()                          #  loop below runs without it)
        if with_kernel_ms:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
            for i in range(reps + 1):
                ev[i].record()
                if i < reps:
                    fn()
            torch.cuda.synchronize()
        calls = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
        ks.sort()
        return (ks[len(ks) // 2] if ks else None), calls[len(calls) // 2], calls[0]

    res = {"workload": "S(seed=0x1234, n_nals=%d, uniform): %.4f GiB Annex-B, ~10 KiB NALs, resident in HBM (BASELINE.json configs[1] and configs[3])" % (n, sb / 2**30),
           "stream_bytes": sb, "nals": n, "rbsp_bytes": rb}
    # config 2, with the arena
    k_ms, c_med, c_min = timed(lambda: ctx.index_extract_async(stream, index, cap, rbsp, summary), True)
    s = ctx.read_summary(summary)
    assert int(s["error"]) == 0 and int(s["nal_count"]) == n and int(s["rbsp_bytes"]) == rb, s
    algo = sb + rb + 32 * n
    kern = KERNEL_NAMES.get(ctx.last_kernel(), "?")
    res["config2_extract"] = {"value": round(sb / k_ms / 1e6, 1), "unit": "GB/s scanned", "kernel": kern, "kernel_ms": round(k_ms, 4), "call_ms": round(c_med, 4),
                              "call_ms_min": round(c_min, 4), "algorithmic_bytes": algo,
                              "roofline": {"bound": "hbm", "achieved": round(algo / k_ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": round(algo / k_ms / 1e6 / HBM_PEAK_GBS, 4), "frac_of_call": round(algo / c_med / 1e6 / HBM_PEAK_GBS, 4)},
                              "arena_placement": place_arena}
    # the same call with every tile by ticket (hbs_ctx_set_device_exclusive(0), the library's default: what a caller gets who
    # shares the device with other contexts)
    if getattr(ctx, "exclusive", 0):
        ctx.set_device_exclusive(0)
        kt_ms, ct_med, _ = timed(lambda: ctx.index_extract_async(stream, index, cap, rbsp, summary), True)
        ctx.set_device_exclusive(1)
        res["config2_extract"]["all_tiles_by_ticket"] = {"kernel_ms": round(kt_ms, 4), "call_ms": round(ct_med, 4)}
    # config 2, the scan alone
    index5 = torch.empty_like(index)
    summary5 = torch.zeros_like(summary)
    k_ms, c_med, c_min = timed(lambda: ctx.index_extract_async(stream, index5, cap, None, summary5), True)
    s5 = ctx.read_summary(summary5)
    assert int(s5["error"]) == 0 and int(s5["nal_count"]) == n
    algo5 = sb + 32 * n
    kern5 = KERNEL_NAMES.get(ctx.last_kernel(), "?")
    res["config2_index_only"] = {"value": round(sb / k_ms / 1e6, 1), "unit": "GB/s scanned", "kernel": kern5, "kernel_ms": round(k_ms, 4), "call_ms": round(c_med, 4),
                                 "call_ms_min": round(c_min, 4), "algorithmic_bytes": algo5,
                                 "roofline": {"bound": "hbm", "achieved": round(algo5 / k_ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                              "frac": round(algo5 / k_ms / 1e6 / HBM_PEAK_GBS, 4), "frac_of_call": round(algo5 / c_med / 1e6 / HBM_PEAK_GBS, 4)}}
    # config 4: the arena the scan extracted, re-emitted (default path)
    torch.cuda.empty_cache()
    out, place_out = ctx.pair_alloc(rbsp, sb + 4096) if PLACED else (torch.empty(sb + 4096, dtype=torch.uint8, device="cuda"), None)
    idx_out = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
    esum = torch.zeros(64, dtype=torch.uint8, device="cuda")
    _, c_med, c_min = timed(lambda: ctx.emit_annexb_async(rbsp, rb, index, n, 1, out, idx_out, esum), False)
    se = ctx.read_summary(esum)
    assert int(se["error"]) == 0 and int(se["stream_bytes"]) == sb, se
    by_tiles = int(ctx.lib.hbs_ctx_last_emit_by_tiles(ctx.h))
    assert torch.equal(out[:sb], stream), "config 4: re-emitted stream != input bytes"
    algo4 = rb + sb + 64 * n                                  # arena read, stream written, an index entry read and one written per NAL
    res["config4_emit"] = {"value": round(sb / c_med / 1e6, 1), "unit": "GB/s emitted", "kernel": "hbs::k3_tiles<0>" if by_tiles else "hbs::k3_fused / k3_emit",
                           "call_ms": round(c_med, 4), "call_ms_min": round(c_min, 4), "algorithmic_bytes": algo4,
                           "roofline": {"bound": "hbm", "achieved": round(algo4 / c_med / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": round(algo4 / c_med / 1e6 / HBM_PEAK_GBS, 4),
                                        "note": "of the WHOLE call (every launch of it): the library records no events inside hbs_emit_annexb"},
                           "output_placement": place_out, "round_trip": "emitted bytes == input stream (compared on the device)"}
    if check:
        cb = cpu_baseline_1gib(stream, sb, n, index, rbsp, rb, index5, out, idx_out)
        res["config2_extract"]["cpu_baseline"] = cb["config2"]
        res["config2_index_only"]["parity"] = "start / end of every entry equal to the same walk's"
        res["config4_emit"]["cpu_baseline"] = cb["config4"]
    del out, idx_out, index5, index, rbsp
    torch.cuda.empty_cache()
    return res


def cpu_baseline_compact(parsed, structs_dev, compact_parsed_dev, compact_dev):
    """checking leg of parse_headers_compact: every per-NAL record and the sixteen members of every slice's compact record against
    the structs of the full parse (which cpu_baseline_parse compares with the reference); member offsets from the layout the
    ABI test pins (tests/golden/field_layout.json through tests/_orc.py).  Returns the number of slices compared."""
    import numpy as np
    from tests import _orc
    from hevcbitstream_amd.api import COMPACT, COMPACT_FIELDS, PARSED
    cp_h = compact_parsed_dev.cpu().numpy().view(PARSED)
    cc_h = compact_dev.cpu().numpy().view(COMPACT)
    st_h = structs_dev.cpu().numpy()
    for f in ("rc", "nal_unit_type", "nal_layer_id", "nal_temporal_id_plus1", "slice_data_size", "slice_data_off"):
        assert np.array_equal(cp_h[f], parsed[f]), "compact parse: record field %s differs from the full parse's" % f
    t = parsed["nal_unit_type"]
    sl = np.flatnonzero((((t >= 0) & (t <= 9)) | ((t >= 16) & (t <= 21))) & (parsed["struct_off"] != np.uint64(0xFFFFFFFFFFFFFFFF)))
    offs = parsed["struct_off"][sl].astype(np.int64)
    words = st_h[: (len(st_h) // 4) * 4].view(np.int32)
    where = {name: i for name, i, cnt in _orc.flat_fields("hevc_slice_header_t")}
    for f in COMPACT_FIELDS:
        assert np.array_equal(cc_h[f][sl], words[offs // 4 + where[f]]), "compact parse: member %s differs from the full struct's" % f
    return int(len(sl))


def pmc_traffic(kernel_name, algo_bytes):
    """profiles/r*/traffic_<kernel>.json of the newest round -- if it is about this kernel, this workload AND this
    source: the file records the digest of hevcbitstream_amd/csrc at profiling time; after any change to the kernels
    the figure is stale and the bench line says traffic: null until the PMC passes are run again."""
    import glob
    import hevcbitstream_amd as hbs
    short = kernel_name.split("::")[-1]
    digest = hbs.source_digest()
    stale = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic_%s*.json" % short)), reverse=True):      # ..._zero_heavy.json: the same kernel on the stress workload
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("algorithmic_bytes_per_launch") != algo_bytes:
            continue
        if d.get("source_sha256") != digest:
            stale = stale or os.path.relpath(f, ROOT)
            continue
        return {"traffic_bytes_per_launch": d["traffic_bytes_per_launch"],
                "source": "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x2 per the gfx950 note; "
                          "kernel sources unchanged since: sha256 %s): fetch %d + write %d B per launch"
                          % (os.path.relpath(f, ROOT), digest[:12], d["fetch_bytes_per_launch"], d["write_bytes_per_launch"])}
    if stale:
        return {"traffic_bytes_per_launch": None, "source": "%s was measured on other kernel sources (digest now %s): stale, not quoted" % (stale, digest[:12])}
    return None


def config3_end_to_end(torch, hbs, ctx, d_small, index_s, rbsp_s, m, parsed_s, structs_s, cpu_parse=False):
    """BASELINE config 3 as ONE pipeline at the size of a real 4K30 stream: the ~100 k NALs of the synthetic 4K30 sequence with
    slice payloads of 16-28 KiB (about 2.2 GiB; built on the device: every slice's RBSP is its original bytes followed by random
    payload, last byte 80, then K3), timed as start-code scan + index + RBSP extraction followed by the header parse, back to back
    on one stream.  The structs must come out as they do for the small stream (same headers)."""
    import numpy as np
    from hevcbitstream_amd.api import NAL_ENTRY, PARSED, SUMMARY
    ent = index_s[: m * 32].cpu().numpy().view(NAL_ENTRY)
    old_off = ent["rbsp_off"].astype(np.int64)
    old_len = ent["rbsp_len"].astype(np.int64)
    is_slice = parsed_s["nal_unit_type"] < 32
    rng = np.random.RandomState(3)
    new_len = old_len + np.where(is_slice, rng.randint(16 << 10, 28 << 10, size=m), 0)
    new_off = np.concatenate([[0], np.cumsum(new_len)[:-1]])
    total = int(new_len.sum())
    dev = d_small.device
    big = torch.randint(0, 256, (total + 64,), dtype=torch.uint8, device=dev)
    shift = torch.from_numpy(new_off - old_off).to(dev)
    old_total = int(old_off[-1] + old_len[-1])
    pos = torch.repeat_interleave(shift, torch.from_numpy(old_len).to(dev)) + torch.arange(old_total, device=dev)
    big[pos] = rbsp_s[:old_total]
    big[torch.from_numpy(new_off + new_len - 1).to(dev)[torch.from_numpy(is_slice).to(dev)]] = 0x80
    ent2 = np.zeros(m, dtype=NAL_ENTRY)
    ent2["rbsp_off"] = new_off
    ent2["rbsp_len"] = new_len
    idx2 = torch.from_numpy(ent2.view(np.uint8).copy()).to(dev)
    cap_out = int(ctx.lib.hbs_annexb_bound(total, m))
    stream2 = torch.empty(cap_out, dtype=torch.uint8, device=dev)
    summ = torch.zeros(SUMMARY.itemsize, dtype=torch.uint8, device=dev)
    ctx.emit_annexb_async(big, total, idx2, m, 1, stream2, None, summ)
    s = ctx.read_summary(summ)
    assert int(s["error"]) == 0
    sb2 = int(s["stream_bytes"])
    del pos, shift
    index2, rbsp2, summ2, cap2 = ctx.alloc_outputs(sb2, index_cap=m + 8)
    parsed2 = torch.empty(m * PARSED.itemsize, dtype=torch.uint8, device=dev)
    structs2 = torch.empty_like(structs_s)
    psum = torch.zeros(SUMMARY.itemsize, dtype=torch.uint8, device=dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    for i in range(5):
        ev[i].record()
        if i < 4:
            ctx.index_extract_async(stream2[:sb2], index2, cap2, rbsp2, summ2)
            ctx.parse_headers_async(rbsp2, index2, m, parsed2, structs2, psum)
    torch.cuda.synchronize()
    ms = min(ev[i].elapsed_time(ev[i + 1]) for i in range(1, 4))
    s2 = ctx.read_summary(summ2)
    assert int(s2["error"]) == 0 and int(s2["nal_count"]) == m and int(s2["rbsp_bytes"]) == total
    assert torch.equal(rbsp2[:total], big[:total]), "extracted RBSP != the RBSP the stream was made from"
    p2 = parsed2.cpu().numpy().view(PARSED)
    # rc = bytes of the NAL consumed (grows with the payload) or -1
    assert np.array_equal(p2["rc"] < 0, parsed_s["rc"] < 0) and np.array_equal(p2["nal_unit_type"], parsed_s["nal_unit_type"])
    used = structs_s.numel() - 16                             # (the arena is allocated 16 bytes longer than the parse fills)
    assert torch.equal(structs2[:used], structs_s[:used]), "header structs differ from those of the same headers in the small stream"
    full_size_parity = None
    if cpu_parse:
        # round 5's verdict, thin spot (ii): this run was pinned on the reference through the short-payload run only.  The reference's
        # read_hevc_nal_unit over the 2.1 GiB stream itself, NAL by NAL (it strips every whole NAL: ~2.2 GB through nal_to_rbsp), and
        # rc, NAL header, every struct member and every slice payload of THIS run's outputs compared with it
        full_size_parity = cpu_baseline_parse(stream2[:sb2], index2, rbsp2, m, p2, structs2)
    res = {"value": round(sb2 / ms / 1e6, 1), "unit": "GB/s of stream, scan + index + extraction + header parse", "ms": round(ms, 3),
           "nal_per_s": round(m / ms * 1e3, 1), "stream_bytes": sb2, "nals": m,
           "workload": "synthetic 4K30 sequence, %d NALs, slice payloads 16-28 KiB (%.2f GiB): hbs_index_extract then hbs_parse_headers, "
                       "back to back on one stream; structs equal to those of the same headers with short payloads" % (m, sb2 / 2**30)}
    # the same answer without the arena (hbs_index_parse: index-only scan, then the parse on 512-byte windows stripped from the
    # stream): what config 3 asks for -- "NAL index + VPS/SPS/PPS/slice_segment_header parse" -- costs 1 B/B, not 2
    index3 = torch.empty_like(index2)
    parsed3 = torch.empty_like(parsed2)
    structs3 = torch.empty_like(structs_s)
    pay3 = torch.empty(m, dtype=torch.int64, device=dev)
    ssum, psum3 = torch.zeros(SUMMARY.itemsize, dtype=torch.uint8, device=dev), torch.zeros(SUMMARY.itemsize, dtype=torch.uint8, device=dev)
    ts = []
    for i in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got = ctx.index_parse_async(stream2[:sb2], index3, cap2, parsed3, structs3, ssum, psum3, payload_off=pay3)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ms3 = min(ts[1:])
    assert got == m and int(ctx.read_summary(psum3)["error"]) == 0
    assert torch.equal(index3[: m * 32], index2[: m * 32]) and torch.equal(parsed3, parsed2), "hbs_index_parse: index / records differ from the arena path"
    assert torch.equal(structs3[:used], structs_s[:used]), "hbs_index_parse: header structs differ"
    first = stream2[pay3.clamp(min=0, max=sb2 - 1)]                       # the byte at every slice's payload offset ...
    sl = torch.from_numpy(is_slice).to(dev)
    ent2d = index2[: m * 32].view(torch.int64).view(m, 4)
    want = rbsp2[(ent2d[:, 2] + torch.from_numpy(p2["slice_data_off"].astype(np.int64)).to(dev)).clamp(max=total - 1)]
    assert torch.equal(first[sl], want[sl]), "payload offsets"          # ... is the RBSP byte the arena path's slice_data_off names
    # ... and with the compact parse behind the scan (hbs_index_parse_compact): no slice structs at all
    from hevcbitstream_amd.api import COMPACT
    cc3 = torch.empty(m * COMPACT.itemsize, dtype=torch.uint8, device=dev)
    parsed4 = torch.empty_like(parsed2)
    index4 = torch.empty_like(index2)
    ts = []
    for i in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got = ctx.index_parse_compact_async(stream2[:sb2], index4, cap2, parsed4, cc3, structs3, ssum, psum3)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ms4 = min(ts[1:])
    assert got == m and int(ctx.read_summary(psum3)["error"]) == 0
    p4 = parsed4.cpu().numpy().view(PARSED)
    for f in ("rc", "nal_unit_type", "slice_data_size", "slice_data_off"):
        assert np.array_equal(p4[f], p2[f]), "hbs_index_parse_compact: record field %s" % f
    assert torch.equal(index4[: m * 32], index2[: m * 32])
    res["without_arena_compact"] = {"value": round(sb2 / ms4 / 1e6, 1), "unit": "GB/s of stream, scan + index + compact header parse (hbs_index_parse_compact), host wall time of the call",
                                    "ms": round(ms4, 3), "nal_per_s": round(m / ms4 * 1e3, 1),
                                    "note": "index and per-NAL records equal to the arena path's; the slice records are those of parse_headers_compact (compared member by member there)"}
    if full_size_parity is not None:
        res["cpu_baseline_at_full_size"] = full_size_parity
    res["without_arena"] = {"value": round(sb2 / ms3 / 1e6, 1), "unit": "GB/s of stream, scan + index + header parse (hbs_index_parse), host wall time of the call",
                            "ms": round(ms3, 3), "nal_per_s": round(m / ms3 * 1e3, 1),
                            "note": "index, records and structs equal to the arena path's; includes the call's one wait (for the scan's NAL count)"}
    return res


def make_mixed(torch, stream, sb, percent=1.0, region_bytes=640 << 10):
    """A stream that is sparse on average with dense regions the density probe does not see: `percent` of the stream's bytes
    overwritten by 00 00 03 padding (cabac_zero_words-like runs, valid inside a NAL) in regions of `region_bytes`, placed
    midway BETWEEN the probe's 64 sample windows (k * (sb / 64)).  Returns (mixed stream, dense bytes)."""
    stride = (sb // 64) & ~15
    n_regions = max(1, int(sb * percent / 100.0 / region_bytes))
    pat = torch.tensor([0, 0, 3], dtype=torch.uint8, device=stream.device).repeat(region_bytes // 3 + 1)[:region_bytes]
    mixed = stream.clone()
    placed = 0
    for k in range(n_regions):
        off = (k % 64) * stride + stride // 2 + (k // 64) * (region_bytes + 4096)
        off = (off // 3) * 3
        if off + region_bytes + 16 >= sb:
            continue
        mixed[off: off + region_bytes] = pat
        mixed[off + region_bytes] = 0x80        # the byte behind the last 00 00 03: neither <= 3 nor part of a zero run
        placed += 1
    return mixed, placed * region_bytes


def mixed_stream_line(torch, ctx, stream, sb, n_cap, uniform_ms):
    """the automatic mode on the mixed stream against the uniform one; outputs checked against the LDS-image kernel (whose
    cost does not depend on the data) entry by entry and byte by byte on the device"""
    mixed, dense_bytes = make_mixed(torch, stream, sb)
    index, rbsp, summary, cap = ctx.alloc_outputs(sb, index_cap=n_cap, peer=mixed if PLACED else None)      # like the uniform run's arena
    ks = []
    for i in range(4):
        ctx.index_extract_async(mixed, index, cap, rbsp, summary)
        if i:
            ks.append(ctx.kernel_ms())
    s = ctx.read_summary(summary)
    kern = ctx.last_kernel()
    ks.sort()
    ms = ks[len(ks) // 2]
    # the same calls with the dense tiles counted in place (round 4's way; hbs_ctx_set_count_ahead), same process, same buffers
    ctx.set_count_ahead(0)
    ks0 = []
    for i in range(4):
        ctx.index_extract_async(mixed, index, cap, rbsp, summary)
        if i:
            ks0.append(ctx.kernel_ms())
    ctx.read_summary(summary)
    ks0.sort()
    ms0 = ks0[len(ks0) // 2]                                   # the median of three, as for the default mode above
    ctx.set_count_ahead(1)
    ctx.index_extract_async(mixed, index, cap, rbsp, summary)      # (what is compared below comes from the default mode)
    s = ctx.read_summary(summary)
    index2, rbsp2, summary2, _ = ctx.alloc_outputs(sb, index_cap=n_cap)
    ctx.set_kernel(2)
    ctx.index_extract_async(mixed, index2, cap, rbsp2, summary2)
    s2 = ctx.read_summary(summary2)
    ctx.set_kernel(0)
    m, rb = int(s2["nal_count"]), int(s2["rbsp_bytes"])
    assert int(s["error"]) == 0 and int(s["nal_count"]) == m and int(s["rbsp_bytes"]) == rb
    assert torch.equal(index[: m * 32], index2[: m * 32]) and torch.equal(rbsp[:rb], rbsp2[:rb]), "mixed stream: differs from the LDS-image kernel"
    return {"value": round(sb / ms / 1e6, 1), "unit": "GB/s scanned", "kernel_ms": round(ms, 4), "kernel": kern,
            "over_uniform": round(ms / uniform_ms, 3), "dense_bytes": dense_bytes, "nals": m,
            "without_count_ahead": {"kernel_ms": round(ms0, 4), "over_uniform": round(ms0 / uniform_ms, 3),
                                    "note": "median of three timed calls, like the default arm; kernel_ms covers the launches between the library's "
                                            "events (k_scan_ahead4 and the main kernel), not the sample that rides in the prologue (~1.7 us a GiB)"},
            "workload": "the bench stream with %.2f %% of its bytes overwritten by 00 00 03 padding in 640 KiB regions between the "
                        "density probe's windows (the probe says sparse; the dense tiles take the per-tile dense path)" % (100.0 * dense_bytes / sb)}


def mixed_index_only(torch, ctx, stream, sb, n_cap, uniform_ms):
    """the index-only scan on the same mixed stream (start / end of every NAL against the extract path's index)"""
    mixed, _ = make_mixed(torch, stream, sb)
    index, rbsp, summary, cap = ctx.alloc_outputs(sb, index_cap=n_cap)
    ctx.index_extract_async(mixed, index, cap, rbsp, summary)
    m = int(ctx.read_summary(summary)["nal_count"])
    del rbsp
    index2 = torch.empty_like(index)
    ks = []
    for i in range(4):
        ctx.index_extract_async(mixed, index2, cap, None, summary)
        if i:
            ks.append(ctx.kernel_ms())
    s = ctx.read_summary(summary)
    a = index[: m * 32].view(torch.int64).view(m, 4)
    b = index2[: m * 32].view(torch.int64).view(m, 4)
    assert int(s["error"]) == 0 and int(s["nal_count"]) == m and torch.equal(a[:, :3], b[:, :3]), "mixed stream, index only: differs from the extract path's index"
    ks.sort()
    ms = ks[len(ks) // 2]
    return {"value": round(sb / ms / 1e6, 1), "unit": "GB/s scanned", "kernel_ms": round(ms, 4), "over_uniform": round(ms / uniform_ms, 3)}


def emit_mixed_lines(torch, ctx, g, n, rb, sb, uniform_ms):
    """hbs_emit_annexb on a copy of the bench arena with 1 % of it overwritten in 640 KiB stretches placed between the density
    probe's 64 windows (scripts/emit_paths.py): `00 00 03` padding, then zeros.  The default path (arena tiles) timed; its bytes and
    output index compared with the kernel by NALs (path 0) on the padding arena."""
    region = 640 << 10
    stride = (rb // 64) & ~15
    out_cap = sb + sb // 40 + 4096
    res = {}
    for name, byte3 in (("padding_00_00_03", 3), ("zeros", 0)):
        arena = g["rbsp"][:rb].clone()
        pat = torch.tensor([0, 0, byte3], dtype=torch.uint8, device="cuda").repeat(region // 3 + 1)[:region]
        dense = 0
        for k in range(max(1, int(rb * 0.01 / region))):
            off = (k % 64) * stride + stride // 2 + (k // 64) * (region + 4096)
            if off + region < rb:
                arena[off: off + region] = pat
                dense += region
        torch.cuda.empty_cache()
        out, placement = ctx.pair_alloc(arena, out_cap) if PLACED else (torch.empty(out_cap, dtype=torch.uint8, device="cuda"), None)
        idx_out = torch.zeros(n * 32, dtype=torch.uint8, device="cuda")
        summary = torch.zeros(64, dtype=torch.uint8, device="cuda")
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ctx.emit_annexb_async(arena, rb, g["index"], n, 1, out, idx_out, summary)
        for i in range(4):
            ev[i].record()
            if i < 3:
                ctx.emit_annexb_async(arena, rb, g["index"], n, 1, out, idx_out, summary)
        torch.cuda.synchronize()
        ms = min(ev[i].elapsed_time(ev[i + 1]) for i in range(3))
        s = ctx.read_summary(summary)
        assert int(s["error"]) == 0, s
        by_tiles = int(ctx.lib.hbs_ctx_last_emit_by_tiles(ctx.h))
        total = int(s["stream_bytes"])
        line = {"ms": round(ms, 3), "over_uniform": round(ms / uniform_ms, 3), "dense_bytes": dense, "emitted_bytes": total,
                "arena_tiles_did_the_call": by_tiles, "output_placement": placement}
        if byte3 == 3:                                               # (the kernel by NALs takes 0.19 s on the zeros)
            out2 = torch.zeros(out_cap, dtype=torch.uint8, device="cuda")
            idx2 = torch.zeros(n * 32, dtype=torch.uint8, device="cuda")
            ctx.set_emit_path(0)
            try:
                ctx.emit_annexb_async(arena, rb, g["index"], n, 1, out2, idx2, summary)
                s2 = ctx.read_summary(summary)
            finally:
                ctx.set_emit_path(-1)
            assert int(s2["error"]) == 0 and int(s2["stream_bytes"]) == total
            assert torch.equal(out[:total], out2[:total]) and torch.equal(idx_out, idx2), "arena tiles != kernel by NALs on the mixed arena"
            line["check"] = "bytes and output index equal to the kernel by NALs'"
            del out2, idx2
        res[name] = line
        del arena, out, idx_out
    return res


def zero_heavy_line(torch, ctx, check):
    """hbs_index_extract over S(seed, 1 540 000 NALs, mode = zero-heavy) = ~16 GiB: kernel time by the library's events, fraction of
    the peak from the algorithmic bytes; the arena and the index against the generator's on the device, and (check) the first
    200 000 NALs through the reference's loop on the host (cpu_baseline)."""
    n1 = 1_540_000
    g1 = ctx.synth_stream(SEED, n1, 1)
    sb1, rb1 = g1["stream_bytes"], g1["rbsp_bytes"]
    stream1 = g1["stream"][:sb1]
    index1, rbsp1, summary1, cap1 = ctx.alloc_outputs(sb1, index_cap=n1 + 8, peer=stream1 if PLACED else None)
    ks = []
    for i in range(5):
        ctx.index_extract_async(stream1, index1, cap1, rbsp1, summary1)
        if i:
            ks.append(ctx.kernel_ms())
    s1 = ctx.read_summary(summary1)
    assert int(s1["error"]) == 0 and int(s1["nal_count"]) == n1 and int(s1["rbsp_bytes"]) == rb1, s1
    assert torch.equal(rbsp1[:rb1], g1["rbsp"][:rb1]), "zero-heavy: extracted RBSP != generated RBSP"
    a = index1[: n1 * 32].view(torch.int64).view(n1, 4)
    b = g1["index"][: n1 * 32].view(torch.int64).view(n1, 4)
    assert torch.equal(a[:, :3], b[:, :3]), "zero-heavy: NAL index != generator's index"
    ks.sort()
    k_ms = ks[len(ks) // 2]
    algo = sb1 + rb1 + 32 * n1
    line = {"value": round(sb1 / k_ms / 1e6, 1), "unit": "GB/s scanned", "kernel": KERNEL_NAMES.get(ctx.last_kernel(), "?"),
            "kernel_ms": round(k_ms, 4), "stream_bytes": sb1, "nals": n1, "algorithmic_bytes": algo,
            "roofline": {"bound": "hbm", "achieved": round(algo / k_ms / 1e6, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(algo / k_ms / 1e6 / HBM_PEAK_GBS, 4)},
            "workload": "S(seed=0x1234, n_nals=%d, zero-heavy): %.3f GiB, resident in HBM; arena and index equal to the generator's" % (n1, sb1 / 2**30)}
    if check:
        line["cpu_baseline"] = cpu_baseline(stream1, index1, rbsp1, n1, 200_000)
    return line


def other_kernels(torch, hbs, ctx, g, n, sweep=True, cpu_parse=True):
    """RBSP -> Annex-B over the bench arena; header parse + writers on BASELINE config 3 (4K30, ~100 k NALs)."""
    import ctypes as C
    import numpy as np
    from hevcbitstream_amd.api import PARSED, SUMMARY
    from hevcbitstream_amd.hevc_synth import stream_4k30
    res = {}
    sb, rb = g["stream_bytes"], g["rbsp_bytes"]
    # the placement pool: the headline arena has just been freed -- its sixteen chunks are on the pool's free list, classed; a new
    # arena against the same stream takes them (no new chunk, only the stream's sixteen pieces are probed again)
    t_a = time.perf_counter()
    again, rep2 = ctx.pair_alloc(g["stream"][:sb], sb + 16)
    torch.cuda.synchronize()
    t_again = time.perf_counter() - t_a
    idx_a = torch.empty((n + 8) * 32, dtype=torch.uint8, device="cuda")
    sum_a = torch.zeros(64, dtype=torch.uint8, device="cuda")
    ka = []
    for i in range(4):
        ctx.index_extract_async(g["stream"][:sb], idx_a, n + 8, again, sum_a)
        if i:
            ka.append(ctx.kernel_ms())
    ka.sort()
    assert int(ctx.read_summary(sum_a)["rbsp_bytes"]) == rb
    del again, idx_a
    res["placement_pool"] = {"second_placed_allocation": dict(rep2, seconds=round(t_again, 4), kernel_ms=round(ka[len(ka) // 2], 4)),
                             "note": "hbs_pair_alloc of 16 GiB against the bench stream after the timed loop's arena was freed: chunks off the pool's free list "
                                     "(from_pool), probes = the stream's pieces only; kernel_ms = hbs_index_extract into it"}
    # BASELINE's own 1 GiB configs (2: scan + nal_to_rbsp, 4: the rbsp_to_nal write path), each with a roofline of its own and the
    # whole stream compared with the reference's walk
    res["configs_1GiB"] = configs_1gib(torch, hbs, ctx, check=cpu_parse)
    # find_nal_unit alone: the same stream, no RBSP arena asked for (the streaming kernel of hbs_scan5.hip)
    index, _, summ0, cap = ctx.alloc_outputs(sb, index_cap=n + 8, want_rbsp=False)
    kms = []
    for i in range(4):
        ctx.index_extract_async(g["stream"][:sb], index, cap, None, summ0)
        if i:
            kms.append(ctx.kernel_ms())
    s0 = ctx.read_summary(summ0)
    assert int(s0["error"]) == 0 and int(s0["nal_count"]) == n
    a = index[: n * 32].view(torch.int64).view(n, 4)
    b = g["index"][: n * 32].view(torch.int64).view(n, 4)
    assert torch.equal(a[:, :3], b[:, :3]), "index-only NAL index != generator's index"
    ms = sum(kms) / len(kms)
    res["index_only"] = {"value": round(sb / ms / 1e6, 1), "unit": "GB/s scanned", "kernel_ms": round(ms, 4),
                         "read_frac_of_hbm_peak": round((sb + 32 * n) / ms / 1e6 / HBM_PEAK_GBS, 4),
                         "kernel": KERNEL_NAMES.get(ctx.last_kernel(), "?"),
                         "workload": "the bench stream, index only (find_nal_unit over the stream, no arena)"}
    del index
    res["mixed_stream"] = mixed_stream_line(torch, ctx, g["stream"][:sb], sb, n + 64, g["uniform_kernel_ms"])
    res["mixed_stream"]["index_only"] = mixed_index_only(torch, ctx, g["stream"][:sb], sb, n + 64, ms)
    # SURVEY 8(d)'s second mode: the zero-heavy stress stream at 16 GiB (every NAL full of zero runs and emulation prevention
    # bytes: thousands of elements per tile), same call, its own roofline; a sample of it through the reference on the host
    try:
        res["zero_heavy_16GiB"] = zero_heavy_line(torch, ctx, cpu_parse)
    except Exception as e:                                            # noqa: BLE001  (reported, never allowed to take the line down)
        res["zero_heavy_16GiB"] = {"error": "%s: %s" % (type(e).__name__, e)}
    torch.cuda.empty_cache()                                          # (what torch keeps cached is not free memory to hbs_pair_alloc's candidates)
    out, emit_placement = ctx.pair_alloc(g["rbsp"], sb + 4096) if PLACED else (torch.empty(sb + 4096, dtype=torch.uint8, device="cuda"), None)
    idx_out = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
    summary = torch.zeros(64, dtype=torch.uint8, device="cuda")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ctx.emit_annexb_async(g["rbsp"], rb, g["index"], n, 1, out, idx_out, summary)
    for i in range(4):
        ev[i].record()
        if i < 3:
            ctx.emit_annexb_async(g["rbsp"], rb, g["index"], n, 1, out, idx_out, summary)
    torch.cuda.synchronize()
    ms = min(ev[i].elapsed_time(ev[i + 1]) for i in range(3))
    assert torch.equal(out[:sb], g["stream"][:sb]), "emitted stream != generated stream"
    res["emit_annexb"] = {"value": round(sb / ms / 1e6, 1), "unit": "GB/s emitted", "ms": round(ms, 3),
                          "hbm_traffic_GBs": round((rb + sb) / ms / 1e6, 1), "workload": "the bench arena: %d NALs, %.2f GiB" % (n, rb / 2**30),
                          "output_placement": ("hbs_pair_alloc against the arena: %s" % json.dumps(emit_placement)) if PLACED else "torch allocator"}
    uniform_emit_ms = ms
    del out, idx_out
    # K3 on an arena with stretches the density probe does not see (1 % of it in 640 KiB stretches of 00 00 03 padding, then of
    # zeros): the tile kernel walks those tiles by rows; bytes and output index compared with the kernel by NALs.  Reported, never
    # allowed to take the line down: a failure here is recorded as text.
    try:
        res["emit_annexb"]["mixed_arena"] = emit_mixed_lines(torch, ctx, g, n, rb, sb, uniform_emit_ms)
    except Exception as e:                                            # noqa: BLE001
        res["emit_annexb"]["mixed_arena"] = {"error": "%s: %s" % (type(e).__name__, e)}
    torch.cuda.empty_cache()
    stream, _ = stream_4k30(11, n_pictures=12500, slices_per_picture=8, idr_every=60, payload_bytes=(60, 120))
    d = torch.from_numpy(np.frombuffer(stream, dtype=np.uint8).copy()).cuda()
    index, rbsp, summ, cap = ctx.alloc_outputs(d.numel())
    ctx.index_extract_async(d, index, cap, rbsp, summ)
    m = int(ctx.read_summary(summ)["nal_count"])
    parsed, structs = ctx.parse_headers(rbsp, index, m)
    assert int((parsed["rc"] < 0).sum()) == 0
    pt = torch.empty(m * PARSED.itemsize, dtype=torch.uint8, device="cuda")
    sm = torch.zeros(SUMMARY.itemsize, dtype=torch.uint8, device="cuda")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    for i in range(6):
        ctx.parse_headers_async(rbsp, index, m, pt, structs, sm)
        ev[i].record()
    torch.cuda.synchronize()
    ms = min(ev[i].elapsed_time(ev[i + 1]) for i in range(5))
    res["parse_headers"] = {"value": round(m / ms / 1e3, 1), "unit": "M NAL/s", "ms": round(ms, 3),
                            "workload": "synthetic 4K30 stream, %d NALs (VPS/SPS/PPS every 60 pictures, 8 slices per picture)" % m}
    # the compact parse of the same batch (hbs_parse_headers_compact): the same walk into sinks, a 64-byte record per slice instead
    # of a 4 024-byte struct; every member of every record compared with the struct the full parse filled
    from hevcbitstream_amd.api import COMPACT
    cpt = torch.empty(m * PARSED.itemsize, dtype=torch.uint8, device="cuda")
    cct = torch.empty(m * COMPACT.itemsize, dtype=torch.uint8, device="cuda")
    ctx.parse_compact_async(rbsp, index, m, cpt, cct, None, sm)
    cneed = int(ctx.read_summary(sm)["reserved"][0])
    cst = torch.empty(cneed + 16, dtype=torch.uint8, device="cuda")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    for i in range(6):
        ctx.parse_compact_async(rbsp, index, m, cpt, cct, cst, sm)
        ev[i].record()
    torch.cuda.synchronize()
    cms = min(ev[i].elapsed_time(ev[i + 1]) for i in range(5))
    assert int(ctx.read_summary(sm)["error"]) == 0
    n_slices = cpu_baseline_compact(parsed, structs, cpt, cct) if cpu_parse else None
    res["parse_headers_compact"] = {"value": round(m / cms / 1e3, 1), "unit": "M NAL/s", "ms": round(cms, 3), "struct_arena_bytes": cneed,
                                    "full_parse_struct_arena_bytes": int(structs.numel()),
                                    "check": ("rc, NAL header, slice_data_off / slice_data_size of all %d NALs and the sixteen members of all %d slice records equal "
                                              "to the full parse's, and (below) to the reference's own structs" % (m, n_slices)) if cpu_parse else "not run (--cpu-sample-nals 0)"}
    if cpu_parse:
        # config 3 pinned on the reference: the structs of this sequence against read_hevc_nal_unit on the host, all NALs;
        # config3_end_to_end below requires the 2.1 GiB pipeline (with and without arena) to produce these same structs
        res["parse_headers"]["cpu_baseline"] = cpu_baseline_parse(d, index, rbsp, m, parsed, structs, compact=(cpt, cct))
        res["parse_headers_compact"]["checked_against_the_reference_directly"] = res["parse_headers"]["cpu_baseline"]["compact_slices_checked"]
    res["config3_end_to_end"] = config3_end_to_end(torch, hbs, ctx, d, index, rbsp, m, parsed, structs, cpu_parse=cpu_parse)
    if cpu_parse:
        res["config3_end_to_end"]["parity"] = ("the 2.1 GiB run's own outputs (with the arena) compared with the %s on all %d NALs (cpu_baseline_at_full_size); "
                                               "the runs without an arena equal to that run's index, records and structs"
                                               % (res["parse_headers"]["cpu_baseline"]["kind"], m))
    parsed_dev = torch.from_numpy(parsed.view(np.uint8).copy()).cuda()
    wcap = 256
    written, wout = ctx.write_headers(parsed_dev, structs, m, wcap)
    assert int((written["rc"] < 0).sum()) == 0
    wr = torch.empty(m * 16, dtype=torch.uint8, device="cuda")
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    for i in range(5):
        ctx._bind_stream()
        rc = ctx.lib.hbs_write_headers(ctx.h, C.c_void_p(parsed_dev.data_ptr()), m, C.c_void_p(structs.data_ptr()), None, None,
                                       C.c_void_p(wout.data_ptr()), wcap, C.c_void_p(wr.data_ptr()))
        assert rc == 0
        ev[i].record()
    torch.cuda.synchronize()
    ms = min(ev[i].elapsed_time(ev[i + 1]) for i in range(4))
    res["write_headers"] = {"value": round(m / ms / 1e3, 1), "unit": "M NAL/s", "ms": round(ms, 3), "workload": "the structs of that parse"}
    # stream shape: extract / index only / emit against the mean NAL size (2 GiB of random payload each, 3- and 4-byte start
    # codes), every result checked (scripts/nal_sweep.py); the fractions are of the 8 TB/s peak, at a size where the launches
    # and the last partial wave of tiles still cost a few points against the 16 GiB figures above
    del parsed_dev, wout, wr, structs, rbsp, index, d
    torch.cuda.empty_cache()
    if sweep:
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import nal_sweep
        res["nal_size_sweep"] = nal_sweep.sweep(torch, hbs, ctx, [64, 256, 384, 512, 1024, 2048, 4096, 10240, 65536, 524288], 2.0)
    return res


def record_scalars(ok):
    """other_kernels' figures that the round's targets are stated in, flat: {key: number} for `roofline`"""
    r = {}

    def put(key, *path, digits=4):
        v = ok
        try:
            for k in path:
                v = v[k]
            r[key] = round(float(v), digits)
        except (KeyError, TypeError, ValueError, IndexError):
            pass
    put("frac_1GiB_extract", "configs_1GiB", "config2_extract", "roofline", "frac_of_call")
    put("frac_1GiB_extract_kernel", "configs_1GiB", "config2_extract", "roofline", "frac")
    put("frac_1GiB_index_only", "configs_1GiB", "config2_index_only", "roofline", "frac_of_call")
    put("frac_1GiB_index_only_kernel", "configs_1GiB", "config2_index_only", "roofline", "frac")
    put("frac_1GiB_emit", "configs_1GiB", "config4_emit", "roofline", "frac")
    put("ms_1GiB_extract_call", "configs_1GiB", "config2_extract", "call_ms")
    put("ms_1GiB_extract_kernel", "configs_1GiB", "config2_extract", "kernel_ms")
    put("ms_1GiB_index_only_call", "configs_1GiB", "config2_index_only", "call_ms")
    put("ms_1GiB_emit_call", "configs_1GiB", "config4_emit", "call_ms")
    put("frac_index_only_16GiB", "index_only", "read_frac_of_hbm_peak")
    put("ms_index_only_16GiB", "index_only", "kernel_ms")
    put("frac_zero_heavy_16GiB", "zero_heavy_16GiB", "roofline", "frac")
    put("ms_zero_heavy_16GiB", "zero_heavy_16GiB", "kernel_ms")
    try:
        e = ok["emit_annexb"]
        r["frac_emit_16GiB"] = round(e["hbm_traffic_GBs"] / HBM_PEAK_GBS, 4)
        r["ms_emit_16GiB"] = e["ms"]
    except (KeyError, TypeError):
        pass
    put("mixed_over_uniform", "mixed_stream", "over_uniform")
    put("mixed_index_only_over_uniform", "mixed_stream", "index_only", "over_uniform")
    put("mixed_emit_padding_over_uniform", "emit_annexb", "mixed_arena", "padding_00_00_03", "over_uniform")
    put("mixed_emit_zeros_over_uniform", "emit_annexb", "mixed_arena", "zeros", "over_uniform")
    put("config3_ms", "config3_end_to_end", "without_arena_compact", "ms")
    put("config3_full_structs_ms", "config3_end_to_end", "without_arena", "ms")
    put("config3_with_arena_ms", "config3_end_to_end", "ms")
    put("parse_headers_MNALs", "parse_headers", "value", digits=1)
    put("parse_headers_compact_MNALs", "parse_headers_compact", "value", digits=1)
    put("write_headers_MNALs", "write_headers", "value", digits=1)
    for row in ok.get("nal_size_sweep", []) or []:
        try:
            b = int(row["mean_nal_bytes"])
            r["sweep_extract_%d" % b] = row["extract"]["traffic_frac"]
            r["sweep_index_only_%d" % b] = row["index_only"]["read_frac"]
            r["sweep_emit_%d" % b] = row["emit"]["traffic_frac"]
        except (KeyError, TypeError, ValueError):
            pass
    return r


def launch_ranks(n, argv, script=None):
    """One process per GPU on this node, started from a parent that holds no GPU state: `sys.executable bench.py <same
    arguments>` with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set (what torch.distributed.run would set;
    the rendezvous is env:// on 127.0.0.1 -- the container's hostname may not resolve).  Rank 0's stdout is captured and
    its last line -- the JSON line -- printed last; everything else the ranks print goes to stderr.  Returns the exit code:
    0 only when every rank exited 0 and rank 0 printed a JSON line; a rank that fails takes the others down (no retry)."""
    import socket
    import subprocess
    import threading
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    base.setdefault("OMP_NUM_THREADS", "1")
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0")
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=env, cwd=os.getcwd(),
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    out0 = []
    reader = threading.Thread(target=lambda: out0.extend(procs[0].stdout.read().decode(errors="replace").splitlines()), daemon=True)
    reader.start()
    codes = [None] * n
    failed = None
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
                if codes[r] not in (None, 0) and failed is None:
                    failed = r
        if failed is not None:
            break
        time.sleep(0.05)
    if failed is not None:
        # a rank died: the others would wait in a collective for ever -- give them a moment, then end exactly the children started here
        deadline = time.time() + 10.0
        for r, p in enumerate(procs):
            if codes[r] is None:
                try:
                    codes[r] = p.wait(timeout=max(0.1, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    codes[r] = p.wait()
    reader.join(timeout=30)
    line = None
    for text in out0:
        if text.startswith("{") and '"metric"' in text:
            line = text
        else:
            print(text, file=sys.stderr)
    sys.stderr.flush()
    if failed is not None:
        print("bench.py launcher: rank %d exited with code %s (all: %s)" % (failed, codes[failed], codes), file=sys.stderr)
        return codes[failed] if codes[failed] and codes[failed] > 0 else 1
    if line is None:
        print("bench.py launcher: rank 0 printed no JSON line", file=sys.stderr)
        return 1
    print(line, flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--nals", type=int, default=N_NALS_16GIB, help="NALs per GPU (default: 16 GiB of stream)")
    ap.add_argument("--mode", type=int, default=0, help="0 uniform payload (headline), 1 zero-heavy")
    ap.add_argument("--cpu-sample-nals", type=int, default=1_000_000, help="0 disables the CPU baseline leg")
    ap.add_argument("--other-kernels", type=int, default=1, help="0 skips the emit / parse / write measurements (N = 1 only)")
    ap.add_argument("--sweep", type=int, default=1, help="0 skips other_kernels' NAL-size sweep (profiling passes)")
    ap.add_argument("--placed-arena", type=int, default=0,
                    help="1: the RBSP arena (and the other lines' outputs) from hbs_pair_alloc, placed against their inputs by measurement, instead of torch's "
                         "allocator.  Off by default since round 6: over six processes the median gain was 1.3 %% of the kernel's time (profiles/r06/pair_time.txt); "
                         "the default run still reports both arenas (roofline.kernel_ms_placed_arena / kernel_ms_plain_arena)")
    ap.add_argument("--exercise-gather", action="store_true",
                    help="dev aid: run the N > 1 code path (RCCL group, pipelined index gather) with a one-rank group on one GPU")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python3 bench.py --gpus N` as typed: this process becomes the launcher.  It has not imported torch and never
        # touches the GPU; the ranks are fresh child processes and rank 0's JSON line is relayed as the last stdout line.
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import hevcbitstream_amd as hbs

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks (tests/test_gpu_shard.py runs this file with 2 ranks on ONE GPU): every rank on one device, torch.distributed over
    # gloo (RCCL refuses two ranks on a device; the library's own exchange then goes through HBS_RCCL_LIB's stand-in)
    if os.environ.get("HBS_BENCH_ONE_DEVICE"):
        local_rank = int(os.environ["HBS_BENCH_ONE_DEVICE"])
    backend = os.environ.get("HBS_BENCH_DIST_BACKEND", "nccl")
    cdev = "cuda" if backend == "nccl" else "cpu"          # where the tensors of the few host-side collectives live
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dist = None
    multi = world > 1 or args.exercise_gather           # "multi" = the code path with the exchange in it
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group(backend, rank=0, world_size=1, **({"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}))
        else:
            dist.init_process_group(backend, **({"device_id": torch.device("cuda", local_rank)} if backend == "nccl" else {}))

    ctx = hbs.Context(local_rank)
    ctx.enable_timing(True)
    # one process per GPU and this context is the only one that runs persistent kernels on it: say so (first tiles by workgroup
    # number instead of by ticket, ~1 % of a 1 GiB call; the test hook that puts every rank on ONE device must not)
    exclusive = 0 if os.environ.get("HBS_BENCH_ONE_DEVICE") else 1
    ctx.set_device_exclusive(exclusive)
    n = args.nals
    from hevcbitstream_amd.shard import shard_seed
    g = ctx.synth_stream(shard_seed(SEED, rank), n, args.mode)   # independent shard per rank, generated in HBM
    sb, rb = g["stream_bytes"], g["rbsp_bytes"]
    stream = g["stream"][:sb]
    gen_rbsp, gen_index = g["rbsp"], g["index"]
    # the RBSP arena: torch's allocator by default; --placed-arena 1 has the library allocate it AGAINST the stream it will be
    # written from (hbs_pair_alloc: on MI355X a stream / arena pair runs a few per cent slower when both lie in the same one of
    # two classes of physical memory -- DESIGN.md section 3).  Opt-in since round 6: the median gain over six processes was 1.3 %
    t_alloc = time.perf_counter()
    global PLACED
    PLACED = bool(args.placed_arena)
    if not PLACED:
        os.environ["HBS_PLAIN_ALLOC"] = "1"                   # (scripts/nal_sweep.py: its outputs from torch's allocator too)
    index, rbsp, summary, cap = ctx.alloc_outputs(sb, index_cap=n + 8, peer=stream if PLACED else None)
    torch.cuda.synchronize()
    placement = dict(ctx.last_pair_report, seconds=round(time.perf_counter() - t_alloc, 3)) if PLACED else None

    from hevcbitstream_amd import shard
    # N > 1: the one exchange of the path is the gather of the NAL index -- the C ABI's hbs_gather_index (counts, then exactly
    # count x 32 bytes per rank, RCCL).  It runs on a stream of its own while the next step's scan fills the other index buffer.
    indexes = [index, torch.empty_like(index)] if multi else [index]
    gatherer = shard.PipelinedLibraryGather(torch, hbs, local_rank, dist, rank, world, cap, depth=2) if multi else None
    if gatherer is not None and gatherer.comm.world_seen() != world:
        raise SystemExit("bench.py: the communicator reports %d ranks, %d were launched" % (gatherer.comm.world_seen(), world))
    spare_wgs = 0
    if gatherer is not None:
        # the scan's persistent workgroups fill the GPU; RCCL's kernels need somewhere to run beside it: 8 slots per peer,
        # 32 ... 64 of the 512 (hbs_comm_reserve_hint; HBS_BENCH_SPARE_WGS overrides for experiments)
        spare_wgs = int(os.environ.get("HBS_BENCH_SPARE_WGS", gatherer.comm.reserve_hint()))
        ctx.reserve_workgroups(spare_wgs)
    counter = [0]
    pending = [None]

    def step():
        k = counter[0] % len(indexes)
        buf = indexes[k]
        counter[0] += 1
        if gatherer is not None:
            gatherer.release(k)          # the gather of two steps ago still reads (and the scan's prologue clears) this buffer
        ctx.index_extract_async(stream, buf, cap, rbsp, summary)
        if gatherer is not None:
            gatherer.mark_scan(k)
            if pending[0] is not None:   # the previous step's gather: its host side waits for that scan, this one is already queued
                gatherer.submit(pending[0], indexes[pending[0]], n)
            pending[0] = k

    def fence():
        if gatherer is not None:
            if pending[0] is not None:
                gatherer.submit(pending[0], indexes[pending[0]], n)
                pending[0] = None
            gatherer.drain()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    if gatherer is not None:            # RCCL builds its rings on first use: not inside anybody's timed region, whatever --warmup says
        gatherer.mark_scan(0)
        gatherer.submit(0, indexes[0], 0)
        gatherer.drain()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    # the kernel's duration in EVERY step of the timed loop: the library records an event pair per call (a ring of 64) on the
    # stream the kernel runs on; they are read here, behind the fence -- no wait inside the loop, and kernel_ms <= ms_per_step
    kms = [ctx.kernel_ms_back(b) for b in range(min(args.steps, 64))]
    gms = gatherer.gather_ms(args.steps) if gatherer is not None else []
    s = ctx.read_summary(summary)

    # parity properties at full size (outside the timed region)
    assert int(s["error"]) == 0 and int(s["nal_count"]) == n and int(s["stop_reason"]) == -1, s
    assert int(s["rbsp_bytes"]) == rb
    assert torch.equal(rbsp[:rb], gen_rbsp[:rb]), "extracted RBSP != generated RBSP"
    a = index[: n * 32].view(torch.int64).view(n, 4)
    b = gen_index[: n * 32].view(torch.int64).view(n, 4)
    assert torch.equal(a[:, :3], b[:, :3]), "NAL index != generator's index"

    if multi:
        # what the exchange delivered in the last step: every rank's rows back to back in rank order; MY rows must be my index
        # (the others are checked by their owners, and all ranks hold the same bytes)
        last_k = (counter[0] - 1) % len(indexes)
        all_index, counts = gatherer.result(last_k)
        assert counts == [n] * world, counts
        mine = all_index[rank * n * 32: (rank + 1) * n * 32].view(torch.int64).view(n, 4)
        assert torch.equal(mine[:, :3], b[:, :3]), "gathered index rows of this rank != its index"
    if multi:
        tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        tot = torch.tensor([float(sb), float(n)], dtype=torch.float64, device=cdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_bytes, total_nals = float(tot[0].item()), float(tot[1].item())
        # every rank's own kernel and gather times, for the per-rank lines
        mine_t = torch.tensor([sum(kms) / len(kms), (sum(gms) / len(gms)) if gms else 0.0, float(sb + rb + 32 * n)], dtype=torch.float64, device=cdev)
        per_rank_t = torch.empty(world * 3, dtype=torch.float64, device=cdev)
        dist.all_gather_into_tensor(per_rank_t, mine_t)
        per_rank_t = per_rank_t.view(world, 3).cpu().tolist()
        # every rank's arena placement (chunks wanted / placed in the class its piece of the stream is not in): a slow rank can
        # then be told from a misplaced one
        per_rank_place = [None] * world
        dist.all_gather_object(per_rank_place, placement)
    else:
        total_bytes, total_nals = float(sb), float(n)
        per_rank_t = None
        per_rank_place = None

    if rank == 0:
        ms_step = dt / args.steps * 1e3
        k_ms = sum(kms) / len(kms)
        algo_bytes = sb + rb + 32 * n
        achieved = algo_bytes / (k_ms * 1e-3) / 1e9
        blocks, per_cu = ctx.grid()
        variant = ctx.last_kernel()          # automatic mode: the kernel the density probe picked
        kernel_name = KERNEL_NAMES[variant]
        geometry = KERNEL_GEOMETRY[variant]
        out = {
            "metric": "Annex-B GB/s scanned + NAL units/s, 16 GiB synthetic stream, 1/2/4/8 MI355X",
            "value": round(total_bytes * args.steps / dt / 1e9, 2),
            "unit": "GB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "nal_units_per_s": round(total_nals * args.steps / dt, 1),
            "config": {"workload": "start-code scan + NAL index + RBSP extraction (hbs_index_extract) over "
                                   "S(seed=0x1234+rank, n_nals=%d, mode=%s): %.3f GiB Annex-B per GPU, ~10 KiB NALs, "
                                   "resident in HBM%s" % (n, "uniform" if args.mode == 0 else "zero-heavy", sb / 2**30,
                                                          "; + RCCL all-gather of the NAL index" if multi else ""),
                       "stream_bytes_per_gpu": sb, "nals_per_gpu": n,
                       "parallelism": "%d independent shard(s), one per GPU%s" % (world, "; index gathered to every rank by hbs_gather_index "
                                       "(C ABI, RCCL world %d as the communicator reports it), pipelined under the next step's scan" % gatherer.comm.world_seen() if multi else ""),
                       "grid": "%d persistent workgroups (%d per CU) x %s" % (blocks, per_cu, geometry),
                       "device_exclusive": exclusive,
                       "arena_placement": ("hbs_pair_alloc against the stream (1 GiB chunks classed by measurement, outside the timed region): %s"
                                           % json.dumps(placement)) if placement is not None else "torch allocator (--placed-arena 1: hbs_pair_alloc)"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                         "kernel": kernel_name, "kernel_ms": round(k_ms, 4),
                         "kernel_ms_steps": [round(x, 3) for x in reversed(kms)],      # every timed step, first to last; kernel_ms is their mean
                         "algorithmic_bytes": algo_bytes,
                         "note": "bytes = stream read once + RBSP written once + 32 B/NAL index; "
                                 "read-only fraction = %.4f" % (sb / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS)},
        }
        if multi:
            # the exchange reported by itself -- events on its own stream around hbs_gather_index (all-gather of
            # the counts, the host's wait for them, count x 32 B per rank to every rank), mean over the timed steps, rank 0's
            # and the slowest rank's; it overlaps the next step's kernel, so it adds to ms_per_step only where it is longer
            out["gather"] = {"gather_ms": round(sum(gms) / len(gms), 4) if gms else None,
                             "gather_ms_max_over_ranks": round(max(r[1] for r in per_rank_t), 4),
                             "bytes_received_per_rank_per_step": int(total_nals) * 32,
                             "api": "hbs_gather_index (include/hevcbitstream_amd.h), root = -1", "rccl_world": gatherer.comm.world_seen(),
                             "kernel_ms": round(k_ms, 4)}
            out["per_rank"] = [{"rank": r, "kernel_ms": round(t[0], 4), "gather_ms": round(t[1], 4),
                                "roofline_frac": round(t[2] / (t[0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                "arena_placement": per_rank_place[r]} for r, t in enumerate(per_rank_t)]
            out["gather"]["reserved_workgroups"] = spare_wgs
        # HBM bytes per launch by PMC (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 passes): counters cannot be read from
        # inside this process, so the figure of the committed profile of this very workload and kernel is quoted
        tr = pmc_traffic(kernel_name, algo_bytes)
        if tr is not None:
            out["roofline"]["traffic"] = tr["traffic_bytes_per_launch"]
            out["roofline"]["traffic_source"] = tr["source"]
        if world == 1 and args.other_kernels:
            # what placement buys, in THIS process: the same steps into the other kind of arena (outside the timed region) -- torch's
            # allocator when the timed steps' arena was placed, hbs_pair_alloc against the stream otherwise
            t_b = time.perf_counter()
            index_p, rbsp_p, summary_p, cap_p = ctx.alloc_outputs(sb, index_cap=n + 8, peer=None if PLACED else stream)
            torch.cuda.synchronize()
            other_report = None if PLACED else dict(ctx.last_pair_report, seconds=round(time.perf_counter() - t_b, 3))
            kp = []
            for i in range(6):
                ctx.index_extract_async(stream, index_p, cap_p, rbsp_p, summary_p)
                if i:
                    kp.append(ctx.kernel_ms())
            assert int(ctx.read_summary(summary_p)["rbsp_bytes"]) == rb
            kp.sort()
            del rbsp_p, index_p
            torch.cuda.empty_cache()
            k_other = kp[len(kp) // 2]
            k_plain, k_placed = (k_other, k_ms) if PLACED else (k_ms, k_other)
            out["roofline"]["kernel_ms_plain_arena"] = round(k_plain, 4)
            out["roofline"]["kernel_ms_placed_arena"] = round(k_placed, 4)
            out["roofline"]["placement"] = {"timed_steps_arena": "placed" if PLACED else "plain", "kernel_ms_placed_arena": round(k_placed, 4),
                                            "kernel_ms_plain_arena": round(k_plain, 4), "plain_over_placed": round(k_plain / k_placed, 4),
                                            "placed_allocation": other_report if not PLACED else placement,
                                            "note": "same process, same stream: an arena from torch's allocator against one from hbs_pair_alloc (placed against the "
                                                    "stream by measurement); median of five calls each"}
        if world == 1 and args.cpu_sample_nals > 0:            # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(stream, index, rbsp, n, min(args.cpu_sample_nals, n))
        if world == 1 and args.other_kernels:
            g["uniform_kernel_ms"] = k_ms
            del rbsp, index
            ok = other_kernels(torch, hbs, ctx, g, n, sweep=bool(args.sweep), cpu_parse=args.cpu_sample_nals > 0)
            # What a reader of the DRIVER's record must be able to check is repeated as scalar keys of `roofline` (the driver keeps
            # those and cuts nested objects and the long line's head): fractions of the 8 TB/s peak unless the key says ms / M NAL/s.
            out["roofline"].update(record_scalars(ok))
            # ... and the line's tail is what survives of other_kernels: the sweep first, BASELINE's own 1 GiB configs last
            first = ["nal_size_sweep", "placement_pool", "write_headers", "parse_headers", "parse_headers_compact"]
            last = ["emit_annexb", "mixed_stream", "index_only", "zero_heavy_16GiB", "config3_end_to_end", "configs_1GiB"]
            order = [k for k in first if k in ok] + [k for k in ok if k not in first and k not in last] + [k for k in last if k in ok]
            out["other_kernels"] = {k: ok[k] for k in order}
        # RCCL writes a version banner to C stdout, which is block-buffered when piped: push it out first, so that the JSON
        # line is the LAST line of rank 0's stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)
    if multi:
        gatherer.close()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
