"""Parity at full size against the REFERENCE, not against the device generator: a bench stream of more than
5 GiB (offsets past 4 GiB) goes through the compiled reference's own loop on the host -- find_nal_unit +
nal_to_rbsp per NAL (hevc_analyze.c:135-177, h264_nal.c:38-200; oracle/_ref/libref_driver.so) -- and EVERY start /
end / rbsp_off / rbsp_len of the GPU index and EVERY byte of the GPU's RBSP arena are compared with what it
produced; then the way back: the stream K3 emits from that arena against the reference's rbsp_to_nal
(h264_nal.c:92-132) over the reference's arena.  Uniform and zero-heavy payload.  Where the prebuilt reference
did not travel with the tree, the oracle's restatement (pinned to it by tests/test_oracle_l2.py) stands in."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_ENTRY = np.dtype([("start", "<u8"), ("end", "<u8"), ("rbsp_off", "<u8"), ("rbsp_len", "<i4"),
                      ("rc_rbsp", "<i4"), ("rc_find", "<i4"), ("pad", "<i4")])
u8p = C.POINTER(C.c_uint8)


def reference_walk(host, n_cap):
    """(entries, arena, rbsp_bytes, kind) of the reference's loop over `host`"""
    arena = np.empty(len(host) + 64, dtype=np.uint8)
    drv = os.path.join(ROOT, "oracle", "_ref", "libref_driver.so")
    if os.path.exists(drv):
        lib = C.CDLL(drv)
        lib.ref_walk_index.restype = C.c_int64
        lib.ref_walk_index.argtypes = [u8p, C.c_int64, u8p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        ent = np.zeros(n_cap, dtype=REF_ENTRY)
        tot = C.c_int64(0)
        n = lib.ref_walk_index(host.ctypes.data_as(u8p), len(host), arena.ctypes.data_as(u8p), len(arena), ent.ctypes.data, n_cap, C.byref(tot))
        assert 0 <= n <= n_cap
        return ent[:n], arena, int(tot.value), "reference"
    from tests import _orc
    orc = _orc.oracle()
    idx = np.zeros(n_cap, dtype=_orc.NAL_ENTRY)
    why = C.c_int(0)
    n = orc.lib.orc_index_stream(host.ctypes.data_as(u8p), len(host), idx.ctypes.data, n_cap, C.byref(why))
    tot = orc.lib.orc_extract_rbsp(host.ctypes.data_as(u8p), idx.ctypes.data, n, arena.ctypes.data_as(u8p), len(arena))
    ent = np.zeros(n, dtype=REF_ENTRY)
    for f in ("start", "end", "rbsp_off"):
        ent[f] = idx[f][:n]
    ent["rbsp_len"] = idx["rbsp_len"][:n]
    ent["rc_rbsp"] = np.where(idx["status"][:n] & 1, -1, 0)
    return ent, arena, int(tot), "port"


def reference_emit(arena, ent, out_cap):
    out = np.empty(out_cap, dtype=np.uint8)
    drv = os.path.join(ROOT, "oracle", "_ref", "libref_driver.so")
    off = np.ascontiguousarray(ent["rbsp_off"], dtype=np.uint64)
    ln = np.ascontiguousarray(ent["rbsp_len"], dtype=np.int32)
    if os.path.exists(drv):
        lib = C.CDLL(drv)
        lib.ref_emit_synthetic.restype = C.c_int64
        lib.ref_emit_synthetic.argtypes = [u8p, C.c_void_p, C.c_void_p, C.c_int64, u8p, C.c_int64]
        m = lib.ref_emit_synthetic(arena.ctypes.data_as(u8p), off.ctypes.data, ln.ctypes.data, len(ent), out.ctypes.data_as(u8p), out_cap)
    else:
        from tests import _orc
        orc = _orc.oracle()
        m, rs, ns = 0, C.c_int(0), C.c_int(0)
        for k in range(len(ent)):          # start code rule of the synthetic stream, then the oracle's rbsp_to_nal
            sc = 4 if k % 4 == 0 else 3
            out[m:m + sc - 1] = 0
            out[m + sc - 1] = 1
            m += sc
            rs.value = int(ln[k])
            orc.lib.orc_rbsp_to_nal(arena[int(off[k]):].ctypes.data_as(u8p), C.byref(rs), out[m:].ctypes.data_as(u8p), C.byref(ns))
            m += ns.value
    assert m > 0
    return out, int(m)


def device_equals_host(dev, host, nbytes, what):
    """dev[:nbytes] == host[:nbytes], a piece at a time (bounded host memory)"""
    import torch
    step = 1 << 29
    for lo in range(0, nbytes, step):
        hi = min(nbytes, lo + step)
        piece = dev[lo:hi].cpu().numpy()
        if not np.array_equal(piece, host[lo:hi]):
            bad = lo + int(np.flatnonzero(piece != host[lo:hi])[0])
            raise AssertionError("%s differs from the reference's at byte %d of %d" % (what, bad, nbytes))
    del torch


@pytest.mark.parametrize("mode,n_nals", [(0, 600_000), (1, 540_000)], ids=["uniform-5.7GiB", "zero-heavy-5.2GiB"])
def test_index_arena_and_reemission_against_the_reference(mode, n_nals):
    import torch
    import hevcbitstream_amd as hbs
    ctx = hbs.Context(0)
    g = ctx.synth_stream(0x1234, n_nals, mode)
    sb, rb = g["stream_bytes"], g["rbsp_bytes"]
    assert sb > (5 << 30), "the stream must reach past 4 GiB offsets"
    stream = g["stream"][:sb]
    host = stream.cpu().numpy()

    # the reference's walk over the same bytes
    ent, ref_arena, ref_rb, kind = reference_walk(host, n_nals + 16)
    assert len(ent) == n_nals, (len(ent), n_nals, kind)

    # scan + index + extraction on the GPU: the default (automatic) path, then every kernel pinned
    index, rbsp, summary, cap = ctx.alloc_outputs(sb, index_cap=n_nals + 16)
    for kernel in (0, 4, 2):
        ctx.set_kernel(kernel)
        index.zero_()
        rbsp.zero_()
        ctx.index_extract_async(stream, index, cap, rbsp, summary)
        s = ctx.read_summary(summary)
        assert int(s["error"]) == 0 and int(s["nal_count"]) == n_nals and int(s["stop_reason"]) == -1, (kernel, s)
        assert int(s["rbsp_bytes"]) == ref_rb, (kernel, int(s["rbsp_bytes"]), ref_rb)
        got = index[: n_nals * 32].cpu().numpy().view(hbs.NAL_ENTRY)
        for f in ("start", "end", "rbsp_off"):
            assert np.array_equal(got[f], ent[f]), (kernel, f, int(np.flatnonzero(got[f] != ent[f])[0]))
        assert np.array_equal(got["rbsp_len"].astype(np.int64), ent["rbsp_len"].astype(np.int64)), (kernel, "rbsp_len")
        assert np.array_equal((got["status"] & hbs.ST_ERROR) != 0, ent["rc_rbsp"] < 0), (kernel, "status")
        assert int(got["start"][-1]) > (4 << 30)
        device_equals_host(rbsp, ref_arena, ref_rb, "kernel %d: RBSP arena" % kernel)
    # the index-only kernel: same entries without an arena
    ctx.set_kernel(5)
    index.zero_()
    ctx.index_extract_async(stream, index, cap, None, summary)
    s = ctx.read_summary(summary)
    assert int(s["error"]) == 0 and int(s["nal_count"]) == n_nals
    got5 = index[: n_nals * 32].cpu().numpy().view(hbs.NAL_ENTRY)
    for f in ("start", "end"):
        assert np.array_equal(got5[f], ent[f]), (5, f)
    ctx.set_kernel(0)
    del host

    # the way back: K3 over the GPU's arena against the reference's rbsp_to_nal over the reference's arena
    ref_stream, ref_sb = reference_emit(ref_arena, ent, sb + (1 << 16))      # room for the 3/2 bound of the last NAL
    assert ref_sb == sb, (ref_sb, sb)
    out = torch.empty(sb + 4096, dtype=torch.uint8, device="cuda")
    idx_out = torch.empty(n_nals * 32, dtype=torch.uint8, device="cuda")
    dev_index = torch.from_numpy(got.view(np.uint8).copy()).cuda()
    for path in (0, 1, 2):                   # single pass (by items), three-step, arena tiles
        ctx.set_emit_path(path)
        out.zero_()
        ctx.emit_annexb_async(rbsp, ref_rb, dev_index, n_nals, 1, out, idx_out, summary)
        s = ctx.read_summary(summary)
        assert int(s["error"]) == 0 and int(s["stream_bytes"]) == sb, (path, s)
        device_equals_host(out, ref_stream, sb, "emit path %d: re-emitted stream" % path)
    ctx.set_emit_path(-1)
    ctx.close()
