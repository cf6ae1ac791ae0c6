"""The reference's own loop over a host buffer and its way back, shared by tests/test_gpu_fullsize.py and the checking legs of
bench.py (cpu_baseline*): find_nal_unit + nal_to_rbsp per NAL (hevc_analyze.c:135-177, h264_nal.c:38-200) and rbsp_to_nal
(h264_nal.c:92-132) through the compiled reference (oracle/_ref/libref_driver.so) when its prebuilt copy travelled with the
tree, else through the oracle's restatement.  TEST INFRASTRUCTURE: never imported by the product."""
import ctypes as C
import os

import numpy as np

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
