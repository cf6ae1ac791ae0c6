"""ctypes access to tests/sim/libhbs_sim.so: the product's per-tile device logic
single-stepped on the CPU (tests only; see tests/sim/hbs_sim.cpp)."""
import ctypes as C
import os
import subprocess

import numpy as np

from tests._orc import NAL_ENTRY

HERE = os.path.dirname(os.path.abspath(__file__))
SIM_DIR = os.path.join(HERE, "sim")

SUMMARY = np.dtype([("nal_count", "<u8"), ("nal_found", "<u8"), ("rbsp_bytes", "<u8"),
                    ("stream_bytes", "<u8"), ("stop_reason", "<i4"), ("error", "<i4"),
                    ("reserved", "<u8", (3,))])
_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-s", "-C", SIM_DIR])
        _lib = C.CDLL(os.path.join(SIM_DIR, "libhbs_sim.so"))
        _lib.sim_index_extract.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                           C.c_void_p, C.c_uint64, C.c_void_p]
        _lib.sim_index_extract.restype = C.c_int
    return _lib


def index_extract(stream, index_cap=None, want_rbsp=True):
    stream = np.ascontiguousarray(stream, dtype=np.uint8)
    n = len(stream)
    # the device code reads the stream with guarded loads only; give the sim an exact-size buffer
    cap = (n // 3 + 2) if index_cap is None else index_cap
    idx = np.zeros(max(cap, 1), dtype=NAL_ENTRY)
    arena = np.full(n + 32, 0xAB, dtype=np.uint8)
    summ = np.zeros(1, dtype=SUMMARY)
    rc = lib().sim_index_extract(stream.ctypes.data if n else None, n, idx.ctypes.data, cap,
                                 arena.ctypes.data if want_rbsp else None, n + 16, summ.ctypes.data)
    assert rc == 0, rc
    s = summ[0]
    assert (arena[n + 16:] == 0xAB).all()
    return idx[:int(s["nal_count"])].copy(), arena[:int(s["rbsp_bytes"])].copy(), s
