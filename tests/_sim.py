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
        _lib.sim3_index_extract.argtypes = _lib.sim_index_extract.argtypes
        _lib.sim3_index_extract.restype = C.c_int
        _lib.sim4_index_extract.argtypes = _lib.sim_index_extract.argtypes
        _lib.sim4_index_extract.restype = C.c_int
        _lib.sim_index_extract_host.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64,
                                                C.c_void_p, C.c_uint64, C.c_void_p]
        _lib.sim_index_extract_host.restype = C.c_int
        _lib.sim_emit_annexb.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p]
        _lib.sim_emit_annexb.restype = C.c_int64
        _lib.sim_synth_rbsp.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]
        _lib.sim_synth_rbsp.restype = C.c_int64
        _lib.sim_dz_walk.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.sim_dz_walk.restype = None
    return _lib


def dz_walk(chunks, start_at, gaps):
    """k3_dense_tile's chunk algebra over len(chunks) / 16 chunks: (bytes that go in, count behind) for the entering counts 0, 1, 2"""
    chunks = np.ascontiguousarray(chunks, dtype=np.uint8)
    start_at = np.ascontiguousarray(start_at, dtype=np.int8)
    gaps = np.ascontiguousarray(gaps, dtype=np.uint32)
    assert len(chunks) % 16 == 0 and len(start_at) == len(chunks) // 16 == len(gaps)
    tot = np.zeros(3, dtype=np.uint32)
    st = np.zeros(3, dtype=np.uint32)
    lib().sim_dz_walk(chunks.ctypes.data, len(chunks) // 16, start_at.ctypes.data, gaps.ctypes.data, tot.ctypes.data, st.ctypes.data)
    return [int(x) for x in tot], [int(x) for x in st]


VARIANT = 2          # 2: LDS-image kernel logic (hbs_tile.h), 3: register-resident kernel logic (hbs_chunk.h)


def index_extract(stream, index_cap=None, want_rbsp=True, variant=None):
    stream = np.ascontiguousarray(stream, dtype=np.uint8)
    n = len(stream)
    # the device code reads the stream with guarded loads only; give the sim an exact-size buffer
    cap = (n // 3 + 2) if index_cap is None else index_cap
    idx = np.zeros(max(cap, 1), dtype=NAL_ENTRY)
    arena = np.full(n + 32, 0xAB, dtype=np.uint8)
    summ = np.zeros(1, dtype=SUMMARY)
    fn = {2: lib().sim_index_extract, 3: lib().sim3_index_extract, 4: lib().sim4_index_extract}[variant or VARIANT]
    rc = fn(stream.ctypes.data if n else None, n, idx.ctypes.data, cap,
                                 arena.ctypes.data if want_rbsp else None, n + 16, summ.ctypes.data)
    assert rc == 0, rc
    s = summ[0]
    assert (arena[n + 16:] == 0xAB).all()
    return idx[:int(s["nal_count"])].copy(), arena[:int(s["rbsp_bytes"])].copy(), s


def index_extract_windowed(stream, window_bytes, index_cap=None, want_rbsp=True):
    """hbs_ingest.h's driver over the CPU single-stepper (event-sparse tile logic per window)"""
    stream = np.ascontiguousarray(stream, dtype=np.uint8)
    n = len(stream)
    cap = (n // 3 + 2) if index_cap is None else index_cap
    idx = np.zeros(max(cap, 1), dtype=NAL_ENTRY)
    arena = np.full(n + 32, 0xAB, dtype=np.uint8)
    summ = np.zeros(1, dtype=SUMMARY)
    rc = lib().sim_index_extract_host(stream.ctypes.data if n else None, n, window_bytes, idx.ctypes.data, cap,
                                      arena.ctypes.data if want_rbsp else None, n + 16, summ.ctypes.data)
    s = summ[0]
    return rc, idx[:int(s["nal_count"])].copy(), arena[:int(s["rbsp_bytes"])].copy(), s


def emit_annexb(arena, idx, gap_mode=0):
    arena = np.ascontiguousarray(arena, dtype=np.uint8)
    idx = np.ascontiguousarray(idx)
    cap = len(arena) * 3 // 2 + 16 * len(idx) + int(idx["start"].max() if len(idx) else 0) + 64
    out = np.full(cap + 32, 0xAB, dtype=np.uint8)
    idx_out = np.zeros(max(len(idx), 1), dtype=NAL_ENTRY)
    pad = np.concatenate([arena, np.full(32, 0xCD, dtype=np.uint8)])
    n = lib().sim_emit_annexb(pad.ctypes.data, idx.ctypes.data, len(idx), gap_mode, out.ctypes.data, cap, idx_out.ctypes.data)
    assert n >= 0
    assert (out[cap:] == 0xAB).all() and (out[n:n + 16] == 0xAB).all()
    return out[:n].copy(), idx_out[:len(idx)]


def synth_rbsp(seed, n_nals, mode):
    arena = np.zeros(n_nals * 12289 + 16, dtype=np.uint8)
    idx = np.zeros(max(n_nals, 1), dtype=NAL_ENTRY)
    tot = lib().sim_synth_rbsp(seed, n_nals, mode, arena.ctypes.data, idx.ctypes.data)
    return arena[:tot].copy(), idx[:n_nals]


def parse_state(rbsp, idx):
    """the derived RPS tables behind the last NAL of the batch, as k4_state computes them (every row from its last writer):
    (int32 array in the layout of hbs::RpsTables, ok flag)"""
    L = lib()
    L.sim_rps_tables_bytes.restype = C.c_uint64
    L.sim_parse_set_state_out.argtypes = [C.c_void_p]
    out = np.zeros(int(L.sim_rps_tables_bytes()) // 4, dtype=np.int32)
    L.sim_parse_set_state_out(out.ctypes.data)
    try:
        parse_headers(rbsp, idx, fix=1)
    finally:
        L.sim_parse_set_state_out(None)
    return out, bool(L.sim_parse_state_ok())


def parse_headers(rbsp, idx, fix=0, stats=None):
    """K4 single-stepped on the CPU.  Returns (parsed ndarray[PARSED], struct arena uint8).
    fix: 0 the batch parse alone, 1 + the exact re-walk of the slices that need it when a slice raised the flag (what the
    library does), 2 + even when none did; stats: a list that receives [flag raised, slices walked again, chains too deep]."""
    from tests._parsecmp import PARSED
    L = lib()
    L.sim_parse_set_fix.argtypes = [C.c_int]
    L.sim_parse_set_fix(int(fix))
    L.sim_parse_headers.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]
    L.sim_parse_headers.restype = C.c_int64
    rbsp = np.ascontiguousarray(np.concatenate([rbsp, np.zeros(16, dtype=np.uint8)]))
    idx = np.ascontiguousarray(idx)
    n = len(idx)
    parsed = np.zeros(max(n, 1), dtype=PARSED)
    need = L.sim_parse_headers(rbsp.ctypes.data, idx.ctypes.data, n, parsed.ctypes.data, None, 0)
    structs = np.full(need + 64, 0xA5, dtype=np.uint8)
    got = L.sim_parse_headers(rbsp.ctypes.data, idx.ctypes.data, n, parsed.ctypes.data, structs.ctypes.data, need)
    assert got == need
    if stats is not None:
        st = (C.c_int * 3)()
        L.sim_parse_fix_stats(st)
        stats[:] = list(st)
    L.sim_parse_set_fix(0)
    return parsed[:n], structs[:need]


TRACE = np.dtype([("site", "<u4"), ("pos", "<u4"), ("value", "<i4")])


def parse_trace(rbsp, idx, cap=65536):
    """K4 single-stepped with the per-field trace: (parsed, structs, [records ndarray[TRACE] per NAL])"""
    from tests._parsecmp import PARSED
    L = lib()
    L.sim_parse_headers.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64]
    L.sim_parse_headers.restype = C.c_int64
    L.sim_parse_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64,
                                  C.c_void_p, C.c_uint32, C.c_void_p]
    L.sim_parse_trace.restype = C.c_int64
    rbsp = np.ascontiguousarray(np.concatenate([rbsp, np.zeros(16, dtype=np.uint8)]))
    idx = np.ascontiguousarray(idx)
    n = len(idx)
    parsed = np.zeros(max(n, 1), dtype=PARSED)
    need = L.sim_parse_headers(rbsp.ctypes.data, idx.ctypes.data, n, parsed.ctypes.data, None, 0)
    structs = np.zeros(need + 64, dtype=np.uint8)
    trace = np.zeros(max(n, 1) * cap, dtype=TRACE)
    count = np.zeros(max(n, 1), dtype=np.uint32)
    L.sim_parse_trace(rbsp.ctypes.data, idx.ctypes.data, n, parsed.ctypes.data, structs.ctypes.data, need,
                      trace.ctypes.data, cap, count.ctypes.data)
    return parsed[:n], structs[:need], [trace[k * cap:k * cap + min(int(count[k]), cap)].copy() for k in range(n)]


WRITTEN = np.dtype([("rc", "<i4"), ("rbsp_size", "<u4"), ("slice_data_size", "<i4"), ("pad", "<u4")])


def write_nal(nal_type, layer, tid, slot, sps_slot, pps, cap):
    """write_one_nal single-stepped: (result record, RBSP bytes written)"""
    L = lib()
    L.sim_write_nal.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.sim_write_nal.restype = C.c_int
    slot = np.ascontiguousarray(slot)
    out = np.zeros(cap + 16, dtype=np.uint8)
    res = np.zeros(1, dtype=WRITTEN)
    L.sim_write_nal(nal_type, layer, tid, slot.ctypes.data,
                    sps_slot.ctypes.data if sps_slot is not None else None, pps.ctypes.data if pps is not None else None,
                    out.ctypes.data, cap, res.ctypes.data)
    r = res[0]
    return r, out[:int(r["rbsp_size"])].copy()
