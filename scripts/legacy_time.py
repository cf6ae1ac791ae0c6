"""Per-call cost of the legacy single-NAL symbols (dev aid): find_nal_unit / nal_to_rbsp / read_hevc_nal_unit
through ctypes on a small parseable stream."""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, ".")
from hevcbitstream_amd.hevc_synth import stream_4k30
import hevcbitstream_amd.api as api
lib = C.CDLL(api.library_path())
u8p = C.POINTER(C.c_uint8)
lib.find_nal_unit.argtypes = [u8p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
lib.hevc_new.restype = C.c_void_p
lib.read_hevc_nal_unit.argtypes = [C.c_void_p, u8p, C.c_int]
lib.nal_to_rbsp.argtypes = [u8p, C.POINTER(C.c_int), u8p, C.POINTER(C.c_int)]
lib.rbsp_to_nal.argtypes = [u8p, C.POINTER(C.c_int), u8p, C.POINTER(C.c_int)]
lib.write_hevc_nal_unit.argtypes = [C.c_void_p, u8p, C.c_int]
stream, n = stream_4k30(11, n_pictures=120, slices_per_picture=8, idr_every=60, payload_bytes=(2000, 9000))
buf = np.frombuffer(stream, dtype=np.uint8).copy()
h = lib.hevc_new()
p, sz = 0, len(buf)
s, e = C.c_int(0), C.c_int(0)
t_find = t_read = t_rbsp = t_nal = t_write = 0.0
back = np.zeros(1 << 20, dtype=np.uint8)
wbuf = np.zeros(1 << 16, dtype=np.uint8)
cnt = 0
out = np.zeros(1 << 20, dtype=np.uint8)
base = buf.ctypes.data
while True:
    t0 = time.perf_counter()
    r = lib.find_nal_unit(C.cast(base + p, u8p), sz - p, C.byref(s), C.byref(e))
    t1 = time.perf_counter()
    if r <= 0:
        break
    ns, rs = C.c_int(e.value - s.value), C.c_int(len(out))
    lib.nal_to_rbsp(C.cast(base + p + s.value, u8p), C.byref(ns), out.ctypes.data_as(u8p), C.byref(rs))
    t2 = time.perf_counter()
    rc = lib.read_hevc_nal_unit(h, C.cast(base + p + s.value, u8p), e.value - s.value)
    t3 = time.perf_counter()
    bs = C.c_int(len(back))
    lib.rbsp_to_nal(out.ctypes.data_as(u8p), C.byref(rs), back.ctypes.data_as(u8p), C.byref(bs))
    t4 = time.perf_counter()
    lib.write_hevc_nal_unit(h, wbuf.ctypes.data_as(u8p), len(wbuf))
    t5 = time.perf_counter()
    if cnt >= 20:                      # skip warm-up (context creation, first launches)
        t_find += t1 - t0; t_rbsp += t2 - t1; t_read += t3 - t2; t_nal += t4 - t3; t_write += t5 - t4
    cnt += 1
    p += e.value
m = cnt - 20
print("NALs %d (%.1f KiB avg): find_nal_unit %.0f us, nal_to_rbsp %.0f us, read_hevc_nal_unit %.0f us, rbsp_to_nal %.0f us, write_hevc_nal_unit %.0f us per call"
      % (cnt, len(buf) / cnt / 1024, t_find / m * 1e6, t_rbsp / m * 1e6, t_read / m * 1e6, t_nal / m * 1e6, t_write / m * 1e6))
