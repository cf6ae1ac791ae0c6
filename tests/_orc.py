"""ctypes access to the test oracle (oracle/liboracle.so) and, when it has been
built in the dev container, to the REAL reference (oracle/_ref/libhevcref.so).

Test infrastructure only: nothing in hevcbitstream_amd/ imports this module."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_DIR = os.path.join(ROOT, "oracle")

NAL_ENTRY = np.dtype([("start", "<u8"), ("end", "<u8"), ("rbsp_off", "<u8"),
                      ("rbsp_len", "<u4"), ("status", "<i4")])
ST_ERROR, ST_TRAILING03, ST_UNTERMINATED = 1, 2, 4

_u8p = C.POINTER(C.c_uint8)
_ip = C.POINTER(C.c_int)


def _ptr(a):
    return a.ctypes.data_as(_u8p)


def build_oracle():
    so = os.path.join(ORC_DIR, "liboracle.so")
    srcs = [os.path.join(ORC_DIR, f) for f in os.listdir(ORC_DIR) if f.endswith((".c", ".h"))]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-s", "-C", ORC_DIR, "liboracle.so"])
    return so


class _L2:
    """find_nal_unit / nal_to_rbsp / rbsp_to_nal over numpy byte arrays; the same
    wrapper drives the oracle (prefix 'orc_') and the reference (no prefix)."""

    def __init__(self, lib, prefix):
        self.lib = lib
        self.f_find = getattr(lib, prefix + "find_nal_unit")
        self.f_find.argtypes = [_u8p, C.c_int, _ip, _ip]
        self.f_find.restype = C.c_int
        self.f_n2r = getattr(lib, prefix + "nal_to_rbsp")
        self.f_n2r.argtypes = [_u8p, _ip, _u8p, _ip]
        self.f_n2r.restype = C.c_int
        self.f_r2n = getattr(lib, prefix + "rbsp_to_nal")
        self.f_r2n.argtypes = [_u8p, _ip, _u8p, _ip]
        self.f_r2n.restype = C.c_int

    def find_nal_unit(self, data, size=None, pad=8):
        """data: bytes; the buffer handed over is followed by `pad` 0xFF bytes
        (the reference reads up to 3 bytes past `size`, see hbs_oracle.h)."""
        size = len(data) if size is None else size
        buf = np.frombuffer(bytes(data[:size]) + b"\xff" * pad, dtype=np.uint8).copy()
        s, e = C.c_int(-7), C.c_int(-7)
        r = self.f_find(_ptr(buf), size, C.byref(s), C.byref(e))
        return r, s.value, e.value

    def nal_to_rbsp(self, nal):
        n = len(nal)
        src = np.frombuffer(bytes(nal) + b"\xff" * 4, dtype=np.uint8).copy()
        dst = np.full(n + 8, 0xEE, dtype=np.uint8)
        ns, rs = C.c_int(n), C.c_int(n)
        r = self.f_n2r(_ptr(src), C.byref(ns), _ptr(dst), C.byref(rs))
        return r, ns.value, rs.value, (bytes(dst[:r]) if r >= 0 else None)

    def rbsp_to_nal(self, rbsp):
        n = len(rbsp)
        src = np.frombuffer(bytes(rbsp) + b"\xff" * 4, dtype=np.uint8).copy()
        dst = np.full(n * 3 // 2 + 8, 0xEE, dtype=np.uint8)
        rs, ns = C.c_int(n), C.c_int(len(dst))
        r = self.f_r2n(_ptr(src), C.byref(rs), _ptr(dst), C.byref(ns))
        return r, ns.value, bytes(dst[:r])


class Oracle(_L2):
    def __init__(self):
        lib = C.CDLL(build_oracle())
        super().__init__(lib, "orc_")
        lib.orc_index_stream.argtypes = [_u8p, C.c_int64, C.c_void_p, C.c_int64, _ip]
        lib.orc_index_stream.restype = C.c_int64
        lib.orc_extract_rbsp.argtypes = [_u8p, C.c_void_p, C.c_int64, _u8p, C.c_int64]
        lib.orc_extract_rbsp.restype = C.c_int64
        lib.orc_emit_annexb.argtypes = [_u8p, C.c_void_p, C.c_int64, _u8p, C.c_int64]
        lib.orc_emit_annexb.restype = C.c_int64
        lib.orc_gen_stream.argtypes = [C.c_uint64, C.c_int64, C.c_int, _u8p, C.c_int64, C.c_void_p, _u8p]
        lib.orc_gen_stream.restype = C.c_int64
        lib.orc_gen_rbsp_len.argtypes = [C.c_uint64, C.c_uint64]
        lib.orc_gen_rbsp_len.restype = C.c_uint32

    def index_stream(self, stream):
        """stream: uint8 ndarray.  Returns (entries[NAL_ENTRY], stop_reason)."""
        stream = np.ascontiguousarray(stream, dtype=np.uint8)
        cap = len(stream) // 4 + 2
        out = np.zeros(cap, dtype=NAL_ENTRY)
        why = C.c_int(0)
        n = self.lib.orc_index_stream(_ptr(stream), len(stream), out.ctypes.data, cap, C.byref(why))
        return out[:n].copy(), why.value

    def index_extract(self, stream):
        """Index + RBSP arena.  Returns (entries, arena, stop_reason)."""
        stream = np.ascontiguousarray(stream, dtype=np.uint8)
        idx, why = self.index_stream(stream)
        arena = np.zeros(len(stream) + 16, dtype=np.uint8)
        tot = self.lib.orc_extract_rbsp(_ptr(stream), idx.ctypes.data, len(idx), _ptr(arena), len(arena))
        assert tot >= 0
        return idx, arena[:tot].copy(), why

    def emit_annexb(self, arena, idx):
        arena = np.ascontiguousarray(arena, dtype=np.uint8)
        idx = np.ascontiguousarray(idx)
        cap = int(len(arena) * 3 // 2 + 8 * len(idx) + int(idx["start"][-1] if len(idx) else 0) + 64)
        out = np.zeros(cap, dtype=np.uint8)
        n = self.lib.orc_emit_annexb(_ptr(arena), idx.ctypes.data, len(idx), _ptr(out), cap)
        assert n >= 0, n
        return out[:n].copy()

    def gen_stream(self, seed, n_nals, mode=0, want_arena=True):
        """Synthetic stream S(seed, n_nals, mode).  Returns (stream, entries, arena)."""
        cap = n_nals * (12289 * 3 // 2 + 8) + 64
        out = np.zeros(cap, dtype=np.uint8)
        idx = np.zeros(n_nals, dtype=NAL_ENTRY)
        arena = np.zeros(n_nals * 12289 + 16, dtype=np.uint8) if want_arena else None
        n = self.lib.orc_gen_stream(seed, n_nals, mode, _ptr(out), cap, idx.ctypes.data,
                                    _ptr(arena) if want_arena else None)
        assert n >= 0
        tot = int(idx["rbsp_off"][-1] + idx["rbsp_len"][-1]) if n_nals else 0
        return out[:n].copy(), idx, (arena[:tot].copy() if want_arena else None)


class Reference(_L2):
    def __init__(self, path):
        super().__init__(C.CDLL(path), "")


_oracle = None
_reference = False


def oracle():
    global _oracle
    if _oracle is None:
        _oracle = Oracle()
    return _oracle


def reference():
    global _reference
    if _reference is False:
        p = os.path.join(ORC_DIR, "_ref", "libhevcref.so")
        _reference = Reference(p) if os.path.exists(p) else None
    return _reference


# ---- HEVC header layer ---------------------------------------------------------------

import json as _json

_LAYOUT = None
STRUCTS = ("nal", "vps", "sps", "pps", "aud", "sh")          # pointer order in hevc_stream_t
STRUCT_TYPES = {"nal": "hevc_nal_t", "vps": "hevc_vps_t", "sps": "hevc_sps_t", "pps": "hevc_pps_t",
                "sh": "hevc_slice_header_t"}


def layout():
    global _LAYOUT
    if _LAYOUT is None:
        _LAYOUT = _json.load(open(os.path.join(ROOT, "tests", "golden", "field_layout.json")))
    return _LAYOUT


def flat_fields(type_name, prefix="", base=0):
    """[(dotted name, int32 index, count)] of every int member, nested structs expanded."""
    out = []
    for f in layout()[type_name]["fields"]:
        if f is None:
            continue
        name, off, dims, sub = f
        if sub is None:
            cnt = int(np.prod(dims)) if dims else 1
            out.append((prefix + name, (base + off) // 4, cnt))
        else:
            n = dims[0] if dims else 1
            size = layout()[sub]["size"]
            for i in range(n):
                tag = "%s%s[%d]." % (prefix, name, i) if dims else "%s%s." % (prefix, name)
                out.extend(flat_fields(sub, tag, base + off + i * size))
    return out


class _HevcParser:
    """hevc_stream_t-shaped parser object of either library, viewed as int32 arrays."""

    def _views(self, hptr):
        ptrs = (C.c_void_p * 7).from_address(hptr)       # nal vps sps pps aud sh slice_data
        self.v = {}
        for i, nm in enumerate(STRUCTS):
            if nm in STRUCT_TYPES:
                size = layout()[STRUCT_TYPES[nm]]["size"]
                self.v[nm] = np.ctypeslib.as_array((C.c_int32 * (size // 4)).from_address(ptrs[i]))
        self._slice_data = ptrs[6]

    def slice_data(self):
        size = C.c_int.from_address(self._slice_data).value
        ptr = C.c_void_p.from_address(self._slice_data + 8).value
        if size < 0 or not ptr:
            return size, None
        return size, bytes((C.c_uint8 * size).from_address(ptr))

    def snapshot(self):
        return {k: a.copy() for k, a in self.v.items()}


class OracleHevc(_HevcParser):
    def __init__(self):
        o = oracle()
        L = o.lib
        L.orc_hevc_new.restype = C.c_void_p
        L.orc_hevc_free.argtypes = [C.c_void_p]
        L.orc_hevc_stream_ptr.argtypes = [C.c_void_p]
        L.orc_hevc_stream_ptr.restype = C.c_void_p
        L.orc_read_hevc_nal_unit.argtypes = [C.c_void_p, _u8p, C.c_int]
        self.L = L
        self.o = L.orc_hevc_new()
        self._views(L.orc_hevc_stream_ptr(self.o))

    def read(self, nal):
        buf = np.frombuffer(bytes(nal) + b"\xff" * 8, dtype=np.uint8).copy()
        return self.L.orc_read_hevc_nal_unit(self.o, _ptr(buf), len(nal))

    def tables(self):
        """the derived RPS tables as they stand (hevc_stream.c:26-32): int32[3 * 32 + 4 * 32 * 32], the layout of hbs::RpsTables"""
        self.L.orc_hevc_tables.argtypes = [C.c_void_p]
        self.L.orc_hevc_tables.restype = C.POINTER(C.c_int32)
        n = 3 * 32 + 4 * 32 * 32
        return np.ctypeslib.as_array(self.L.orc_hevc_tables(self.o), shape=(n,)).copy()

    def close(self):
        self.L.orc_hevc_free(self.o)


class ReferenceHevc(_HevcParser):
    def __init__(self):
        r = reference()
        assert r is not None
        L = r.lib
        L.hevc_new.restype = C.c_void_p
        L.hevc_free.argtypes = [C.c_void_p]
        L.read_hevc_nal_unit.argtypes = [C.c_void_p, _u8p, C.c_int]
        L.write_hevc_nal_unit.argtypes = [C.c_void_p, _u8p, C.c_int]
        self.L = L
        self.h = L.hevc_new()
        self._views(self.h)

    def read(self, nal):
        buf = np.frombuffer(bytes(nal) + b"\xff" * 8, dtype=np.uint8).copy()
        return self.L.read_hevc_nal_unit(self.h, _ptr(buf), len(nal))

    def close(self):
        pass        # the reference leaks slice_data->rbsp_buf; keep the object alive instead
