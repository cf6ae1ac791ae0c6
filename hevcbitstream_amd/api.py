"""ctypes binding of include/hevcbitstream_amd.h over torch device tensors."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# layout of hbs_nal_entry / hbs_summary (include/hevcbitstream_amd.h)
NAL_ENTRY = np.dtype([("start", "<u8"), ("end", "<u8"), ("rbsp_off", "<u8"),
                      ("rbsp_len", "<u4"), ("status", "<i4")])
SUMMARY = np.dtype([("nal_count", "<u8"), ("nal_found", "<u8"), ("rbsp_bytes", "<u8"),
                    ("stream_bytes", "<u8"), ("stop_reason", "<i4"), ("error", "<i4"),
                    ("reserved", "<u8", (3,))])
ST_ERROR, ST_TRAILING03, ST_UNTERMINATED = 1, 2, 4
# layout of hbs_parsed_nal
WRITTEN = np.dtype([("rc", "<i4"), ("rbsp_size", "<u4"), ("slice_data_size", "<i4"), ("pad", "<u4")])
PARSED = np.dtype([("rc", "<i4"), ("nal_unit_type", "<i4"), ("nal_layer_id", "<i4"), ("nal_temporal_id_plus1", "<i4"),
                   ("struct_off", "<u8"), ("slice_data_size", "<i4"), ("slice_data_off", "<u4")])

# layout of hbs_slice_compact: sixteen members of hevc_slice_header_t, by name
COMPACT_FIELDS = ("first_slice_segment_in_pic_flag", "no_output_of_prior_pics_flag", "pic_parameter_set_id", "dependent_slice_segment_flag",
                  "slice_segment_address", "slice_type", "pic_output_flag", "slice_pic_order_cnt_lsb",
                  "short_term_ref_pic_set_sps_flag", "short_term_ref_pic_set_idx", "num_long_term_pics", "slice_temporal_mvp_enabled_flag",
                  "num_ref_idx_l0_active_minus1", "num_ref_idx_l1_active_minus1", "slice_qp_delta", "num_entry_point_offsets")
COMPACT = np.dtype([(f, "<i4") for f in COMPACT_FIELDS])

SEI_MAX_MESSAGES = 6
EXT_NAL = np.dtype([("num_sei_messages", "<i4"), ("primary_pic_type", "<i4"), ("filler_bytes", "<u4"), ("reserved", "<u4"),
                    ("sei", [("payloadType", "<i4"), ("payloadSize", "<i4"), ("payload_off", "<u4"), ("reserved", "<u4")], (SEI_MAX_MESSAGES,))])

EXPORTS = ["hbs_version", "hbs_ctx_create", "hbs_ctx_destroy", "hbs_ctx_set_stream", "hbs_ctx_use_own_stream",
           "hbs_ctx_get_stream",
           "hbs_ctx_synchronize", "hbs_last_error", "hbs_index_extract", "hbs_index_extract_host", "hbs_read_summary",
           "hbs_workspace_bytes", "hbs_write_headers", "hbs_parse_headers_trace", "hbs_emit_annexb", "hbs_annexb_bound", "hbs_synth_rbsp",
           "hbs_synth_rbsp_bound", "hbs_ctx_enable_timing", "hbs_ctx_kernel_ms", "hbs_ctx_kernel_ms_back", "hbs_ctx_grid",
           "hbs_parse_headers", "hbs_ctx_set_kernel", "hbs_ctx_set_count_ahead", "hbs_ctx_get_kernel", "hbs_ctx_last_kernel",
           "hbs_host_alloc", "hbs_host_free", "hbs_copy_to_device_async", "hbs_copy_device",
           "hbs_ctx_set_sequential_parse", "hbs_ctx_set_emit_path", "hbs_parse_extended",
           "hbs_comm_unique_id", "hbs_comm_create", "hbs_comm_adopt", "hbs_comm_destroy", "hbs_comm_rank", "hbs_comm_world", "hbs_comm_reserve_hint", "hbs_parse_headers_compact", "hbs_parse_materialize", "hbs_index_parse_compact", "hbs_gather_parts", "hbs_index_parse", "hbs_ctx_reserve_workgroups",
           "hbs_gather_index", "hbs_ctx_device", "hbs_find_cut_host", "hbs_trim_part", "hbs_annexb_bound_gaps", "hbs_ctx_device_bytes", "hbs_ctx_set_ingest_window_max", "hbs_pair_alloc", "hbs_pair_free", "hbs_pair_pool_trim", "hbs_pair_pool_stats", "hbs_parse_headers_state", "hbs_ctx_last_emit_by_tiles", "hbs_ctx_set_device_exclusive"]


PAIR_REPORT = np.dtype([("chunks", "<u4"), ("probed", "<u4"), ("rejected", "<u4"), ("accepted_fast", "<u4"),
                        ("unprobed_after_budget", "<u4"), ("from_pool", "<u4"), ("from_table", "<u4"), ("reserved", "<u4")])


class _PairedMemory:
    """memory from hbs_pair_alloc, handed to torch through __cuda_array_interface__; given back when the last tensor on it dies"""

    def __init__(self, lib, ptr, nbytes):
        self.lib, self.ptr = lib, ptr
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}

    def __del__(self):
        try:
            if self.ptr:
                self.lib.hbs_pair_free(None, C.c_void_p(self.ptr))
                self.ptr = 0
        except Exception:
            pass


class HbsError(RuntimeError):
    pass


def library_path():
    # HBS_LIB: a development build of the same library (make variant NAME=...), for A/B timing only
    return os.environ.get("HBS_LIB") or os.path.join(_HERE, "libhevcbitstream_amd.so")


def source_digest():
    """sha256 over the kernel sources (csrc/*.hip, *.h, *.c, sorted by name): profiles record it, and bench.py quotes a
    profile's counter figures only while it still matches (a changed kernel must be profiled again)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(_HERE, "csrc", "*"))):
        if f.endswith((".hip", ".h", ".c")):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()


_lib = None


class _DevLibrary:
    """HBS_LIB only (A/B timing against a development build, possibly an older round's): an entry point the build lacks
    becomes a stub that raises when CALLED, instead of failing the load.  Never used for the shipped library."""

    def __init__(self, lib):
        self.__dict__["_lib"] = lib

    def __getattr__(self, name):
        try:
            return getattr(self._lib, name)
        except AttributeError:
            class _Missing:
                argtypes = None
                restype = None

                def __call__(self, *a):
                    raise HbsError("%s: not in the development build %s" % (name, os.environ["HBS_LIB"]))
            m = _Missing()
            self.__dict__[name] = m
            return m


def load_library():
    """Load the gfx950 library.  Raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    p = library_path()
    if not os.path.exists(p):
        raise HbsError("%s not built: run `make lib` (hipcc --offload-arch=gfx950); "
                       "there is no CPU fallback" % p)
    lib = C.CDLL(p)
    if os.environ.get("HBS_LIB"):
        lib = _DevLibrary(lib)
    lib.hbs_version.restype = C.c_char_p
    lib.hbs_ctx_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    lib.hbs_ctx_destroy.argtypes = [C.c_void_p]
    lib.hbs_ctx_destroy.restype = None
    lib.hbs_ctx_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.hbs_ctx_use_own_stream.argtypes = [C.c_void_p]
    lib.hbs_ctx_get_stream.argtypes = [C.c_void_p]
    lib.hbs_ctx_get_stream.restype = C.c_void_p
    lib.hbs_ctx_synchronize.argtypes = [C.c_void_p]
    lib.hbs_ctx_enable_timing.argtypes = [C.c_void_p, C.c_int]
    lib.hbs_ctx_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    lib.hbs_ctx_kernel_ms_back.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float)]
    lib.hbs_ctx_grid.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.hbs_ctx_set_kernel.argtypes = [C.c_void_p, C.c_int]
    lib.hbs_ctx_get_kernel.argtypes = [C.c_void_p]
    lib.hbs_ctx_last_kernel.argtypes = [C.c_void_p]
    lib.hbs_last_error.argtypes = [C.c_void_p]
    lib.hbs_last_error.restype = C.c_char_p
    lib.hbs_index_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                      C.c_void_p, C.c_uint64, C.c_void_p]
    lib.hbs_index_extract_host.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_uint64,
                                           C.c_void_p, C.c_uint64, C.c_void_p]
    lib.hbs_write_headers.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_uint32, C.c_void_p]
    lib.hbs_read_summary.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.hbs_workspace_bytes.argtypes = [C.c_uint64]
    lib.hbs_workspace_bytes.restype = C.c_uint64
    lib.hbs_emit_annexb.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int,
                                    C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.hbs_annexb_bound.argtypes = [C.c_uint64, C.c_uint64]
    lib.hbs_annexb_bound.restype = C.c_uint64
    lib.hbs_synth_rbsp.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64,
                                   C.c_void_p, C.c_void_p]
    lib.hbs_parse_headers.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                      C.c_uint64, C.c_void_p]
    lib.hbs_synth_rbsp_bound.argtypes = [C.c_uint64]
    lib.hbs_synth_rbsp_bound.restype = C.c_uint64
    lib.hbs_pair_alloc.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p), C.c_void_p]
    lib.hbs_pair_free.argtypes = [C.c_void_p, C.c_void_p]
    lib.hbs_ctx_last_emit_by_tiles.argtypes = [C.c_void_p]
    _lib = lib
    return lib


class Context:
    """One per GPU (per rank).  Device buffers are torch uint8 CUDA tensors; the
    library runs on torch's current stream so that ordering with torch ops is
    the stream order."""

    def __init__(self, device=0):
        import torch
        self.torch = torch
        self.lib = load_library()
        self.device = int(device)
        h = C.c_void_p()
        rc = self.lib.hbs_ctx_create(C.byref(h), self.device)
        if rc != 0:
            raise HbsError("hbs_ctx_create(device=%d) failed: %d (no gfx950 GPU? there is no CPU fallback)"
                           % (self.device, rc))
        self.h = h
        self._bind_stream()

    def _bind_stream(self):
        s = self.torch.cuda.current_stream(self.device).cuda_stream
        self.lib.hbs_ctx_set_stream(self.h, C.c_void_p(s))

    def close(self):
        if getattr(self, "h", None):
            self.lib.hbs_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise HbsError("%s failed: %d (%s)" % (what, rc, self.lib.hbs_last_error(self.h).decode()))

    def set_sequential_parse(self, on=True):
        """parse batches NAL after NAL with ONE set of derived RPS tables, as the reference does (slow, exact on any input)"""
        self.lib.hbs_ctx_set_sequential_parse.argtypes = [C.c_void_p, C.c_int]
        self._check(self.lib.hbs_ctx_set_sequential_parse(self.h, 1 if on else 0), "hbs_ctx_set_sequential_parse")

    def set_emit_path(self, path=-1):
        """-1 = picked per call (default), 0 = the single-pass emit kernel by NALs, 1 = count / scan / emit, 2 = the single pass by
        arena tiles whenever the index allows it"""
        self.lib.hbs_ctx_set_emit_path.argtypes = [C.c_void_p, C.c_int]
        self._check(self.lib.hbs_ctx_set_emit_path(self.h, path), "hbs_ctx_set_emit_path")

    def set_kernel(self, variant):
        """0 = automatic (density probe picks 4 or 2 on the device; the default), 2 = LDS-image
        scan/extract kernel, 4 = event-sparse one, 5 = index-only streaming one"""
        self._check(self.lib.hbs_ctx_set_kernel(self.h, variant), "hbs_ctx_set_kernel")

    def set_count_ahead(self, mode=1):
        """kernel 4's dense tiles counted ahead of it: 0 never, 1 on streams of 3 GiB and more (default), 2 always"""
        self.lib.hbs_ctx_set_count_ahead.argtypes = [C.c_void_p, C.c_int]     # (bound here: development libraries of earlier rounds load too)
        self._check(self.lib.hbs_ctx_set_count_ahead(self.h, mode), "hbs_ctx_set_count_ahead")

    def kernel(self):
        return self.lib.hbs_ctx_get_kernel(self.h)

    def last_kernel(self):
        """the kernel that ran the last index_extract (waits for it)"""
        return self.lib.hbs_ctx_last_kernel(self.h)

    def enable_timing(self, on=True):
        self._check(self.lib.hbs_ctx_enable_timing(self.h, 1 if on else 0), "hbs_ctx_enable_timing")

    def kernel_ms(self):
        """Duration of the last fused scan/extract kernel (HIP events on its own stream)."""
        ms = C.c_float()
        self._check(self.lib.hbs_ctx_kernel_ms(self.h, C.byref(ms)), "hbs_ctx_kernel_ms")
        return ms.value

    def reserve_workgroups(self, spare):
        """leave `spare` workgroup slots of the persistent scan kernels free (for RCCL's kernels beside the scan)"""
        self.lib.hbs_ctx_reserve_workgroups.argtypes = [C.c_void_p, C.c_int]
        self._check(self.lib.hbs_ctx_reserve_workgroups(self.h, int(spare)), "hbs_ctx_reserve_workgroups")

    def set_device_exclusive(self, on):
        """1: this context's calls are the only persistent kernels on the device while they run (first tiles by workgroup
        number); 0 (default): every tile by ticket -- safe with several contexts or processes on one device"""
        self.lib.hbs_ctx_set_device_exclusive.argtypes = [C.c_void_p, C.c_int]
        self._check(self.lib.hbs_ctx_set_device_exclusive(self.h, int(on)), "hbs_ctx_set_device_exclusive")
        self.exclusive = int(on)

    def device_bytes(self):
        """device memory the context itself holds (grow-only scratch; caller-owned buffers are not counted)"""
        self.lib.hbs_ctx_device_bytes.restype = C.c_uint64
        self.lib.hbs_ctx_device_bytes.argtypes = [C.c_void_p]
        return int(self.lib.hbs_ctx_device_bytes(self.h))

    def kernel_ms_back(self, back):
        """Duration of the timed call `back` calls ago (0 = the last one; the library keeps the last 64)."""
        ms = C.c_float()
        self._check(self.lib.hbs_ctx_kernel_ms_back(self.h, back, C.byref(ms)), "hbs_ctx_kernel_ms_back")
        return ms.value

    def grid(self):
        a, b = C.c_int(), C.c_int()
        self.lib.hbs_ctx_grid(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def pair_alloc(self, peer, nbytes):
        """hbs_pair_alloc: `nbytes` of device memory placed against the tensor `peer` (the buffer it will be written from /
        read into), as a uint8 torch tensor that frees the memory when it dies.  Returns (tensor, report dict).  The
        probe runs on the current torch stream; `peer`'s contents do not matter."""
        t = self.torch
        self._bind_stream()
        self.lib.hbs_pair_alloc.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_void_p), C.c_void_p]
        self.lib.hbs_pair_free.argtypes = [C.c_void_p, C.c_void_p]
        rep = np.zeros(1, dtype=PAIR_REPORT)
        ptr = C.c_void_p()
        rc = self.lib.hbs_pair_alloc(self.h, C.c_void_p(peer.data_ptr()) if peer is not None else None,
                                     peer.numel() * peer.element_size() if peer is not None else 0, nbytes, C.byref(ptr), rep.ctypes.data)
        self._check(rc, "hbs_pair_alloc")
        holder = _PairedMemory(self.lib, ptr.value, nbytes)
        tensor = t.as_tensor(holder, device=t.device("cuda", self.device))       # zero-copy: torch keeps `holder` alive
        assert tensor.data_ptr() == ptr.value
        return tensor, {k: int(rep[0][k]) for k in PAIR_REPORT.names if k != "reserved"}

    def alloc_outputs(self, stream_bytes, index_cap=None, want_rbsp=True, peer=None):
        """Device buffers sized for a stream: (index[u8, cap*32], rbsp[u8] or None, summary[u8, 64]).
        peer: the stream tensor the arena will be written from -- the arena is then placed against it (pair_alloc:
        4-5 % on multi-GiB streams; self.last_pair_report says what the probe found)."""
        t = self.torch
        dev = t.device("cuda", self.device)
        if index_cap is None:
            index_cap = self.default_index_cap(stream_bytes)
        index = t.empty(max(index_cap, 1) * NAL_ENTRY.itemsize, dtype=t.uint8, device=dev)
        if want_rbsp and peer is not None:
            rbsp, self.last_pair_report = self.pair_alloc(peer, stream_bytes + 16)
        else:
            rbsp = t.empty(stream_bytes + 16, dtype=t.uint8, device=dev) if want_rbsp else None
        summary = t.zeros(SUMMARY.itemsize, dtype=t.uint8, device=dev)
        return index, rbsp, summary, index_cap

    @staticmethod
    def default_index_cap(stream_bytes):
        """Entries to provide when the caller does not say.  The worst case -- a start code every three bytes -- is
        stream_bytes / 3 entries of 32 bytes, ten times the stream, all of it cleared by every scan: fine for small
        streams (tests full of tiny NALs), ruinous for a 1 GiB one.  From 64 MiB on: one entry per 64 stream bytes (coded
        video has one per several KiB); a stream with more NALs than that reports HBS_E_CAPACITY in its summary and
        nal_found says how many entries it needs (index_extract() below then runs it again with that many)."""
        worst = stream_bytes // 3 + 2
        return worst if stream_bytes <= (64 << 20) else min(worst, stream_bytes // 64 + 4096)

    def index_extract_async(self, stream, index, index_cap, rbsp, summary):
        """Enqueue K12 on the current torch stream.  All arguments are device tensors.  A stream with more NALs than
        index_cap reports HBS_E_CAPACITY in its summary (nal_found = the entries it needs): alloc_outputs() defaults to
        default_index_cap(), which is NOT the worst case above 64 MiB -- index_extract() retries with a larger index,
        callers of this method do that themselves."""
        self._bind_stream()
        rc = self.lib.hbs_index_extract(self.h, C.c_void_p(stream.data_ptr() if stream.numel() else None),
                                        stream.numel(), C.c_void_p(index.data_ptr()), index_cap,
                                        C.c_void_p(rbsp.data_ptr()) if rbsp is not None else None,
                                        rbsp.numel() if rbsp is not None else 0,
                                        C.c_void_p(summary.data_ptr()))
        self._check(rc, "hbs_index_extract")

    def read_summary(self, summary):
        out = np.zeros(1, dtype=SUMMARY)
        rc = self.lib.hbs_read_summary(self.h, C.c_void_p(summary.data_ptr()), out.ctypes.data)
        self._check(rc, "hbs_read_summary")
        return out[0]

    def index_extract(self, stream, index_cap=None, want_rbsp=True):
        """Convenience: run K12 and bring the results to the host.
        Returns (entries ndarray[NAL_ENTRY], arena ndarray[u8] or None, summary record)."""
        index, rbsp, summary, cap = self.alloc_outputs(stream.numel(), index_cap, want_rbsp)
        self.index_extract_async(stream, index, cap, rbsp, summary)
        s = self.read_summary(summary)
        if index_cap is None and int(s["error"]) == -4 and int(s["nal_found"]) > cap:      # HBS_E_CAPACITY: the default was too small
            # only the index grows: the arena and the summary of the first attempt are used again (a second arena next to the
            # first would double the peak for exactly the large streams the reduced default is for)
            del index
            cap = int(s["nal_found"]) + 8
            index = self.torch.empty(cap * NAL_ENTRY.itemsize, dtype=self.torch.uint8, device=summary.device)
            summary.zero_()
            self.index_extract_async(stream, index, cap, rbsp, summary)
            s = self.read_summary(summary)
        n = int(s["nal_count"])
        ent = index[: n * NAL_ENTRY.itemsize].cpu().numpy().view(NAL_ENTRY).copy()
        arena = rbsp[: int(s["rbsp_bytes"])].cpu().numpy() if want_rbsp else None
        return ent, arena, s

    # ---- K3 and the synthetic workload -------------------------------------------------

    def set_ingest_window_max(self, max_bytes):
        """ceiling of the window growth of index_extract_host (a NAL longer than the window doubles it); 0: the default, 1 GiB"""
        self.lib.hbs_ctx_set_ingest_window_max.argtypes = [C.c_void_p, C.c_uint64]
        self._check(self.lib.hbs_ctx_set_ingest_window_max(self.h, int(max_bytes)), "hbs_ctx_set_ingest_window_max")

    def index_extract_host(self, stream, window_bytes=256 << 20, index_cap=None, want_rbsp=True, pinned=True):
        """Windowed ingest of a HOST stream of any length (hbs_index_extract_host): `stream` is a numpy
        uint8 array (or anything np.asarray accepts).  Returns (entries ndarray[NAL_ENTRY], arena ndarray
        or None, summary record).  pinned=True stages the stream and the outputs in page-locked memory so
        that uploads, scans and downloads overlap."""
        t = self.torch
        a = np.ascontiguousarray(np.asarray(stream, dtype=np.uint8))
        n = int(a.size)
        cap = (n // 3 + 2) if index_cap is None else int(index_cap)

        def host(nbytes):
            buf = t.empty(max(nbytes, 16), dtype=t.uint8)
            return buf.pin_memory() if pinned else buf
        h_stream = host(n)
        if n:
            h_stream[:n].copy_(t.from_numpy(a))
        h_index = host(max(cap, 1) * NAL_ENTRY.itemsize)
        h_rbsp = host(n + 16) if want_rbsp else None
        summ = np.zeros(1, dtype=SUMMARY)
        rc = self.lib.hbs_index_extract_host(self.h, C.c_void_p(h_stream.data_ptr()), n, int(window_bytes),
                                             C.c_void_p(h_index.data_ptr()), cap,
                                             C.c_void_p(h_rbsp.data_ptr()) if h_rbsp is not None else None, n + 16,
                                             summ.ctypes.data_as(C.c_void_p))
        self._check(rc, "hbs_index_extract_host")
        s = summ[0]
        cnt = int(s["nal_count"])
        ent = h_index.numpy()[: cnt * NAL_ENTRY.itemsize].view(NAL_ENTRY).copy()
        arena = h_rbsp.numpy()[: int(s["rbsp_bytes"])].copy() if h_rbsp is not None else None
        return ent, arena, s

    def write_headers(self, parsed, structs, n_nals, rbsp_cap):
        """K5: serialise the structs of a parsed batch back to RBSP.  parsed: ndarray[PARSED] (host) or a device
        uint8 tensor of n records; structs: the device struct arena hbs_parse_headers filled.  Returns
        (written ndarray[WRITTEN], rbsp device tensor of n_nals * rbsp_cap bytes)."""
        t = self.torch
        dev = t.device("cuda", self.device)
        if isinstance(parsed, np.ndarray):
            parsed = t.from_numpy(np.ascontiguousarray(parsed).view(np.uint8).copy()).to(dev)
        out = t.empty(max(n_nals, 1) * rbsp_cap, dtype=t.uint8, device=dev)
        written = t.empty(max(n_nals, 1) * WRITTEN.itemsize, dtype=t.uint8, device=dev)
        self._bind_stream()
        rc = self.lib.hbs_write_headers(self.h, C.c_void_p(parsed.data_ptr()), n_nals, C.c_void_p(structs.data_ptr()),
                                        None, None, C.c_void_p(out.data_ptr()), rbsp_cap, C.c_void_p(written.data_ptr()))
        self._check(rc, "hbs_write_headers")
        return written[: n_nals * WRITTEN.itemsize].cpu().numpy().view(WRITTEN).copy(), out

    def emit_annexb_async(self, rbsp, rbsp_bytes, index, n_nals, gap_mode, out, index_out, summary):
        """Enqueue K3.  rbsp/index/out/index_out/summary are device tensors (index_out may be None)."""
        self._bind_stream()
        rc = self.lib.hbs_emit_annexb(self.h, C.c_void_p(rbsp.data_ptr()), rbsp_bytes, C.c_void_p(index.data_ptr()),
                                      n_nals, gap_mode, C.c_void_p(out.data_ptr()), out.numel(),
                                      C.c_void_p(index_out.data_ptr()) if index_out is not None else None,
                                      C.c_void_p(summary.data_ptr()))
        self._check(rc, "hbs_emit_annexb")

    def emit_annexb(self, rbsp, index_entries, gap_mode=0, out_cap=None):
        """Convenience: entries is a host ndarray[NAL_ENTRY]; returns (stream ndarray, entries_out).
        out_cap: size of the output buffer (default: the bound that always fits)."""
        t = self.torch
        dev = t.device("cuda", self.device)
        n = len(index_entries)
        d_idx = t.from_numpy(np.ascontiguousarray(index_entries).view(np.uint8).copy()).to(dev) if n else \
            t.zeros(NAL_ENTRY.itemsize, dtype=t.uint8, device=dev)
        rbsp_bytes = int(rbsp.numel())
        if out_cap is None:
            if gap_mode == 0 and n:          # the recorded gaps come on top of the 3/2 bound
                e = index_entries
                gaps = int(e["start"][0]) + int((e["start"][1:].astype(np.int64) - e["end"][:-1].astype(np.int64)).clip(min=0).sum())
                self.lib.hbs_annexb_bound_gaps.restype = C.c_uint64
                self.lib.hbs_annexb_bound_gaps.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
                out_cap = int(self.lib.hbs_annexb_bound_gaps(rbsp_bytes, n, gaps)) + 64
            else:
                out_cap = int(self.lib.hbs_annexb_bound(rbsp_bytes, n)) + 64
        out = t.empty(out_cap, dtype=t.uint8, device=dev)
        d_out_idx = t.empty(max(n, 1) * NAL_ENTRY.itemsize, dtype=t.uint8, device=dev)
        summary = t.zeros(SUMMARY.itemsize, dtype=t.uint8, device=dev)
        if rbsp_bytes == 0:
            rbsp = t.zeros(16, dtype=t.uint8, device=dev)
        self.emit_annexb_async(rbsp, rbsp_bytes, d_idx, n, gap_mode, out, d_out_idx, summary)
        s = self.read_summary(summary)
        if int(s["error"]) != 0:
            raise HbsError("hbs_emit_annexb: error %d" % int(s["error"]))
        return (out[: int(s["stream_bytes"])].cpu().numpy(),
                d_out_idx[: n * NAL_ENTRY.itemsize].cpu().numpy().view(NAL_ENTRY).copy())

    def synth_stream(self, seed, n_nals, mode=0):
        """Generate S(seed, n_nals, mode) in HBM.  Returns device tensors and sizes:
        dict(stream=u8[stream_bytes...], stream_bytes, rbsp=u8[...], rbsp_bytes, index=u8[n*32])."""
        t = self.torch
        dev = t.device("cuda", self.device)
        rbsp_cap = int(self.lib.hbs_synth_rbsp_bound(n_nals))
        rbsp = t.empty(rbsp_cap, dtype=t.uint8, device=dev)
        index = t.empty(max(n_nals, 1) * NAL_ENTRY.itemsize, dtype=t.uint8, device=dev)
        summary = t.zeros(SUMMARY.itemsize, dtype=t.uint8, device=dev)
        self._bind_stream()
        rc = self.lib.hbs_synth_rbsp(self.h, seed, n_nals, mode, C.c_void_p(rbsp.data_ptr()), rbsp_cap,
                                     C.c_void_p(index.data_ptr()), C.c_void_p(summary.data_ptr()))
        self._check(rc, "hbs_synth_rbsp")
        s = self.read_summary(summary)
        if int(s["error"]) != 0:
            raise HbsError("hbs_synth_rbsp: error %d" % int(s["error"]))
        rbsp_bytes = int(s["stream_bytes"])
        out_cap = int(self.lib.hbs_annexb_bound(rbsp_bytes, n_nals))
        stream = t.empty(out_cap, dtype=t.uint8, device=dev)
        self.emit_annexb_async(rbsp, rbsp_bytes, index, n_nals, 1, stream, index, summary)
        s = self.read_summary(summary)
        if int(s["error"]) != 0:
            raise HbsError("hbs_emit_annexb: error %d" % int(s["error"]))
        return dict(stream=stream, stream_bytes=int(s["stream_bytes"]), rbsp=rbsp, rbsp_bytes=rbsp_bytes,
                    index=index, n_nals=n_nals)

    # ---- K4 ---------------------------------------------------------------------------

    def parse_headers_async(self, rbsp, index, n_nals, parsed, structs, summary):
        """Enqueue K4.  All arguments are device tensors; structs may be None (plan only)."""
        self._bind_stream()
        rc = self.lib.hbs_parse_headers(self.h, C.c_void_p(rbsp.data_ptr()), C.c_void_p(index.data_ptr()), n_nals,
                                        C.c_void_p(parsed.data_ptr()),
                                        C.c_void_p(structs.data_ptr()) if structs is not None else None,
                                        structs.numel() if structs is not None else 0, C.c_void_p(summary.data_ptr()))
        self._check(rc, "hbs_parse_headers")

    def index_parse_async(self, stream, index, index_cap, parsed, structs, scan_summary, parse_summary, window=0, payload_off=None):
        """hbs_index_parse: index-only scan + header parse without an RBSP arena.  All arguments are device tensors
        (structs may be None: plan only; payload_off: optional int64 tensor, one per index entry).  Returns the NAL count
        (the call waits for the scan; the parse is enqueued behind it)."""
        self._bind_stream()
        self.lib.hbs_index_parse.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p,
                                             C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
        n = C.c_uint64(0)
        rc = self.lib.hbs_index_parse(self.h, C.c_void_p(stream.data_ptr() if stream.numel() else None), stream.numel(),
                                      C.c_void_p(index.data_ptr()), index_cap, window, C.c_void_p(parsed.data_ptr()),
                                      C.c_void_p(structs.data_ptr()) if structs is not None else None,
                                      structs.numel() if structs is not None else 0,
                                      C.c_void_p(payload_off.data_ptr()) if payload_off is not None else None,
                                      C.c_void_p(scan_summary.data_ptr()), C.c_void_p(parse_summary.data_ptr()), C.byref(n))
        self._check(rc, "hbs_index_parse")
        return int(n.value)

    def parse_compact_async(self, rbsp, index, n_nals, parsed, compact, structs, summary, want=None, initial_sps_slot=None, initial_pps=None):
        """hbs_parse_headers_compact (want is None) / hbs_parse_materialize (want: int64 / uint64 device tensor of NAL numbers).
        structs None: plan only."""
        self._bind_stream()
        p = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None          # noqa: E731
        common = [self.h, p(rbsp), p(index), n_nals, p(parsed), p(compact), p(structs), structs.numel() if structs is not None else 0,
                  p(initial_sps_slot), p(initial_pps)]
        if want is None:
            self.lib.hbs_parse_headers_compact.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                                           C.c_void_p, C.c_void_p, C.c_void_p]
            self._check(self.lib.hbs_parse_headers_compact(*common, p(summary)), "hbs_parse_headers_compact")
        else:
            self.lib.hbs_parse_materialize.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
            self._check(self.lib.hbs_parse_materialize(*common, p(want), want.numel(), p(summary)), "hbs_parse_materialize")

    def parse_headers_compact(self, rbsp, index, n_nals, want=None, poison=None):
        """Plan, allocate, parse: (parsed ndarray[PARSED], compact ndarray[COMPACT], structs device tensor holding the parameter
        sets and -- want: a list of NAL numbers -- the full slice headers of those NALs)."""
        t = self.torch
        dev = t.device("cuda", self.device)
        parsed = t.empty(max(n_nals, 1) * PARSED.itemsize, dtype=t.uint8, device=dev)
        compact = t.empty(max(n_nals, 1) * COMPACT.itemsize, dtype=t.uint8, device=dev)
        summary = t.zeros(SUMMARY.itemsize, dtype=t.uint8, device=dev)
        want_dev = None if want is None else t.as_tensor(np.asarray(want, dtype=np.int64), device=dev)
        self.parse_compact_async(rbsp, index, n_nals, parsed, compact, None, summary, want_dev)
        need = int(self.read_summary(summary)["reserved"][0])
        structs = t.empty(need + 16, dtype=t.uint8, device=dev)
        if poison is not None:
            structs.fill_(poison)
        self.parse_compact_async(rbsp, index, n_nals, parsed, compact, structs, summary, want_dev)
        s = self.read_summary(summary)
        if int(s["error"]) != 0:
            e = HbsError("hbs_parse_headers_compact: error %d" % int(s["error"]))
            e.code = int(s["error"])
            raise e
        return (parsed[: n_nals * PARSED.itemsize].cpu().numpy().view(PARSED).copy(),
                compact[: n_nals * COMPACT.itemsize].cpu().numpy().view(COMPACT).copy(), structs)

    def index_parse_compact_async(self, stream, index, index_cap, parsed, compact, structs, scan_summary, parse_summary, window=0, payload_off=None):
        """hbs_index_parse_compact: as index_parse_async with the compact parse behind the scan"""
        self._bind_stream()
        self.lib.hbs_index_parse_compact.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                     C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
        n = C.c_uint64(0)
        rc = self.lib.hbs_index_parse_compact(self.h, C.c_void_p(stream.data_ptr() if stream.numel() else None), stream.numel(),
                                              C.c_void_p(index.data_ptr()), index_cap, window, C.c_void_p(parsed.data_ptr()), C.c_void_p(compact.data_ptr()),
                                              C.c_void_p(structs.data_ptr()) if structs is not None else None,
                                              structs.numel() if structs is not None else 0,
                                              C.c_void_p(payload_off.data_ptr()) if payload_off is not None else None,
                                              C.c_void_p(scan_summary.data_ptr()), C.c_void_p(parse_summary.data_ptr()), C.byref(n))
        self._check(rc, "hbs_index_parse_compact")
        return int(n.value)

    def parse_extended(self, rbsp, index, n_nals, parsed_dev):
        """The NAL types the reference never dispatches (AUD, EOS, EOB, filler data, SEI), behind parse_headers on the same
        arrays: updates parsed_dev[k].rc for those NALs in place and returns the hbs_ext_nal records (ndarray[EXT_NAL])."""
        t = self.torch
        ext = t.empty(max(n_nals, 1) * EXT_NAL.itemsize, dtype=t.uint8, device=t.device("cuda", self.device))
        self._bind_stream()
        self.lib.hbs_parse_extended.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        self._check(self.lib.hbs_parse_extended(self.h, C.c_void_p(rbsp.data_ptr()), C.c_void_p(index.data_ptr()), n_nals,
                                                C.c_void_p(parsed_dev.data_ptr()), C.c_void_p(ext.data_ptr())), "hbs_parse_extended")
        return ext[: n_nals * EXT_NAL.itemsize].cpu().numpy().view(EXT_NAL).copy()

    def parse_headers(self, rbsp, index, n_nals, poison=None):
        """Plan, allocate the struct arena, parse.  Returns (parsed ndarray[PARSED], structs device tensor).
        poison: byte the arena is filled with beforehand (tests: the parse has to clear what it fills)."""
        t = self.torch
        dev = t.device("cuda", self.device)
        parsed = t.empty(max(n_nals, 1) * PARSED.itemsize, dtype=t.uint8, device=dev)
        summary = t.zeros(SUMMARY.itemsize, dtype=t.uint8, device=dev)
        self.parse_headers_async(rbsp, index, n_nals, parsed, None, summary)
        need = int(self.read_summary(summary)["reserved"][0])
        structs = t.empty(need + 16, dtype=t.uint8, device=dev)
        if poison is not None:
            structs.fill_(poison)
        self.parse_headers_async(rbsp, index, n_nals, parsed, structs, summary)
        s = self.read_summary(summary)
        if int(s["error"]) != 0:
            raise HbsError("hbs_parse_headers: error %d" % int(s["error"]))
        return parsed[: n_nals * PARSED.itemsize].cpu().numpy().view(PARSED).copy(), structs
