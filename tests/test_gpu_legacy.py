"""The reference's single-NAL C API served by the GPU (hbs_legacy.c over the
batch C ABI): golden vectors generated from the real reference, the parse
goldens, and the reference's unmodified hevc_analyze linked against this library."""
import ctypes as C
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from tests import _orc

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def leg():
    import hevcbitstream_amd as hbs
    return _orc._L2(hbs.load_library(), "")


def test_find_nal_unit_golden(leg):
    gold = json.load(open(os.path.join(HERE, "golden", "l2_vectors.json")))
    for hx, want in gold["find"]:
        assert list(leg.find_nal_unit(bytes.fromhex(hx))) == want, hx


def test_nal_to_rbsp_golden(leg):
    gold = json.load(open(os.path.join(HERE, "golden", "l2_vectors.json")))
    for hx, (r, ns, rs, data) in gold["n2r"]:
        got = leg.nal_to_rbsp(bytes.fromhex(hx))
        assert got[0] == r and got[1] == ns and got[2] == rs, hx
        if r >= 0:
            assert got[3].hex() == data, hx


def test_rbsp_to_nal_golden(leg):
    gold = json.load(open(os.path.join(HERE, "golden", "l2_vectors.json")))
    for hx, (r, data) in gold["r2n"]:
        got = leg.rbsp_to_nal(bytes.fromhex(hx))
        assert got[0] == r and got[2].hex() == data, hx


def test_find_nal_unit_large_buffer(leg, orc):
    """NALs far larger than the first 64 KiB prefix the wrapper uploads"""
    rng = np.random.RandomState(3)
    body = rng.randint(4, 256, size=700000).astype(np.uint8).tobytes()
    buf = b"\x11\x22\x00\x00\x01" + body + b"\x00\x00\x01\x40\x41" + body[:1000]
    assert leg.find_nal_unit(buf) == orc.find_nal_unit(buf)
    assert leg.find_nal_unit(buf[:400000]) == orc.find_nal_unit(buf[:400000])        # no end: -1


def test_find_nal_unit_loop_and_edited_buffer(orc):
    """the caller's loop over ONE buffer (hevc_analyze.c:135-177): the wrapper answers most calls from the previous
    scan of the same bytes -- and must not when the bytes changed, when the loop resumes elsewhere, or when the
    buffer got shorter than the answer needs"""
    import hevcbitstream_amd as hbs
    lib = hbs.load_library()
    u8p = C.POINTER(C.c_uint8)
    lib.find_nal_unit.argtypes = [u8p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    rng = np.random.RandomState(17)
    for rep in range(4):
        parts = []
        for k in range(60):
            parts.append(bytes([0] * int(rng.randint(2, 4)) + [1]))
            body = rng.randint(0 if rep == 3 else 1, 256, size=int(rng.randint(1, 3000))).astype(np.uint8)
            parts.append(body.tobytes())
        buf = np.frombuffer(b"".join(parts), dtype=np.uint8).copy()
        base = buf.ctypes.data

        def ours(p, size):
            s, e = C.c_int(0), C.c_int(0)
            r = lib.find_nal_unit(C.cast(base + p, u8p), size, C.byref(s), C.byref(e))
            return r, s.value, e.value

        p, n = 0, 0
        while True:
            want = orc.find_nal_unit(bytes(buf[p:]))
            assert ours(p, len(buf) - p) == tuple(want), (rep, n, p)
            if want[0] <= 0:
                break
            if n == 5:                                   # the bytes of the next answer change under the wrapper's feet
                q = p + want[2] + 6
                buf[q:q + 3] = (0, 0, 1)
            if n == 9:                                   # ... the caller looks at a shorter buffer
                short = len(buf) - p - want[2] - 2
                assert ours(p + want[2], 3) == tuple(orc.find_nal_unit(bytes(buf[p + want[2]:p + want[2] + 3])))
                del short
            if n == 12:                                  # ... and resumes one byte off the previous NAL's end
                assert ours(p + want[2] + 1, len(buf) - p - want[2] - 1) == tuple(orc.find_nal_unit(bytes(buf[p + want[2] + 1:])))
            p += want[2]
            n += 1
        assert n >= 10


class LegacyHevc(_orc._HevcParser):
    def __init__(self, lib):
        lib.hevc_new.restype = C.c_void_p
        lib.read_hevc_nal_unit.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int]
        self.L = lib
        self.h = lib.hevc_new()
        self._views(self.h)

    def read(self, nal):
        buf = np.frombuffer(bytes(nal) + b"\xff" * 8, dtype=np.uint8).copy()
        return self.L.read_hevc_nal_unit(self.h, buf.ctypes.data_as(C.POINTER(C.c_uint8)), len(nal))


def test_read_hevc_nal_unit_golden():
    """hevc_new + read_hevc_nal_unit per NAL, against the reference's struct dumps"""
    import hevcbitstream_amd as hbs
    gold = json.load(open(os.path.join(HERE, "golden", "parse_vectors.json")))
    for seq in gold[:15]:
        p = LegacyHevc(hbs.load_library())
        for step in seq["steps"]:
            nal = bytes.fromhex(step["nal"])
            assert p.read(nal) == step["rc"], (seq["seed"], step["nal"][:8])
            snap = p.snapshot()
            for k, pairs in step["structs"].items():
                want = np.zeros_like(snap[k])
                for i, v in pairs:
                    want[i] = v
                assert np.array_equal(snap[k], want), (seq["seed"], k, step["nal"][:8])
            if "slice_data" in step:
                size, data = p.slice_data()
                assert size == step["slice_data"][0]
                if data is not None:
                    assert hashlib.md5(data).hexdigest() == step["slice_data"][1]


def test_read_sequential_table_state():
    """read_hevc_nal_unit NAL by NAL keeps ONE set of derived RPS tables like the reference (hevc_stream.c:26-32):
    an SPS cut short that declares no sets, then slices that name set 0 anyway, read what earlier slices left there.
    The oracle follows the reference on these (the batch parse reads zeros: DESIGN.md section 7)."""
    import hevcbitstream_amd as hbs
    from tests.test_sim_parse_logic import broken, sequence
    lib = hbs.load_library()
    seeds = [1036, 1064, 1320, 1496, 5224] + list(range(2000, 2040, 4))
    for seed in seeds:
        nals = broken(sequence(seed), np.random.RandomState(7 * seed + 2), lambda t: True)
        lib.hbs_legacy_reset_tables()
        ours, orc_p = LegacyHevc(lib), _orc.OracleHevc()
        for k, nal in enumerate(nals):
            assert ours.read(nal) == orc_p.read(nal), (seed, k)
            a, b = ours.snapshot(), orc_p.snapshot()
            t = (nal[0] >> 1) & 0x3F
            kind = "sh" if (t <= 9 or 16 <= t <= 21) else {32: "vps", 33: "sps", 34: "pps"}.get(t)
            for name in (["nal"] + ([kind] if kind else [])):
                assert np.array_equal(a[name], b[name]), (seed, k, name)
        orc_p.close()


def test_reference_hevc_analyze_links_and_runs():
    """the reference's own CLI (hevc_analyze.c, unmodified) built against include/ + this library
    by `make analyze`: its NAL walk (find_nal_unit on the GPU) must print the golden offsets/sizes,
    the debug_bytes dumps and the NAL-header lines of the golden stdout."""
    exe = os.path.join(ROOT, "oracle", "_ref", "hevc_analyze_amd")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/hevc_analyze_amd not built (needs /root/reference at build time)")
    out = subprocess.run([exe, os.path.join(HERE, "golden", "ten_nal.hevc")], stdout=subprocess.PIPE, check=True).stdout.decode()
    want = open(os.path.join(HERE, "golden", "ten_nal.analyze.txt")).read()

    def keep(text):
        lines = text.splitlines()
        sel = []
        for i, l in enumerate(lines):
            if l.startswith("!! Found NAL"):
                sel.append(l)
                sel.append(lines[i + 1][12:])        # hex dump without the 4 bytes in front of the buffer
            elif l.startswith(("0.8: forbidden", "0.7: nal->", "0.1: nal->", "1.3: nal->")):
                sel.append(l)
        return sel

    assert keep(out) == keep(want)
    assert len(keep(out)) == 10 * 6


def test_reference_cli_full_stdout():
    """the reference's CLI (unmodified hevc_analyze.c) on this library prints, line for line, what it prints
    on its own library: NAL walk on the GPU, header parse + per-field trace on the GPU, names from
    hbs_trace_names.h.  Golden outputs: tests/golden/make_trace_names.py."""
    import gzip
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "hevc_analyze_amd")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/hevc_analyze_amd not built (needs /root/reference at build time)")
    vectors = json.load(gzip.open(os.path.join(HERE, "golden", "trace_vectors.json.gz"), "rt"))

    def norm(text):
        lines = text.splitlines()
        # the first hex dump starts 4 bytes in front of the file buffer (heap bytes): drop those 4
        for i, l in enumerate(lines):
            if l.startswith("!! Found NAL"):
                lines[i + 1] = lines[i + 1][12:]
                break
        return lines

    for v in vectors:
        with tempfile.NamedTemporaryFile(suffix=".hevc", delete=False) as f:
            f.write(bytes.fromhex(v["stream"]))
            path = f.name
        try:
            out = subprocess.run([exe, path], stdout=subprocess.PIPE, check=True).stdout.decode("latin-1")
        finally:
            os.unlink(path)
        got, want = norm(out), norm(v["stdout"])
        assert got == want, (v["tag"], [(a, b) for a, b in zip(got, want) if a != b][:3], len(got), len(want))


@pytest.mark.parametrize("kind", ["plain-11MB", "out-of-spec", "long-traces", "no-batch"])
def test_reference_cli_on_large_streams_equals_the_reference_binary(kind, tmp_path):
    """The loop of hevc_analyze.c:135-177 answered from ONE batch per buffer (hbs_legacy.c, round 4): the unmodified CLI on this
    library against the same CLI on the compiled reference, whole stdout, on streams large enough for the batch path --
    an 11 MB / 2 015-NAL sequence; one with an out-of-spec slice every 50 (their headers depend on the tables earlier
    NALs left: the batch's exact re-walk and the state behind it are what is tested); NALs whose traces
    are longer than the batch keeps (those NALs go one call at a time, the batch continues behind them); and the old way
    (HBS_LEGACY_NO_BATCH=1) for comparison."""
    from tests.hevc_synth import stream_4k30
    amd = os.path.join(ROOT, "oracle", "_ref", "hevc_analyze_amd")
    ref = os.path.join(ROOT, "oracle", "_ref", "hevc_analyze_ref")
    if not (os.path.exists(amd) and os.path.exists(ref)):
        pytest.skip("oracle/_ref/hevc_analyze_amd / _ref not built (needs /root/reference at build time)")
    if kind == "plain-11MB":
        stream, n = stream_4k30(11, n_pictures=250, slices_per_picture=8, idr_every=60, payload_bytes=(2000, 9000))
    elif kind == "out-of-spec":
        stream, n = stream_4k30(21, n_pictures=300, slices_per_picture=8, idr_every=60, payload_bytes=(200, 900), forbidden_every=50)
    elif kind == "long-traces":      # (the batch made to keep 40 trace records per NAL: parameter sets and many slices exceed that)
        stream, n = stream_4k30(5, n_pictures=200, slices_per_picture=4, idr_every=10, payload_bytes=(400, 1500))
    else:
        stream, n = stream_4k30(13, n_pictures=60, slices_per_picture=8, idr_every=20, payload_bytes=(2000, 4000))
    assert len(stream) > (128 << 10)
    path = str(tmp_path / "s.hevc")
    open(path, "wb").write(stream)
    env = dict(os.environ)
    env.pop("HBS_LEGACY_NO_BATCH", None)
    env.pop("HBS_LEGACY_TRACE_CAP", None)
    if kind == "no-batch":
        env["HBS_LEGACY_NO_BATCH"] = "1"
    if kind == "long-traces":
        env["HBS_LEGACY_TRACE_CAP"] = "40"
    want = subprocess.run([ref, path], stdout=subprocess.PIPE, check=True).stdout.decode("latin-1").splitlines()
    got = subprocess.run([amd, path], stdout=subprocess.PIPE, check=True, env=env).stdout.decode("latin-1").splitlines()
    for lines in (want, got):          # the first hex dump starts 4 bytes in front of the file buffer (heap bytes)
        for i, l in enumerate(lines):
            if l.startswith("!! Found NAL"):
                lines[i + 1] = lines[i + 1][12:]
                break
    assert len(got) == len(want) and got == want, (kind, n, [(i, a, b) for i, (a, b) in enumerate(zip(got, want)) if a != b][:3], len(got), len(want))


def test_write_hevc_nal_unit_golden():
    """hevc_new + read_hevc_nal_unit + (edit) + write_hevc_nal_unit per NAL against the reference's writer
    (tests/golden/make_golden_write.py): bytes, return value and the slice_data side effect."""
    import gzip
    import hevcbitstream_amd as hbs
    from tests._parsecmp import which_struct
    vectors = json.load(gzip.open(os.path.join(HERE, "golden", "write_vectors.json.gz"), "rt"))

    def field_index(kind, name):
        for n, i, c in _orc.flat_fields(_orc.STRUCT_TYPES[kind]):
            if n == name:
                return i
        raise KeyError(name)

    lib = hbs.load_library()
    lib.write_hevc_nal_unit.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int]
    for v in vectors:
        p = LegacyHevc(lib)
        for k, st in enumerate(v["steps"]):
            nal = bytes.fromhex(st["nal"])
            assert p.read(nal) == st["read_rc"], (v["seed"], k)
            if "write_rc" not in st:
                continue
            kind = which_struct(int(p.v["nal"][1]))
            saved = p.v[kind].copy()
            for name, value in st["edits"]:
                p.v[kind][field_index(kind, name)] = value
            out = np.zeros(st["size"] + 16, dtype=np.uint8)
            rc = lib.write_hevc_nal_unit(p.h, out.ctypes.data_as(C.POINTER(C.c_uint8)), st["size"])
            p.v[kind][:] = saved
            assert rc == st["write_rc"], (v["seed"], k, kind, rc, st["write_rc"])
            if rc > 0:
                assert bytes(out[:rc]).hex() == st["out"], (v["seed"], k, kind, st["edits"])
            if kind == "sh":
                assert p.slice_data()[0] == st["slice_data_size"], (v["seed"], k)


def test_peek_hevc_nal_unit_against_the_reference():
    """peek_hevc_nal_unit (hevc_nal.c:97-114): every 2-byte header, and buffers shorter than a header, against the
    compiled reference when its prebuilt library travelled with the tree, else against the rule of :103-111 spelled out"""
    import hevcbitstream_amd as hbs
    lib = hbs.load_library()
    lib.hevc_new.restype = C.c_void_p
    lib.peek_hevc_nal_unit.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int]
    ours = LegacyHevc(lib)
    ref = None
    refso = os.path.join(ROOT, "oracle", "_ref", "libhevcref.so")
    if os.path.exists(refso):
        rl = C.CDLL(refso)
        rl.hevc_new.restype = C.c_void_p
        rl.peek_hevc_nal_unit.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int]
        ref = (rl, _orc._HevcParser.__new__(_orc._HevcParser))
        ref[1].h = rl.hevc_new()
        ref[1]._views(ref[1].h)
    cases = [(bytes([b0, b1]), 2) for b0 in range(256) for b1 in range(0, 256, 5)] + [(b"\x40\x01\x0c", 3), (b"\x26", 1), (b"", 0), (b"\x7e\xff", 2)]
    for data, size in cases:
        buf = np.frombuffer(data + b"\x00" * 8, dtype=np.uint8).copy()
        rc = lib.peek_hevc_nal_unit(ours.h, buf.ctypes.data_as(C.POINTER(C.c_uint8)), size)
        got = tuple(int(x) for x in ours.snapshot()["nal"])
        if ref is not None:
            want_rc = ref[0].peek_hevc_nal_unit(ref[1].h, buf.ctypes.data_as(C.POINTER(C.c_uint8)), size)
            want = tuple(int(x) for x in ref[1].snapshot()["nal"])
        else:
            b0 = data[0] if size > 0 else 0
            b1 = data[1] if size > 1 else 0
            t = (b0 >> 1) & 0x3F
            want = (0, t, ((b0 & 1) << 5) | (b1 >> 3), b1 & 7)
            want_rc = -1 if (t <= 0 or t > 63) else t
        assert rc == want_rc, (data.hex(), size, rc, want_rc)
        assert got[1:] == want[1:], (data.hex(), got, want)


def test_legacy_symbols_from_two_threads(leg, orc):
    """find_nal_unit / nal_to_rbsp / rbsp_to_nal are pure and re-entrant in the reference (h264_nal.c:38-200); here they share one
    GPU context behind a lock: two threads hammering them at once (ctypes drops the GIL inside a call) get the right answers"""
    import threading
    gold = json.load(open(os.path.join(HERE, "golden", "l2_vectors.json")))
    errors = []

    def finder():
        try:
            for _ in range(3):
                for hx, want in gold["find"][:150]:
                    if list(leg.find_nal_unit(bytes.fromhex(hx))) != want:
                        errors.append(("find", hx))
        except Exception as e:          # noqa: BLE001
            errors.append(("find", repr(e)))

    def converter():
        try:
            for _ in range(2):
                for hx, (r, ns, rs, data) in gold["n2r"][:120]:
                    got = leg.nal_to_rbsp(bytes.fromhex(hx))
                    if not (got[0] == r and got[1] == ns and got[2] == rs and (r < 0 or got[3].hex() == data)):
                        errors.append(("n2r", hx))
                for hx, (r, data) in gold["r2n"][:120]:
                    got = leg.rbsp_to_nal(bytes.fromhex(hx))
                    if not (got[0] == r and got[2].hex() == data):
                        errors.append(("r2n", hx))
        except Exception as e:          # noqa: BLE001
            errors.append(("conv", repr(e)))

    th = [threading.Thread(target=finder), threading.Thread(target=converter), threading.Thread(target=finder)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:5]


def test_batch_api_two_contexts_two_threads(orc):
    """the batch API's contract (include/hevcbitstream_amd.h): a context belongs to one thread at a time; different contexts
    (each with its own scratch and its own stream) run side by side.  Plain C ABI through ctypes: no torch in this file."""
    import threading
    import hevcbitstream_amd as hbs
    lib = hbs.load_library()
    lib.hbs_dev_alloc.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
    lib.hbs_dev_free.argtypes = [C.c_void_p, C.c_void_p]
    lib.hbs_copy_to_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    lib.hbs_copy_to_host.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    errors = []
    gen_lock = threading.Lock()          # the oracle is test infrastructure and makes no promise about threads

    def worker(seed):
        try:
            h = C.c_void_p()
            assert lib.hbs_ctx_create(C.byref(h), 0) == 0
            for rep in range(6):
                with gen_lock:
                    stream, idx, arena = orc.gen_stream(seed + rep, 400 + 50 * rep, rep % 2)
                n, cap = len(stream), len(idx) + 8
                bufs = []
                for nbytes in (n + 64, cap * 32, n + 64, 64):
                    p = C.c_void_p()
                    assert lib.hbs_dev_alloc(h, nbytes, C.byref(p)) == 0
                    bufs.append(p)
                d_stream, d_index, d_rbsp, d_sum = bufs
                assert lib.hbs_copy_to_device(h, d_stream, stream.ctypes.data, n) == 0
                assert lib.hbs_index_extract(h, d_stream, n, d_index, cap, d_rbsp, n + 16, d_sum) == 0
                summ = np.zeros(1, dtype=hbs.SUMMARY)
                assert lib.hbs_read_summary(h, d_sum, summ.ctypes.data) == 0
                got_idx = np.zeros(len(idx), dtype=hbs.NAL_ENTRY)
                got_arena = np.zeros(len(arena), dtype=np.uint8)
                assert lib.hbs_copy_to_host(h, got_idx.ctypes.data, d_index, got_idx.nbytes) == 0
                assert lib.hbs_copy_to_host(h, got_arena.ctypes.data, d_rbsp, got_arena.nbytes) == 0
                ok = int(summ[0]["nal_count"]) == len(idx) and int(summ[0]["rbsp_bytes"]) == len(arena)
                ok = ok and all(np.array_equal(got_idx[f], idx[f]) for f in ("start", "end", "rbsp_off", "rbsp_len", "status")) and np.array_equal(got_arena, arena)
                if not ok:
                    errors.append((seed, rep, int(summ[0]["nal_count"]), len(idx), int(summ[0]["rbsp_bytes"]), len(arena), int(summ[0]["error"]),
                                   [f for f in ("start", "end", "rbsp_off", "rbsp_len", "status") if not np.array_equal(got_idx[f], idx[f])]))
                for p in bufs:
                    lib.hbs_dev_free(h, p)
            lib.hbs_ctx_destroy(h)
        except Exception as e:          # noqa: BLE001
            errors.append((seed, repr(e)))

    th = [threading.Thread(target=worker, args=(0x900 + 16 * i,)) for i in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:5]


def _picture_stream(seed, n_pictures, with_aud=True, forbidden_every=0):
    """NALs of a 4K30-like sequence with an AUD in front of every picture and an SEI behind its first slice (neither is
    dispatched by the reference's reader: callers skip them or get -1), padded to batch size"""
    from tests.hevc_synth import Synth
    g = Synth(seed, rich=False)
    rng = np.random.RandomState(seed + 1)
    nals, count = [], 0
    for pic in range(n_pictures):
        idr = pic % 20 == 0
        if with_aud:
            nals.append(bytes([35 << 1, 1, 0x50]))
        if idr:
            nals += [g.vps(), g.sps_nal(3840, 2160, ctb_log2=6), g.pps_nal(force={"tiles": 0, "lists_mod": 1} if forbidden_every else {"tiles": 0})]
        for sl in range(4):
            payload = rng.randint(0, 256, size=int(rng.randint(300, 1500))).astype(np.uint8).tobytes()
            count += 1
            if forbidden_every and not idr and count % forbidden_every == 0:
                nals.append(g.slice_nal(19, first=(sl == 0), payload=payload, slice_type=1, address=sl * 510))
            else:
                nals.append(g.slice_nal(19 if idr else 1, first=(sl == 0), payload=payload, address=sl * 510))
            if sl == 0 and with_aud:
                nals.append(bytes([39 << 1, 1, 1, 2, 3, 0x80]))
    return nals


@pytest.mark.parametrize("caller", ["in-order", "skips-aud-sei", "scratch-copy", "skips-a-slice", "flips-reader"])
def test_batch_loop_with_deviating_callers(caller):
    """One batch per buffer (hbs_legacy.c) under callers that are NOT hevc_analyze's loop (round 4's advice): a caller that skips
    the AUD / SEI NALs (the batch goes on at the NAL it asks for), one that parses every NAL from a scratch copy (the batch never
    fits: after the first, the back-off keeps them from being built once per NAL), one that skips a slice (ends the batch: the
    state behind the skipped range must not include it), one that reads some NALs with the other reader.  Every answer against
    the oracle's sequential parser fed the same calls."""
    import hevcbitstream_amd as hbs
    from tests.hevc_synth import annexb
    lib = hbs.load_library()
    u8p = C.POINTER(C.c_uint8)
    lib.find_nal_unit.argtypes = [u8p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.read_hevc_nal_unit.argtypes = [C.c_void_p, u8p, C.c_int]
    lib.read_debug_hevc_nal_unit.argtypes = [C.c_void_p, u8p, C.c_int]
    lib.hbs_legacy_batch_stats.argtypes = [C.POINTER(C.c_uint64)]
    nals = _picture_stream(31, 120, forbidden_every=9 if caller == "skips-a-slice" else 0)
    # (a sacrificial NAL behind the last one: find_nal_unit does not look for an end in a buffer's last three bytes, h264_nal.c:64-72)
    buf = np.frombuffer(annexb(nals + [bytes([35 << 1, 1, 0x50])]), dtype=np.uint8).copy()
    assert len(buf) > (128 << 10)
    base = buf.ctypes.data
    lib.hbs_legacy_reset_tables()
    ours, orc_p = LegacyHevc(lib), _orc.OracleHevc()
    st0 = (C.c_uint64 * 4)()
    lib.hbs_legacy_batch_stats(st0)
    devnull = os.open(os.devnull, os.O_WRONLY)
    p, k, reads = 0, 0, 0
    while True:
        s, e = C.c_int(0), C.c_int(0)
        r = lib.find_nal_unit(C.cast(base + p, u8p), len(buf) - p, C.byref(s), C.byref(e))
        if r <= 0:
            break
        nal = bytes(buf[p + s.value: p + e.value])
        assert nal == nals[k], k
        t = (nal[0] >> 1) & 0x3F
        skip = (caller == "skips-aud-sei" and t in (35, 39)) or (caller == "skips-a-slice" and k % 37 == 20 and t < 32)
        if not skip:
            if caller == "scratch-copy":
                tmp = np.frombuffer(nal + b"\xff" * 8, dtype=np.uint8).copy()
                ra = lib.read_hevc_nal_unit(ours.h, tmp.ctypes.data_as(u8p), len(nal))
            elif caller == "flips-reader" and k % 50 == 25:
                # the debug reader prints to stdout: send the C library's stdout to /dev/null for this call
                C.CDLL(None).fflush(None)
                keep = os.dup(1)
                os.dup2(devnull, 1)
                try:
                    ra = lib.read_debug_hevc_nal_unit(ours.h, C.cast(base + p + s.value, u8p), len(nal))
                    C.CDLL(None).fflush(None)
                finally:
                    os.dup2(keep, 1)
                    os.close(keep)
            else:
                ra = lib.read_hevc_nal_unit(ours.h, C.cast(base + p + s.value, u8p), len(nal))
            rb = orc_p.read(nal)
            assert ra == rb, (caller, k, t, ra, rb)
            a, b = ours.snapshot(), orc_p.snapshot()
            kind = "sh" if (t <= 9 or 16 <= t <= 21) else {32: "vps", 33: "sps", 34: "pps"}.get(t)
            for name in (["nal"] + ([kind] if kind else [])):
                if not np.array_equal(a[name], b[name]):
                    bad = np.flatnonzero(a[name] != b[name])
                    names = [(f, int(a[name][i]), int(b[name][i])) for f, i0, cnt in _orc.flat_fields(_orc.STRUCT_TYPES[name]) for i in bad[:6] if i0 <= i < i0 + cnt]
                    raise AssertionError((caller, k, name, "type %d" % t, names, [(nals[j][0] >> 1) & 0x3F for j in range(max(0, k - 6), k + 1)]))
            if kind == "sh" and ra >= 0:
                assert ours.slice_data() == orc_p.slice_data(), (caller, k)
            reads += 1
        p += e.value
        k += 1
    os.close(devnull)
    assert k == len(nals)
    st = (C.c_uint64 * 4)()
    lib.hbs_legacy_batch_stats(st)
    built, from_batch, one_by_one, suppressed = [int(st[i]) - int(st0[i]) for i in range(4)]
    assert from_batch + one_by_one == reads
    if caller in ("in-order", "skips-aud-sei"):
        assert built == 1 and from_batch == reads, (built, from_batch, one_by_one, reads)
    if caller == "scratch-copy":
        # batches of 128 KiB+ are tried at find calls 1, 3, 6, 11, 20, ...: a dozen for ~1000 NALs, not one per NAL
        assert from_batch == 0 and built <= 12 and suppressed > 500, (built, suppressed, reads)
    if caller in ("skips-a-slice", "flips-reader"):
        assert from_batch > reads // 2, (built, from_batch, one_by_one, reads)
    orc_p.close()
