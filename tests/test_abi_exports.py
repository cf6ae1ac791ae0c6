"""CPU-side checks of the drop-in boundary: the library loads without a GPU and
exports every symbol the headers in include/ declare (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    names = set()
    for hdr in ("hevcbitstream_amd.h", "h264_stream.h", "hevc_stream.h"):
        src = open(os.path.join(ROOT, "include", hdr)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        src = re.sub(r"^\s*#.*?$", "", src, flags=re.M)
        for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", src):
            n = m.group(1)
            if n not in ("defined", "sizeof"):
                names.add(n)
    return names


def test_library_exports_everything_declared():
    import hevcbitstream_amd as hbs
    lib = hbs.load_library()
    names = declared_functions()
    assert {"hbs_index_extract", "hbs_emit_annexb", "hbs_parse_headers", "find_nal_unit", "nal_to_rbsp",
            "rbsp_to_nal", "read_hevc_nal_unit", "write_hevc_nal_unit", "hevc_new", "hevc_free",
            "read_debug_hevc_nal_unit", "debug_bytes"} <= names
    for n in sorted(names):
        assert hasattr(lib, n), "declared in include/ but not exported: " + n
    C.c_void_p.in_dll(lib, "h264_dbgfile")          # data symbol hevc_analyze.c uses


def test_nothing_is_exported_that_no_header_declares():
    """the other direction: EVERY defined dynamic symbol of the library -- functions and data, C++ names included (round 5's
    library exported 77 mangled launchers and 43 kernel handles beside the C ABI) -- is a function declared in include/ or the
    reference's data symbol h264_dbgfile; csrc/exports.map makes everything else local"""
    import subprocess
    so = os.path.join(ROOT, "hevcbitstream_amd", "libhevcbitstream_amd.so")
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[2].split("@")[0] for ln in out.splitlines() if len(ln.split()) == 3}
    mangled = {n for n in exported if n.startswith("_Z")}
    assert not mangled, "C++ symbols exported: %s ..." % sorted(mangled)[:5]
    extra = exported - declared_functions() - {"h264_dbgfile"}
    assert not extra, "exported but declared in no header of include/: %s" % sorted(extra)


def test_python_binding_lists_the_same_batch_api():
    from hevcbitstream_amd.api import EXPORTS
    for n in EXPORTS:
        assert n in declared_functions(), n


def test_no_gpu_fails_loudly():
    """without a GPU the context cannot be created: there is no CPU fallback"""
    import torch
    import hevcbitstream_amd as hbs
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = hbs.load_library()
    h = C.c_void_p()
    assert lib.hbs_ctx_create(C.byref(h), 0) == -1


def test_struct_sizes_of_public_records():
    import hevcbitstream_amd as hbs
    assert hbs.NAL_ENTRY.itemsize == 32 and hbs.SUMMARY.itemsize == 64 and hbs.PARSED.itemsize == 32


def test_only_the_checkers_touch_the_oracle():
    """oracle/ is test infrastructure: the product (package sources, the C-ABI library's objects, scripts/) never
    names it; bench.py does so only inside its cpu_baseline leg (the functions named cpu_baseline*), __graft_entry__.py in build() (building the
    checker) and smoke() (checking against it).  Tools that compare against the oracle live in tests/tools/."""
    pat = re.compile(r"\boracle\b|_orc\b|liboracle|hbs_oracle")
    product = []
    for d, _, files in os.walk(os.path.join(ROOT, "hevcbitstream_amd")):
        product += [os.path.join(d, f) for f in files if f.endswith((".py", ".hip", ".h", ".c", ".cpp"))]
    product += [os.path.join(ROOT, "scripts", f) for f in os.listdir(os.path.join(ROOT, "scripts")) if f.endswith(".py")]
    for path in product:
        for ln, line in enumerate(open(path, errors="replace"), 1):
            code = line.split("#")[0] if path.endswith(".py") else line
            if path.endswith(".py") and pat.search(code) and ("import" in code or "oracle(" in code or "CDLL" in code):
                raise AssertionError("%s:%d uses the oracle: %s" % (path, ln, line.strip()))
            if not path.endswith(".py") and re.search(r'#\s*include\s*[<"][^>"]*oracle', code):
                raise AssertionError("%s:%d includes an oracle header" % (path, ln))
    # bench.py: every use sits inside cpu_baseline()
    src = open(os.path.join(ROOT, "bench.py")).read()
    rest = src
    bodies = re.findall(r"^def cpu_baseline\w*\(.*?(?=^def |\Z)", src, flags=re.S | re.M)       # cpu_baseline, cpu_baseline_parse
    assert bodies
    for body in bodies:
        rest = rest.replace(body, "")
    for ln, line in enumerate(rest.splitlines(), 1):
        code = line.split("#")[0]
        assert not (pat.search(code) and ("import" in code or "oracle(" in code or "CDLL" in code)), "bench.py outside cpu_baseline: " + line.strip()
    # the shipped library does not link it
    import subprocess
    so = os.path.join(ROOT, "hevcbitstream_amd", "libhevcbitstream_amd.so")
    needed = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
    assert "oracle" not in needed and "hevcref" not in needed
