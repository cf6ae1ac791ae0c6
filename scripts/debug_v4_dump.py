"""Diagnostic build: dump tile 0's flag count, row counts and element list of the event-sparse kernel
and compare with the chunks that hold two adjacent zero bytes (dev aid)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, ".")
so = "build/diag/libhbs_diag.so"
import hevcbitstream_amd.api as api
api.library_path = lambda: so
import hevcbitstream_amd as hbs
ctx = hbs.Context(0); ctx.set_kernel(4)
rng = np.random.RandomState(7)
n = 100000
s = rng.randint(1, 256, size=n).astype(np.uint8)
s = rng.randint(1, 256, size=n).astype(np.uint8)
for at in rng.randint(0, n - 8, size=50):
    s[at:at + 4] = np.frombuffer(b"\x00\x00\x01\x42" if at % 3 else b"\x00\x00\x03\x01", dtype=np.uint8)
d = torch.from_numpy(s).cuda()
ctx.index_extract(d, want_rbsp=False)
out = np.zeros(4096, dtype=np.uint32)
lib = api.load_library()
lib.hbs_debug_dump4.argtypes = [C.c_void_p]
assert lib.hbs_debug_dump4(out.ctypes.data) == 0
z = (s[:-1] == 0) & (s[1:] == 0)
pairs = np.nonzero(z)[0]
exp = sorted(set(int(c) for p in pairs for c in ((p + 2) // 16, (p + 3) // 16, (p + 1) // 16) ))   # rough: chunks near a pair
print("nflag", out[0], "pairs", len(pairs))
print("row_cnt nonzero rows", np.nonzero(out[16:16 + 128])[0][:40], out[16:16+128].sum())
print("list", out[256:256 + int(min(out[0], 80))])
print("pair chunks", sorted(set(int(p) // 16 for p in pairs))[:80])
print("myf nonzero tids", [(int(t), hex(int(out[2048 + t]))) for t in np.nonzero(out[2048:2048 + 256])[0][:40]])
