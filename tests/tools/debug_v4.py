"""Find the smallest stream on which a kernel variant and the oracle disagree (dev aid)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import hevcbitstream_amd as hbs
from tests import _orc
orc = _orc.oracle()
k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ctx = hbs.Context(0); ctx.set_kernel(k)
rng = np.random.RandomState(7)
def run(s, want_rbsp=True):
    d = torch.from_numpy(s).cuda()
    return ctx.index_extract(d, want_rbsp=want_rbsp)
for n, dens in ((100000, 0), (100000, 50), (140000, 50), (300000, 50), (300000, 300), (300000, 1000), (2000000, 300), (2000000, 3000)):
    s = rng.randint(1, 256, size=n).astype(np.uint8)      # no zero bytes at all
    for at in rng.randint(0, n - 8, size=dens):
        s[at:at + 4] = np.frombuffer(b"\x00\x00\x01\x42" if at % 3 else b"\x00\x00\x03\x01", dtype=np.uint8)
    want_idx, want_arena, why = orc.index_extract(s)
    got_idx, got_arena, summ = run(s, want_rbsp=False)
    ok_idx = len(got_idx) == len(want_idx) and all(np.array_equal(got_idx[f], want_idx[f]) for f in ("start", "end", "rbsp_off", "rbsp_len", "status"))
    print("n", n, "patterns", dens, "index-only:", "OK" if ok_idx else "MISMATCH", "nals", len(got_idx), len(want_idx), "rbsp_bytes", int(summ["rbsp_bytes"]), "err", int(summ["error"]))
    if not ok_idx:
        for f in ("start", "end", "rbsp_off", "rbsp_len", "status"):
            m = min(len(got_idx), len(want_idx))
            bad = np.nonzero(got_idx[f][:m] != want_idx[f][:m])[0]
            if len(bad): print("   field", f, "first bad nal", bad[0], "got", got_idx[f][bad[0]], "want", want_idx[f][bad[0]], "at start", want_idx["start"][bad[0]])
        break

# arena comparison on the first small case that has patterns
rng = np.random.RandomState(7)
n = 100000
s = rng.randint(1, 256, size=n).astype(np.uint8)
s = rng.randint(1, 256, size=n).astype(np.uint8)
for at in rng.randint(0, n - 8, size=50):
    s[at:at + 4] = np.frombuffer(b"\x00\x00\x01\x42" if at % 3 else b"\x00\x00\x03\x01", dtype=np.uint8)
want_idx, want_arena, why = orc.index_extract(s)
got_idx, got_arena, summ = run(s, want_rbsp=True)
m = min(len(got_arena), len(want_arena))
bad = np.nonzero(got_arena[:m] != want_arena[:m])[0]
print("arena lens", len(got_arena), len(want_arena), "first bad", bad[:3])
if len(bad):
    b = int(bad[0])
    k = int(np.searchsorted(want_idx["rbsp_off"], b, side="right") - 1)
    print("in nal", k, "start", want_idx["start"][k], "rbsp_off", want_idx["rbsp_off"][k], "offset in nal", b - int(want_idx["rbsp_off"][k]))
    approx = int(want_idx["start"][k]) + b - int(want_idx["rbsp_off"][k])
    print("stream around", approx, bytes(s[approx - 24:approx + 24]).hex())
    print("got ", bytes(got_arena[b - 8:b + 16]).hex())
    print("want", bytes(want_arena[b - 8:b + 16]).hex())
