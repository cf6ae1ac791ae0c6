"""Reproduce the two inputs on which the long soak of round 6 (tests/tools/soak_gpu.py 1200 66) found the arena-tile emit path differing
from the oracle: iterations 47568 and 50920.  usage: python3 scripts/experiments/repro_emit_tiles.py [iter ...]"""
import importlib.util, sys
import numpy as np, torch
sys.path.insert(0, ".")
import hevcbitstream_amd as hbs
from tests import _orc

# make_stream of the soak tool, without running its loop
src = open("tests/tools/soak_gpu.py").read()
ns = {}
exec(src[src.index("ALPHA = "):src.index("t_end = time.time()")], {"np": np}, ns)
make_stream = ns["make_stream"]
make_stream.__globals__.update(ns)

orc = _orc.oracle()
iters = [int(x) for x in sys.argv[1:]] or [47568, 50920]
for it in iters:
    rng = np.random.default_rng(66 * 100003 + it)
    s = make_stream(rng)
    want_idx, want_arena, why = orc.index_extract(s)
    keep = want_idx[(want_idx["status"] & 1) == 0]
    full = want_idx.copy(); full["status"] = 0
    print("iter", it, "stream", len(s), "nals", len(want_idx), "kept", len(keep), "arena", len(want_arena))
    for name, index, gap in (("kept", keep, 0), ("all", full, 1)):
        want = orc.emit_annexb(want_arena, index) if gap == 0 else None
        outs = {}
        for path in (-1, 0, 1, 2):
            c = hbs.Context(0)
            c.set_emit_path(path)
            got, ent = c.emit_annexb(torch.from_numpy(want_arena.copy()).cuda(), index, gap_mode=gap)
            outs[path] = (got, ent)
            c.close()
        ref = want if want is not None else outs[0][0]
        for path, (got, ent) in outs.items():
            same = np.array_equal(got, ref)
            msg = "  %s path %2d: %s len %d (ref %d)" % (name, path, "ok" if same else "DIFFERS", len(got), len(ref))
            if not same:
                n = min(len(got), len(ref))
                d = np.nonzero(got[:n] != ref[:n])[0]
                first = int(d[0]) if len(d) else n
                e0 = outs[0][1]
                k = int(np.searchsorted(e0["start"], first, side="right")) - 1
                msg += " first diff at %d (%d differing), in NAL %d: rbsp_off %d rbsp_len %d out start %d end %d" % (
                    first, len(d), k, int(index["rbsp_off"][k]), int(index["rbsp_len"][k]), int(e0["start"][k]), int(e0["end"][k]))
                msg += "\n      got %s\n      ref %s" % (got[max(0, first - 8): first + 12].tolist(), ref[max(0, first - 8): first + 12].tolist())
                msg += "\n      entries equal: %s" % all(np.array_equal(ent[f], e0[f]) for f in ent.dtype.names)
            print(msg)
