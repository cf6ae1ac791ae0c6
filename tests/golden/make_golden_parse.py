#!/usr/bin/env python3
"""Generate tests/golden/parse_vectors.json from the REAL reference
(oracle/_ref/libhevcref.so, dev container only).

Input NALs come from tests/hevc_synth.py (our own header writer); the expected
values are what the reference's read_hevc_nal_unit (hevc_stream.c:155-240) leaves
in *h->nal / vps / sps / pps / sh and h->slice_data after each NAL of a
sequence, stored sparsely: [int32 index, value] pairs of the non-zero members.
Data only -- nothing of the reference's source is stored."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests import _orc  # noqa: E402
from tests.hevc_synth import Synth  # noqa: E402

out = []
for seed in range(1000, 1040):
    ref = _orc.ReferenceHevc()          # a fresh hevc_new() per sequence, as the tests do
    g = Synth(seed, rich=True)
    rng = np.random.RandomState(seed)
    seq = [g.vps(), g.sps_nal(int(rng.randint(64, 4096)), int(rng.randint(64, 2304))), g.pps_nal()]
    for k in range(7):
        t = int(rng.choice([0, 1, 8, 9, 16, 19, 20, 21]))
        seq.append(g.slice_nal(t, first=bool(rng.randint(0, 2)),
                               payload=rng.randint(0, 256, size=rng.randint(1, 60)).astype(np.uint8).tobytes(),
                               address=int(rng.randint(0, 100))))
        if k == 3 and rng.rand() < 0.5:
            seq.append(g.pps_nal())
    # NAL types the reference does not dispatch (AUD, SEI, EOS): rc -1, only *h->nal changes
    seq.append(bytes([35 << 1, 1, 0x50]))
    seq.append(bytes([39 << 1, 1, 1, 2, 3, 0x80]))
    steps = []
    for nal in seq:
        rc = ref.read(nal)
        snap = ref.snapshot()
        rec = {"nal": nal.hex(), "rc": rc, "structs": {}}
        for k, a in snap.items():
            nz = np.nonzero(a)[0]
            rec["structs"][k] = [[int(i), int(a[i])] for i in nz]
        size, data = ref.slice_data()
        t = (nal[0] >> 1) & 0x3F
        if rc >= 0 and (t <= 9 or 16 <= t <= 21):
            rec["slice_data"] = [size, hashlib.md5(data).hexdigest() if data is not None else None]
        steps.append(rec)
    out.append({"seed": seed, "steps": steps})
json.dump(out, open(os.path.join(HERE, "parse_vectors.json"), "w"), separators=(",", ":"))
print("sequences", len(out), "NALs", sum(len(s["steps"]) for s in out))
