#!/usr/bin/env python3
"""Generate tests/golden/l2_vectors.json and ten_nal.* from the REAL reference.

Run in the dev container only (needs oracle/_ref built from /root/reference by
`make -C oracle ref`).  The outputs are data: inputs (hex) and the reference's
answers.  Nothing of the reference's source is stored.

  find   : find_nal_unit(buf,size) -> [ret, nal_start, nal_end]   (h264_nal.c:38)
  n2r    : nal_to_rbsp -> [ret, nal_size_out, rbsp_size_out, rbsp hex|null] (h264_nal.c:147)
  r2n    : rbsp_to_nal -> [ret, nal hex]                          (h264_nal.c:92)

Buffers handed to find_nal_unit are followed by 0xFF bytes (the reference reads
up to three bytes past `size`; see oracle/hbs_oracle.h).
"""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests import _orc  # noqa: E402

ref = _orc.reference()
assert ref is not None, "build oracle/_ref first (make -C oracle ref)"

# SURVEY.md Appendix F (hand-picked edge cases) ...
FIND = """
00 00 01 40 01 02 03 04
00 00 00 01 40 01 02 03 04
11 22 00 00 01 40 01 02 03 04 05
00 00 01 40 01 02 00 00 01 41 42 43 44
00 00 01 40 01 02 00 00 00 01 41 42 43 44
00 00 01 40 01 02 00 00 01 41
00 00 01 40 01 02 00 00 01
00 00 01 40 01 02 00 00 00
00 00 01 40 01 02 00 00 00 00
11 22 33 00 00 01 40 41
11 22 33 44 00 00 01 40
00 00 01 40
00 00 01 00 00 01 40 41 42 43
00 00 00 00 00 01 40 41 42 43 44
11 22 33 44 55 66 77 88
00 00 01 40 00 00 03 01 02 00 00 01 55 66 77
00 00 01
00 00 00 01
00 00 00 00 01
40 00 00 00 01
40 41 00 00 00 01
00 00 01 40 00 00 00 01
00 00 01 40 41 00 00 00 01
00 00 01 40 41 00 00 00 00 01
00 00 01 40 00 00 01 41
00 00 01 40 41 00 00 01 42
00 00 01 40 41 42 00 00 01 43
00 00 00
00 00 00 00 00 00 00 00
00 00 01 00 00 00 40 41 42 43
00 00 01 40 00 00 00 55 66 00 00 01 41 42 43 44
"""
N2R = """
40 01 00 00 03 01 02
40 01 00 00 03 00 00 03 00 05
40 01 00 00 03
40 01 00 00 03 04
40 01 00 00 02
40 01 00 00 00
00 00 03 00 00 03 00
40 01 00 00 03 03 07
00 00
40 01 00 00 03 00
40 01 00 00 03 00 00 03
40 01 00 00 01
00 00 03
00 00 03 03
00 00 03 03 00 00 03
40
"""
R2N = """
00 00 00 00 00 01
00 00 00 00 01
00 00 00 01
00 00 04
00 00 03
00 00 00 00 00 00 00 05
07 00 00
00 00 02 00 00 01 00 00 00
00
"""


def hexlines(s):
    return [bytes.fromhex(l.replace(" ", "")) for l in s.strip().splitlines()]


rng = np.random.RandomState(20260102)
ALPHA = np.array([0, 0, 0, 0, 1, 1, 2, 3, 3, 4, 0x40, 0x80, 0xFF], dtype=np.uint8)


def dense(n):
    return bytes(ALPHA[rng.randint(0, len(ALPHA), size=n)])


find_cases = hexlines(FIND) + [dense(rng.randint(3, 40)) for _ in range(400)]
n2r_cases = hexlines(N2R) + [dense(rng.randint(1, 40)) for _ in range(400)]
r2n_cases = hexlines(R2N) + [dense(rng.randint(1, 40)) for _ in range(300)]

out = {"find": [], "n2r": [], "r2n": []}
for b in find_cases:
    out["find"].append([b.hex(), list(ref.find_nal_unit(b))])
for b in n2r_cases:
    r, ns, rs, data = ref.nal_to_rbsp(b)
    out["n2r"].append([b.hex(), [r, ns, rs, data.hex() if data is not None else None]])
for b in r2n_cases:
    r, ns, data = ref.rbsp_to_nal(b)
    out["r2n"].append([b.hex(), [r, data.hex()]])

with open(os.path.join(HERE, "l2_vectors.json"), "w") as f:
    json.dump(out, f, indent=0, separators=(",", ":"))
print("find %d  n2r %d  r2n %d" % (len(out["find"]), len(out["n2r"]), len(out["r2n"])))

# 10-NAL stream: index via the hevc_analyze loop and the
# reference CLI's stdout.
stream = open(os.path.join(HERE, "ten_nal.hevc"), "rb").read()
idx, p = [], 0
while True:
    r, s, e = ref.find_nal_unit(stream[p:])
    if r <= 0:
        if r == -1:
            idx.append([p + s, p + e, -1])
        break
    idx.append([p + s, p + e, r])
    p += e
json.dump(idx, open(os.path.join(HERE, "ten_nal.index.json"), "w"))
exe = os.path.join(ROOT, "oracle", "_ref", "hevc_analyze_ref")
txt = subprocess.run([exe, os.path.join(HERE, "ten_nal.hevc")], stdout=subprocess.PIPE, check=True).stdout
open(os.path.join(HERE, "ten_nal.analyze.txt"), "wb").write(txt)
print("ten_nal: %d NALs, %d stdout lines" % (len(idx), txt.count(b"\n")))
