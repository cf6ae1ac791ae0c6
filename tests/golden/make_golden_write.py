#!/usr/bin/env python3
"""Golden vectors for the syntax writers: the reference's write_hevc_nal_unit (hevc_stream.c:1249-1327)
run in this container on parsed (and on edited) structs.  Needs oracle/_ref/libhevcref.so.

For every NAL of a few synthetic sequences: read it with the reference, optionally edit a field of the
parsed struct, write it back with the reference into a buffer of `size` bytes; recorded: the bytes
written, the return value and what the writer left in h->slice_data->rbsp_size.

usage: python tests/golden/make_golden_write.py   ->  tests/golden/write_vectors.json.gz
"""
import ctypes as C
import gzip
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests import _orc                                           # noqa: E402
from tests._parsecmp import which_struct                         # noqa: E402
from tests.test_sim_parse_logic import sequence                  # noqa: E402

EDITS = {  # struct kind -> [(field, delta)] applied to every second NAL of that kind
    "sps": [("pic_width_in_luma_samples", 16), ("log2_max_pic_order_cnt_lsb_minus4", 1)],
    "pps": [("init_qp_minus26", -3), ("pps_cb_qp_offset", 2)],
    "sh": [("slice_qp_delta", 5), ("slice_type", 0)],
    "vps": [("vps_max_layer_id", 1)],
}


def field_index(kind, name):
    for n, i, c in _orc.flat_fields(_orc.STRUCT_TYPES[kind]):
        if n == name:
            return i
    raise KeyError(name)


def one(seed):
    if True:
        nals = sequence(seed)
        r = _orc.ReferenceHevc()
        steps = []
        seen = {}
        for nal in nals:
            rc = r.read(nal)
            t = int(r.v["nal"][1])
            kind = which_struct(t)
            step = {"nal": bytes(nal).hex(), "read_rc": rc}
            if rc >= 0 and kind is not None:
                seen[kind] = seen.get(kind, 0) + 1
                edits = []
                saved = r.v[kind].copy()
                if seen[kind] % 2 == 0:
                    for name, delta in EDITS[kind]:
                        i = field_index(kind, name)
                        new = int(r.v[kind][i]) + delta if name != "slice_type" else 2       # I slice: no ref lists
                        r.v[kind][i] = new
                        edits.append([name, new])
                size = 2 * len(nal) + 64
                out = np.zeros(size + 16, dtype=np.uint8)
                wrc = r.L.write_hevc_nal_unit(r.h, out.ctypes.data_as(C.POINTER(C.c_uint8)), size)
                r.v[kind][:] = saved          # the edit is for this write only: later NALs are read against the stream's own structs
                step.update({"edits": edits, "size": size, "write_rc": int(wrc),
                             "out": bytes(out[:max(wrc, 0)]).hex(), "slice_data_size": int(r.slice_data()[0])})
            steps.append(step)
        return {"seed": seed, "steps": steps}


def main():
    import subprocess
    if len(sys.argv) > 2 and sys.argv[1] == "--seed":
        print(json.dumps(one(int(sys.argv[2]))))
        return
    vectors = []
    for seed in range(1, 40):
        # the reference's writer can crash on streams outside its envelope: one process per sequence
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--seed", str(seed)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        if p.returncode == 0:
            vectors.append(json.loads(p.stdout.decode().strip().splitlines()[-1]))
        if len(vectors) == 10:
            break
    with gzip.open(os.path.join(HERE, "write_vectors.json.gz"), "wt") as f:
        json.dump(vectors, f)
    print("sequences", len(vectors), "written NALs", sum(1 for v in vectors for s in v["steps"] if "write_rc" in s))


if __name__ == "__main__":
    main()
