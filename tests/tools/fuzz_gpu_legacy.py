"""read_hevc_nal_unit NAL by NAL (the legacy symbol on the GPU) against the oracle's sequential parser on corrupted
sequences, parameter sets included (dev aid).  usage: python3 tests/tools/fuzz_gpu_legacy.py [first_seed] [count]"""
import sys
import numpy as np
sys.path.insert(0, ".")
import hevcbitstream_amd as hbs
from tests import _orc
from tests.test_gpu_legacy import LegacyHevc
from tests.test_sim_parse_logic import broken, sequence
first = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 400
lib = hbs.load_library()
bad = nals_total = 0
for seed in range(first, first + count):
    for variant, which in ((2, lambda t: True), (1, lambda t: t not in (33, 34))):
        nals = broken(sequence(seed), np.random.RandomState(7 * seed + variant), which)
        lib.hbs_legacy_reset_tables()
        ours, orc_p = LegacyHevc(lib), _orc.OracleHevc()
        for k, nal in enumerate(nals):
            ra, rb = ours.read(nal), orc_p.read(nal)
            a, b = ours.snapshot(), orc_p.snapshot()
            t = (nal[0] >> 1) & 0x3F
            kind = "sh" if (t <= 9 or 16 <= t <= 21) else {32: "vps", 33: "sps", 34: "pps"}.get(t)
            ok = ra == rb and np.array_equal(a["nal"], b["nal"]) and (kind is None or np.array_equal(a[kind], b[kind]))
            if ok and kind == "sh" and ra >= 0:
                ok = ours.slice_data() == orc_p.slice_data()
            nals_total += 1
            if not ok:
                bad += 1
                print("MISMATCH seed", seed, "variant", variant, "NAL", k, "type", t, "rc", ra, rb)
                break
        orc_p.close()
print("seeds", first, "..", first + count - 1, "NALs", nals_total, "mismatching sequences", bad)
