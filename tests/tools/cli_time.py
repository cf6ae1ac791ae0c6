"""Wall time of the reference's CLI, unmodified, on the reference library (CPU) and on this library (GPU) (dev aid)."""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, ".")
from tests.hevc_synth import stream_4k30
stream, n = stream_4k30(11, n_pictures=250, slices_per_picture=8, idr_every=60, payload_bytes=(2000, 9000))
path = "/tmp/cli_time.hevc"
open(path, "wb").write(stream)
for exe in ("oracle/_ref/hevc_analyze_ref", "oracle/_ref/hevc_analyze_amd"):
    if not os.path.exists(exe):
        print(exe, "missing"); continue
    best = 1e9
    out_len = 0
    for _ in range(2):
        t0 = time.perf_counter()
        r = subprocess.run([exe, path], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        best = min(best, time.perf_counter() - t0)
        out_len = len(r.stdout)
    print("%-32s %d NALs, %.1f MB: %.2f s (%.0f NAL/s), stdout %d bytes, rc %d" % (exe, n, len(stream) / 1e6, best, n / best, out_len, r.returncode))
