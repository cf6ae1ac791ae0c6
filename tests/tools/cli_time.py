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

# what of that is start-up (process, HIP, context): the same binaries on a 10-NAL file
small = "tests/golden/ten_nal.hevc"
for exe in ("oracle/_ref/hevc_analyze_ref", "oracle/_ref/hevc_analyze_amd"):
    if not os.path.exists(exe) or not os.path.exists(small):
        continue
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        subprocess.run([exe, small], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
        best = min(best, time.perf_counter() - t0)
    print("%-32s 10 NALs (start-up): %.3f s" % (exe, best))
for mode in ("batch", "HBS_LEGACY_NO_BATCH=1"):
    env = dict(os.environ)
    if mode != "batch":
        env["HBS_LEGACY_NO_BATCH"] = "1"
    env["HBS_LEGACY_TIMING"] = "1"
    best, inner = 1e9, 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        r = subprocess.run(["oracle/_ref/hevc_analyze_amd", path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        best = min(best, time.perf_counter() - t0)
        for line in r.stderr.decode().splitlines():
            if "between the context being ready" in line:
                inner = min(inner, float(line.split(":")[1].split()[0]))
    print("hevc_analyze_amd, %s: %.3f s for %d NALs, of which %.4f s behind the GPU start-up" % (mode, best, n, inner))
