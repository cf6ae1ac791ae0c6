"""HBM traffic per launch of the fused kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).
usage: pmc_traffic.py <fetch_dir> <write_dir> <kernel-substring> <algorithmic_bytes> [out.json] [near_max]
near_max: the profiled command also ran the kernel on smaller inputs (bench.py's config-3 leg): keep only the launches whose
counter is within 10 % of the largest one, i.e. those on the 16 GiB workload; the dropped ones stay listed in raw_KiB.
Counters are in KiB; FETCH_SIZE is doubled per the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md."""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hevcbitstream_amd

def is_kernel(kernel_name, sub):
    """`sub` as a whole identifier inside the kernel's name: k_scan_extract4 is not k_scan_extract4_r24 (round 6), but k3_tiles is k3_tiles<0>"""
    i = kernel_name.find(sub)
    while i >= 0:
        nxt = kernel_name[i + len(sub): i + len(sub) + 1]
        if not (nxt.isalnum() or nxt == "_"):
            return True
        i = kernel_name.find(sub, i + 1)
    return False


def rows(d, sub, name):
    out = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if is_kernel(r["Kernel_Name"], sub) and r["Counter_Name"] == name:
                out.append(r)
    by = {}
    for r in out:
        by.setdefault(r["Dispatch_Id"], 0.0)
        by[r["Dispatch_Id"]] += float(r["Counter_Value"])
    meta = out[0] if out else {}
    return list(by.values()), meta

fd, wd, sub, algo = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
f_all, fm = rows(fd, sub, "FETCH_SIZE")
w_all, wm = rows(wd, sub, "WRITE_SIZE")
near_max = len(sys.argv) > 6 and sys.argv[6] == "near_max"
f = [x for x in f_all if not near_max or x >= 0.9 * max(f_all)]
w = [x for x in w_all if not near_max or x >= 0.9 * max(w_all)]
fetch = sum(f) / len(f) * 1024 * 2
write = sum(w) / len(w) * 1024
res = {"kernel_substring": sub, "launches": {"fetch_pass": len(f), "write_pass": len(w)},
       "fetch_bytes_per_launch": int(fetch), "write_bytes_per_launch": int(write),
       "traffic_bytes_per_launch": int(fetch + write), "algorithmic_bytes_per_launch": algo,
       "ratio_traffic_over_algorithmic": round((fetch + write) / algo, 4),
       "raw_KiB": {"FETCH_SIZE": f_all, "WRITE_SIZE": w_all},
       "launches_kept": "within 10 % of the largest (the 16 GiB workload)" if near_max else "all",
       "source_sha256": hevcbitstream_amd.source_digest(),     # of hevcbitstream_amd/csrc: bench.py quotes this file only while it matches
       "dispatch": {k: fm.get(k) for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count")},
       "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (no trace domains); KiB units; FETCH_SIZE x2 (gfx950)"}
js = json.dumps(res, indent=1)
print(js)
if len(sys.argv) > 5:
    open(sys.argv[5], "w").write(js + "\n")
