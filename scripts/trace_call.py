"""dev aid: from a rocprofv3 kernel-trace csv, the timeline of the LAST call of a kernel sequence: start offsets and durations (us)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2]          # name fragment of the kernel that begins a call
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
i0 = starts[-2] if len(starts) > 1 else starts[-1]
i1 = starts[-1] if len(starts) > 1 else len(rows)
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f  +gap %6.1f  dur %8.1f  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, r["Kernel_Name"].split("(")[0][:60]))
    prev_end = e
print("call span %.1f us" % ((prev_end - t0) / 1e3))
