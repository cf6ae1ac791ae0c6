#!/bin/bash
# Effective GPU clock of K12 per process (guide: GRBM_GUI_ACTIVE / 8 XCDs / kernel wall time): is the process-to-process spread
# of the kernel's time (5.9 ... 6.2 ms) the clock?   run on the GPU box from the repo root; writes gpurun_out/r03d/clock_probe.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d; mkdir -p $O; : > $O/clock_probe.txt
for i in 1 2 3 4 5 6; do
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/clk$i -- python3 scripts/experiments/k12_time_unchecked.py > $O/clk_line$i.txt 2>/dev/null
  f=$(find $O/clk$i -name "*counter_collection.csv" | head -1)
  python3 - "$f" $i >> $O/clock_probe.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_scan_extract4" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
by = {}
for r in rows:
    by.setdefault(r["Dispatch_Id"], [0.0, r]); by[r["Dispatch_Id"]][0] += float(r["Counter_Value"])
out = []
for d, (v, r) in by.items():
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) if "End_Timestamp" in r else 0
    out.append((v, dur))
print("process", sys.argv[2], " ".join("cycles/8=%.3fM dur=%.3fms clock=%.3fGHz" % (v / 8 / 1e6, dur / 1e6, (v / 8 / dur) if dur else 0) for v, dur in out[1:]))
PY
  tail -1 $O/clk_line$i.txt >> $O/clock_probe.txt
  find $O/clk$i -type f -delete
done
cat $O/clock_probe.txt
