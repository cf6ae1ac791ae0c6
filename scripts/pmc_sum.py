"""Sum rocprofv3 --pmc counter rows per kernel and counter.  usage: pmc_sum.py <dir> [kernel-substring]"""
import csv, glob, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else "k_scan_extract"
acc = collections.defaultdict(float); launches = collections.defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); launches[r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(acc):
    print("%-24s %14.4g per launch" % (k, acc[k] / max(1, len(launches[k]))))
