S=$(date +%s)
python3 bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
echo "rc $? elapsed $(( $(date +%s) - S )) s"
tail -3 gpurun_out/bench_final.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_final.json").read().strip().splitlines()[-1])
r=d["roofline"]; print(d["value"], r["frac"], r["kernel_ms"], r["traffic"])
c=d["other_kernels"]["config3_end_to_end"]; print(c.get("cpu_baseline_at_full_size")); print(c.get("parity"))
PY
