cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05exp5; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; tail -12 $O/pytest_gpu.txt | cut -c1-300
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -3 $O/bench.err | cut -c1-300
python - <<PY
import json
try:
    d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
    print("value", d["value"], "frac", d["roofline"]["frac"], "placement", json.dumps(d["roofline"].get("placement"))[:700])
    c=d["other_kernels"]["configs_1GiB"]
    for k in ("config2_extract","config2_index_only","config4_emit"):
        print(k, {x:c[k].get(x) for x in ("kernel_ms","call_ms")}, c[k]["roofline"]["frac"], c[k]["roofline"].get("frac_of_call"))
    print("parse", d["other_kernels"]["parse_headers"]["value"], "config3", d["other_kernels"]["config3_end_to_end"]["ms"], d["other_kernels"]["config3_end_to_end"]["without_arena"]["ms"])
    print("emit", d["other_kernels"]["emit_annexb"]["ms"], d["other_kernels"]["emit_annexb"]["hbm_traffic_GBs"])
except Exception as e:
    print("parse error", e)
PY
