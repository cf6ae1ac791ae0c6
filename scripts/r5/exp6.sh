cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05exp6; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ingest.py tests/test_gpu_pair.py tests/test_gpu_emit.py tests/test_gpu_legacy.py -x -q 2>&1 | tail -4
timeout 300 python scripts/config3_time.py > $O/config3.txt 2>&1; tail -1 $O/config3.txt | cut -c1-600
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/tr3 -- python3 scripts/config3_time.py > $O/config3_prof.txt 2>&1
f=$(find $O/tr3 -name "*kernel_trace.csv" | head -1); python scripts/trace_call.py $f k_scan_prologue > $O/trace_config3.txt 2>&1; cat $O/trace_config3.txt | cut -c1-120; find $O/tr3 -type f -delete
for s in 384 512 1024; do
  timeout 300 python scripts/nal_sweep.py --gib 2 --sizes $s > $O/sw_$s.txt 2>&1; echo "auto $s: $(tail -1 $O/sw_$s.txt | cut -c1-500)"
  HBS5_TILE_ROWS=256 timeout 300 python scripts/nal_sweep.py --gib 2 --sizes $s > $O/sw256_$s.txt 2>&1; echo "r256 $s: $(tail -1 $O/sw256_$s.txt | cut -c1-500)"
done
timeout 300 python scripts/config_1gib.py > $O/cfg.txt 2>&1; python - <<PY
import json
d=json.loads(open("$O/cfg.txt").read().strip().splitlines()[-1])
print("extract k %.4f call %.4f | idx k %.4f call %.4f | emit call %.4f tiles-only %.4f" % (d["config2_extract"]["kernel_ms"], d["config2_extract"]["call_ms"], d["config2_index_only"]["kernel_ms"], d["config2_index_only"]["call_ms"], d["config4_emit"]["call_ms"], d["config4_emit"]["tiles_only_path"]["call_ms"]))
PY
