# round 5 experiment 4: compact parse tests; legacy batch debug; index-only strided geometry; K12 / K3 first tile by number (A/B)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05exp4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_compact.py -x -q > $O/pytest_compact.txt 2>&1; tail -25 $O/pytest_compact.txt | cut -c1-250
HBS_LEGACY_DEBUG=1 timeout 600 python -m pytest "tests/test_gpu_legacy.py::test_batch_loop_with_deviating_callers" -x -q > $O/pytest_legacy.txt 2>&1; grep -n "AssertionError\|passed\|failed" $O/pytest_legacy.txt | cut -c1-600
timeout 900 python -m pytest tests/test_gpu_scan.py tests/test_gpu_emit.py tests/test_gpu_parse.py tests/test_gpu_index_parse.py -x -q 2>&1 | tail -3
for rep in 1 2; do
  timeout 300 python scripts/config_1gib.py > $O/cfg_new_$rep.txt 2>&1
  HBS_LIB=build/variants/first_tk/libhbs.so timeout 300 python scripts/config_1gib.py > $O/cfg_old_$rep.txt 2>&1
  for v in new old; do python - <<PY
import json
d=json.loads(open("$O/cfg_${v}_$rep.txt").read().strip().splitlines()[-1])
print("$v $rep", "extract k %.4f call %.4f | idx k %.4f call %.4f | emit call %.4f" % (d["config2_extract"]["kernel_ms"], d["config2_extract"]["call_ms"], d["config2_index_only"]["kernel_ms"], d["config2_index_only"]["call_ms"], d["config4_emit"]["call_ms"]))
PY
  done
done
timeout 200 python scripts/scan_time.py --nals 209715 --reps 10 > $O/idx_2g.txt 2>&1; echo "2GiB: $(tail -1 $O/idx_2g.txt | cut -c60-400)"
timeout 200 python scripts/scan_time.py --nals 314572 --reps 10 > $O/idx_3g.txt 2>&1; echo "3GiB: $(tail -1 $O/idx_3g.txt | cut -c60-400)"
timeout 200 python scripts/scan_time.py --reps 6 > $O/idx_16g.txt 2>&1; echo "16GiB: $(tail -1 $O/idx_16g.txt | cut -c60-400)"
HBS_LIB=build/variants/first_tk/libhbs.so timeout 200 python scripts/scan_time.py --reps 6 > $O/idx_16g_old.txt 2>&1; echo "16GiB old: $(tail -1 $O/idx_16g_old.txt | cut -c60-400)"
