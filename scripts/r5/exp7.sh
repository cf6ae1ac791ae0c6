cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05exp7; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parse.py tests/test_gpu_compact.py tests/test_gpu_index_parse.py tests/test_gpu_legacy.py -x -q 2>&1 | tail -4
timeout 300 python scripts/config3_time.py > $O/config3.txt 2>&1; tail -1 $O/config3.txt | cut -c1-600
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/tr3 -- python3 scripts/config3_time.py > $O/config3_prof.txt 2>&1
f=$(find $O/tr3 -name "*kernel_trace.csv" | head -1); python scripts/trace_call.py $f k_scan_prologue > $O/trace_config3.txt 2>&1; cat $O/trace_config3.txt | cut -c1-120; find $O/tr3 -type f -delete
# same-box A/B against round 4's library
for rep in 1 2; do
for L in "" build/variants/r04/libhbs.so; do
  HBS_LIB=$L timeout 200 python scripts/scan_time.py --nals 104857 --reps 10 > $O/s1.txt 2>&1; echo "lib=[$L] 1GiB: $(tail -1 $O/s1.txt | cut -c60-400)"
  HBS_LIB=$L timeout 200 python scripts/emit_time.py > $O/e1.txt 2>&1; echo "lib=[$L] emit 1GiB: $(tail -1 $O/e1.txt | cut -c1-200)"
done
done
for L in "" build/variants/r04/libhbs.so; do
  HBS_LIB=$L timeout 200 python scripts/scan_time.py --reps 6 > $O/s16.txt 2>&1; echo "lib=[$L] 16GiB: $(tail -1 $O/s16.txt | cut -c60-400)"
  for s in 512 1024; do HBS_LIB=$L timeout 300 python scripts/nal_sweep.py --gib 2 --sizes $s > $O/sw.txt 2>&1; echo "lib=[$L] $s: $(tail -1 $O/sw.txt | cut -c1-480)"; done
done
