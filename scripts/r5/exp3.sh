# round 5 experiment 3: index-only first tile by block number (A/B against tickets for every tile); legacy batch debug
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05exp3; mkdir -p $O
HBS_LEGACY_DEBUG=1 timeout 600 python -m pytest "tests/test_gpu_legacy.py::test_batch_loop_with_deviating_callers" -x -q -s > $O/pytest_legacy.txt 2>&1; tail -40 $O/pytest_legacy.txt | cut -c1-200
for rep in 1 2; do
for v in new old; do
  for r in 0 256 512; do
    L=""; [ $v = old ] && L="build/variants/tk_old/libhbs.so"
    HBS_LIB=$L HBS5_TILE_ROWS=$r timeout 200 python scripts/scan_time.py --nals 104857 --reps 10 > $O/idx_${v}_r$r.txt 2>&1
    echo "$v rows $r 1GiB: $(tail -1 $O/idx_${v}_r$r.txt | cut -c230-400)"
  done
done
done
HBS_LIB=build/variants/tk_old/libhbs.so timeout 200 python scripts/scan_time.py --reps 6 > $O/idx16_old.txt 2>&1; echo "old 16GiB: $(tail -1 $O/idx16_old.txt | cut -c230-400)"
timeout 200 python scripts/scan_time.py --reps 6 > $O/idx16_new.txt 2>&1; echo "new 16GiB: $(tail -1 $O/idx16_new.txt | cut -c230-400)"
timeout 300 python -m pytest tests/test_gpu_scan.py -x -q 2>&1 | tail -2
