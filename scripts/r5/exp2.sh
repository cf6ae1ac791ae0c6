# round 5 experiment 2: index-only with per-stream tile height + fused aggregate launch; the new legacy tests; 1 GiB config lines
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05exp2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_scan.py tests/test_gpu_legacy.py tests/test_gpu_index_parse.py tests/test_gpu_fullsize.py -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for r in 0 256 352 512 176; do
  if [ $r = 0 ]; then timeout 200 python scripts/scan_time.py --nals 104857 --reps 10 > $O/idx_r$r.txt 2>&1
  else HBS5_TILE_ROWS=$r timeout 200 python scripts/scan_time.py --nals 104857 --reps 10 > $O/idx_r$r.txt 2>&1; fi
  echo "rows $r 1GiB: $(tail -1 $O/idx_r$r.txt | cut -c230-400)"
done
timeout 200 python scripts/scan_time.py --nals 209715 --reps 10 > $O/idx_2g.txt 2>&1; echo "auto 2GiB: $(tail -1 $O/idx_2g.txt | cut -c230-400)"
HBS5_TILE_ROWS=256 timeout 200 python scripts/scan_time.py --nals 209715 --reps 10 > $O/idx_2g_256.txt 2>&1; echo "256 2GiB: $(tail -1 $O/idx_2g_256.txt | cut -c230-400)"
timeout 200 python scripts/scan_time.py --reps 6 > $O/idx_16g.txt 2>&1; echo "auto 16GiB: $(tail -1 $O/idx_16g.txt | cut -c230-400)"
timeout 300 python scripts/config_1gib.py > $O/config_1gib.txt 2>&1; tail -1 $O/config_1gib.txt | cut -c1-1500
