cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_pair.py -x -q > $O/pytest_pair.txt 2>&1; tail -15 $O/pytest_pair.txt
for i in 1 2 3 4 5 6 7 8 9; do timeout 300 python scripts/r4/pair_time.py >> $O/pair_time.txt 2>&1; done
cat $O/pair_time.txt
timeout 900 python -m pytest tests/test_gpu_scan.py -x -q > $O/pytest_scan.txt 2>&1; tail -3 $O/pytest_scan.txt
timeout 300 python scripts/scan_time.py --nals 104857 > $O/scan_time_1GiB.txt 2>&1; tail -5 $O/scan_time_1GiB.txt
