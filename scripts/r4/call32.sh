cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_scan.py tests/test_gpu_pair.py -x -q > $O/pytest_scan.txt 2>&1; grep -n "passed\|failed" $O/pytest_scan.txt | tail -2 | cut -c1-300
for i in 1 2 3; do HBS_PAIR_DEBUG=1 timeout 300 python scripts/r4/pair_time.py 2> $O/pair_dbg_$i.err | tail -2 | cut -c1-600; grep -c "not needed" $O/pair_dbg_$i.err; done
timeout 600 python3 bench.py --steps 5 --warmup 1 --cpu-sample-nals 0 --other-kernels 0 2>/dev/null | cut -c1-1400
