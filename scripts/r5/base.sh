# round 5 baseline: GPU suite + 1 GiB call timings on the round-4 sources
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05base; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?"
tail -3 $O/pytest_gpu.txt
timeout 300 python scripts/scan_time.py --nals 104857 > $O/scan_time_1GiB.txt 2>&1; tail -1 $O/scan_time_1GiB.txt
timeout 300 python scripts/emit_time.py > $O/emit_time_1GiB.txt 2>&1; tail -2 $O/emit_time_1GiB.txt
