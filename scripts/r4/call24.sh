cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_scan.py tests/test_gpu_emit.py tests/test_gpu_fullsize.py -x -q > $O/pytest_call24.txt 2>&1; grep -n "passed\|failed" $O/pytest_call24.txt | tail -3 | cut -c1-300
timeout 300 python scripts/emit_time.py > $O/emit_time_1GiB.txt 2>&1; tail -2 $O/emit_time_1GiB.txt
timeout 300 python scripts/scan_time.py --nals 104857 > $O/scan_time_1GiB.txt 2>&1; tail -1 $O/scan_time_1GiB.txt | cut -c1-500
timeout 900 python scripts/nal_sweep.py --gib 2 --sizes 512,1024,2048,10240 > $O/nal_sweep_mid.txt 2>&1; tail -4 $O/nal_sweep_mid.txt | cut -c1-420
