cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_emit.py tests/test_gpu_scan.py -x -q > $O/pytest_emit.txt 2>&1; grep -n "passed\|failed" $O/pytest_emit.txt | tail -2 | cut -c1-300
timeout 900 python scripts/nal_sweep.py --gib 2 --sizes 256,512,1024,2048,10240 2>&1 | tail -5 | cut -c1-100,230-420
HBS_EMIT_NALS=1677000 timeout 300 python scripts/emit_time.py 2>&1 | tail -2
timeout 300 python scripts/emit_time.py 2>&1 | tail -2
