cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_emit.py tests/test_gpu_legacy.py -x -q > $O/pytest_emit.txt 2>&1; tail -5 $O/pytest_emit.txt
timeout 900 python scripts/nal_sweep.py --sizes 64,128,256,384,512,640,768,1024 > $O/nal_sweep_tiny.txt 2>&1; cat $O/nal_sweep_tiny.txt | cut -c1-400
timeout 300 python tests/tools/cli_time.py > $O/cli_time.txt 2>&1; tail -3 $O/cli_time.txt
timeout 600 python tests/tools/soak_emit_small.py > $O/soak_emit_small.txt 2>&1; tail -3 $O/soak_emit_small.txt
