cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_legacy.py tests/test_gpu_pair.py -x -q > $O/pytest_legacy.txt 2>&1; tail -15 $O/pytest_legacy.txt
timeout 300 python tests/tools/cli_time.py > $O/cli_time.txt 2>&1; cat $O/cli_time.txt
timeout 900 python tests/tools/fuzz_gpu_legacy.py > $O/fuzz_gpu_legacy.txt 2>&1; tail -3 $O/fuzz_gpu_legacy.txt
