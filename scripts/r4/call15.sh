cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_b.txt 2>&1; tail -3 $O/pytest_gpu_b.txt
timeout 300 python tests/tools/cli_time.py > $O/cli_time.txt 2>&1; cat $O/cli_time.txt
bash scripts/prof_r04.sh > $O/prof_r04.log 2>&1; tail -5 $O/prof_r04.log
