cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_emit.py -x -q > $O/pytest_emit.txt 2>&1; tail -25 $O/pytest_emit.txt | cut -c1-400
