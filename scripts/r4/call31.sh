cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_c.txt 2>&1; tail -5 $O/pytest_gpu_c.txt | cut -c1-400
