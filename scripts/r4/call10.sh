cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_a.txt 2>&1; tail -5 $O/pytest_gpu_a.txt
timeout 900 python bench.py > $O/bench_a.json 2> $O/bench_a.err; tail -c 6000 $O/bench_a.json; tail -5 $O/bench_a.err
