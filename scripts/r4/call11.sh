cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
HBS_PAIR_DEBUG=1 timeout 600 python -m pytest tests/test_gpu_pair.py -x -q -s > $O/pytest_pair.txt 2>&1; tail -5 $O/pytest_pair.txt
rm -f $O/pair_time.txt $O/pair_time.err
for i in 1 2 3 4 5 6; do timeout 300 python scripts/r4/pair_time.py >> $O/pair_time.txt 2>> $O/pair_time.err; done
cat $O/pair_time.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_a.txt 2>&1; tail -5 $O/pytest_gpu_a.txt
timeout 900 python bench.py > $O/bench_a.json 2> $O/bench_a.err; tail -c 3000 $O/bench_a.json; tail -5 $O/bench_a.err
