cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
HBS_PAIR_DEBUG=1 timeout 600 python -m pytest tests/test_gpu_pair.py -x -q -s > $O/pytest_pair.txt 2>&1; tail -5 $O/pytest_pair.txt
rm -f $O/pair_time.txt
for i in 1 2 3 4 5 6; do HBS_PAIR_DEBUG=1 timeout 300 python scripts/r4/pair_time.py >> $O/pair_time.txt 2>> $O/pair_time.err; done
cat $O/pair_time.txt
