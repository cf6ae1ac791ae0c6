cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
HBS_PAIR_DEBUG=1 timeout 600 python -m pytest tests/test_gpu_pair.py -x -q -s > $O/pytest_pair.txt 2>&1; tail -5 $O/pytest_pair.txt; grep "hbs_pair_alloc" $O/pytest_pair.txt | grep -v candidate | head -20
timeout 900 python -m pytest tests/test_gpu_parse.py tests/test_gpu_legacy.py tests/test_gpu_index_parse.py -x -q > $O/pytest_parse.txt 2>&1; tail -5 $O/pytest_parse.txt
timeout 600 python scripts/r4/fix_time.py > $O/fix_time.txt 2>&1; cat $O/fix_time.txt
timeout 900 python tests/tools/fuzz_gpu_parse.py 1000 300 > $O/fuzz_gpu_parse.txt 2>&1; tail -3 $O/fuzz_gpu_parse.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu_a.txt 2>&1; tail -5 $O/pytest_gpu_a.txt
HBS_PAIR_DEBUG=1 timeout 900 python bench.py > $O/bench_a.json 2> $O/bench_a.err; tail -c 1500 $O/bench_a.json; grep -v candidate $O/bench_a.err | tail -12
