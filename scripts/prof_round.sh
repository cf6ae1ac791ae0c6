set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bench -- python3 bench.py --steps 5 --warmup 1 --cpu-sample-nals 0 --other-kernels 0 > $O/bench_line_under_rocprof.json 2> $O/rocprof_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_other -- python3 bench.py --steps 2 --warmup 1 --cpu-sample-nals 0 > $O/bench_line_other_under_rocprof.json 2> $O/rocprof_other.err
find $O/stats_bench $O/stats_other -name "*kernel_stats.csv" | head
for d in stats_bench stats_other; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_$d.csv; done
find $O/stats_bench $O/stats_other -type f ! -name "*kernel_stats.csv" -delete
python3 bench.py > $O/bench_line_default.json 2> $O/bench_default.err
python3 bench.py --mode 1 --steps 5 --warmup 1 --cpu-sample-nals 200000 > $O/bench_line_zero_heavy.json 2> $O/bench_zero.err
