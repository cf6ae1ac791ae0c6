cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05p2; mkdir -p $O
B="python3 bench.py --steps 5 --warmup 1 --cpu-sample-nals 0"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bench -- $B --other-kernels 0 > $O/bench_line_under_rocprof.json 2> $O/rocprof_bench.err
f=$(find $O/stats_bench -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_stats_bench.csv; find $O/stats_bench -type f -delete
grep "k_scan_extract4" $O/kernel_stats_bench.csv | cut -c1-40,380-480; tail -1 $O/bench_line_under_rocprof.json | cut -c1-300
timeout 900 python3 bench.py > $O/bench_line_default.json 2> $O/bench_default.err; echo "bench rc $?"; tail -1 $O/bench_line_default.json | cut -c1-400
# K12: is the fixed cost of a call a whole-round effect?  5 120 / 5 463 / 5 632 tiles = 10 / 10.67 / 11 rounds of 512 workgroups
for n in 98280 104857 108100; do timeout 200 python scripts/scan_time.py --nals $n --reps 12 > $O/k12_$n.txt 2>&1; echo "$n: $(tail -1 $O/k12_$n.txt | cut -c20-260)"; done
