# round 6: kernel stats and the PMC passes behind profiles/r06/ (run on the GPU box from the repo root; every rocprofv3 bounded)
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p; mkdir -p $O
B="python3 bench.py --steps 5 --warmup 1 --cpu-sample-nals 0"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bench -- $B --other-kernels 0 > $O/bench_line_under_rocprof.json 2> $O/rocprof_bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_other -- python3 bench.py --steps 2 --warmup 1 --cpu-sample-nals 0 --sweep 0 > $O/bench_line_other_under_rocprof.json 2> $O/rocprof_other.err
# BASELINE's own 1 GiB configs by themselves (bench.py's other_kernels.configs_1GiB without the host-side comparison)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_config_1gib -- python3 scripts/config_1gib.py > $O/config_1gib_under_rocprof.json 2> $O/rocprof_config_1gib.err
for d in stats_bench stats_other stats_config_1gib; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); cp $f $O/kernel_$d.csv; done
find $O/stats_bench $O/stats_other $O/stats_config_1gib -type f -delete
timeout 300 python3 scripts/config_1gib.py > $O/config_1gib.json 2> $O/config_1gib.err
P="python3 bench.py --steps 2 --warmup 1 --cpu-sample-nals 0"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch0 -- $P --other-kernels 0 > $O/pmc_fetch0_line.json 2> $O/pmc_fetch0.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write0 -- $P --other-kernels 0 > $O/pmc_write0_line.json 2> $O/pmc_write0.err
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $P --sweep 0 > $O/pmc_fetch_line.json 2> $O/pmc_fetch.err
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $P --sweep 0 > $O/pmc_write_line.json 2> $O/pmc_write.err
# the zero-heavy stress workload: since round 6 on the kernel's 24-row geometry (the density probe's choice)
Z="python3 bench.py --mode 1 --steps 2 --warmup 1 --cpu-sample-nals 0 --other-kernels 0"
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetchz -- $Z > $O/pmc_fetchz_line.json 2> $O/pmc_fetchz.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_writez -- $Z > $O/pmc_writez_line.json 2> $O/pmc_writez.err
ZALGO=$(python3 -c "import json,sys; print(json.loads(open('$O/pmc_fetchz_line.json').read().strip().splitlines()[-1])['roofline']['algorithmic_bytes'])")
python3 scripts/pmc_traffic.py $O/pmc_fetchz $O/pmc_writez k_scan_extract4_r24 $ZALGO $O/traffic_k_scan_extract4_r24_zero_heavy.json | tail -14
for p in fetch write; do f=$(find $O/pmc_${p}z -name "*counter_collection.csv" | head -1); (head -1 $f; grep "k_scan_extract4_r24" $f) > $O/pmc_${p}_k_scan_extract4_r24_zero_heavy.csv; done
find $O/pmc_fetchz $O/pmc_writez -type f -delete
python3 scripts/pmc_traffic.py $O/pmc_fetch0 $O/pmc_write0 k_scan_extract4 34403064115 $O/traffic_k_scan_extract4.json | tail -14
python3 scripts/pmc_traffic.py $O/pmc_fetch $O/pmc_write k_index5_stream 17231091218 $O/traffic_k_index5_stream.json near_max | tail -14
python3 scripts/pmc_traffic.py $O/pmc_fetch $O/pmc_write k3_tiles 34403064115 $O/traffic_k3_tiles.json near_max | tail -14
for p in fetch write; do
  f=$(find $O/pmc_${p}0 -name "*counter_collection.csv" | head -1)
  (head -1 $f; grep "k_scan_extract4(" $f) > $O/pmc_${p}_k_scan_extract4.csv
  for k in k_index5_stream k3_tiles; do
    f=$(find $O/pmc_$p -name "*counter_collection.csv" | head -1)
    (head -1 $f; grep "$k" $f) > $O/pmc_${p}_$k.csv
  done
done
find $O/pmc_fetch $O/pmc_write $O/pmc_fetch0 $O/pmc_write0 -type f -delete
timeout 900 python3 bench.py > $O/bench_line_default.json 2> $O/bench_default.err
timeout 300 python3 bench.py --mode 1 --steps 5 --warmup 1 --cpu-sample-nals 200000 --sweep 0 > $O/bench_line_zero_heavy.json 2> $O/bench_zero.err
ls -la $O
