# the two PMC passes behind profiles/rNN/traffic_*.json (separate passes, no trace domains); run on the GPU box from the repo root
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02; mkdir -p $O
# the fused kernel: the timed steps only (every launch is on the bench stream)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch0 -- python3 bench.py --steps 2 --warmup 1 --cpu-sample-nals 0 --other-kernels 0 > $O/pmc_fetch0_line.json 2> $O/pmc_fetch0.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write0 -- python3 bench.py --steps 2 --warmup 1 --cpu-sample-nals 0 --other-kernels 0 > $O/pmc_write0_line.json 2> $O/pmc_write0.err
# the other kernels: bench.py's other_kernels leg (index-only kernel x4 and K3 x5 on the 16 GiB stream / arena; its config-3 leg runs K3 once on 2 GiB: dropped by near_max)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --cpu-sample-nals 0 > $O/pmc_fetch_line.json 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 2 --warmup 1 --cpu-sample-nals 0 > $O/pmc_write_line.json 2> $O/pmc_write.err
python3 scripts/pmc_traffic.py $O/pmc_fetch0 $O/pmc_write0 k_scan_extract4 34403064115 $O/traffic_k_scan_extract4.json | tail -12
python3 scripts/pmc_traffic.py $O/pmc_fetch $O/pmc_write k_scan_index5 17231091218 $O/traffic_k_scan_index5.json | tail -12
python3 scripts/pmc_traffic.py $O/pmc_fetch $O/pmc_write k3_tiles 34403064115 $O/traffic_k3_tiles.json near_max | tail -12
for p in fetch write; do
  f=$(find $O/pmc_${p}0 -name "*counter_collection.csv" | head -1)
  (head -1 $f; grep "k_scan_extract4" $f) > $O/pmc_${p}_k_scan_extract4.csv
  for k in k_scan_index5 k3_tiles; do
    f=$(find $O/pmc_$p -name "*counter_collection.csv" | head -1)
    (head -1 $f; grep "$k" $f) > $O/pmc_${p}_$k.csv
  done
done
find $O/pmc_fetch $O/pmc_write $O/pmc_fetch0 $O/pmc_write0 -type f -delete
