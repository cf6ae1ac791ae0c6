cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
for s in 512 1024 10240; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$s -- python3 scripts/nal_sweep.py --gib 2 --sizes $s > $O/st_$s.txt 2>&1
f=$(find $O/st_$s -name "*kernel_stats.csv" | head -1); grep -v "at::native" $f > $O/idx5_stats_$s.csv
find $O/st_$s -type f -delete
done
