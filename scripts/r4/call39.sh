cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e; mkdir -p $O gpurun_out/r04p
bash scripts/prof_r04.sh > gpurun_out/r04p/prof_r04.log 2>&1; tail -2 gpurun_out/r04p/prof_r04.log
timeout 1200 python scripts/nal_sweep.py --gib 2 --sizes 64,128,192,224,256,320,384,448,512,640,768,1024,2048,4096,10240,65536,524288 > $O/nal_sweep.txt 2>&1
timeout 300 python scripts/scan_time.py --nals 104857 > $O/scan_time_1GiB.txt 2>&1
timeout 300 python scripts/scan_time.py > $O/scan_time_16GiB.txt 2>&1
timeout 300 python scripts/mixed_time.py > $O/mixed_time.txt 2>&1
for s in 512 1024 10240; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$s -- python3 scripts/nal_sweep.py --gib 2 --sizes $s > $O/st_$s.txt 2>&1
  f=$(find $O/st_$s -name "*kernel_stats.csv" | head -1); grep -v "at::native" $f > $O/idx5_stats_$s.csv; find $O/st_$s -type f -delete; rm -f $O/st_$s.txt
done
timeout 500 python tests/tools/soak_gpu.py 240 23 > $O/soak_r04.txt 2>&1; tail -1 $O/soak_r04.txt
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r04p/pytest_gpu_final.txt 2>&1; tail -2 gpurun_out/r04p/pytest_gpu_final.txt
python3 bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err; cut -c1-200 gpurun_out/bench_final.json
