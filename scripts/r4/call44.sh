cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e; mkdir -p $O gpurun_out/r04p
bash scripts/prof_r04.sh > gpurun_out/r04p/prof_r04.log 2>&1; tail -2 gpurun_out/r04p/prof_r04.log
HBS_EMIT_NALS=1677000 timeout 600 python scripts/emit_paths.py > $O/emit_paths_16GiB.txt 2>&1
HBS_EMIT_NALS=1677000 HBS_EMIT_MIXED=1 timeout 600 python scripts/emit_paths.py > $O/emit_paths_16GiB_mixed.txt 2>&1
HBS_EMIT_NALS=1677000 HBS_EMIT_MIXED=2 timeout 900 python scripts/emit_paths.py > $O/emit_paths_16GiB_mixed_zeros.txt 2>&1
HBS_EMIT_NALS=1540000 timeout 600 python scripts/emit_paths.py 1 > $O/emit_paths_16GiB_zero_heavy.txt 2>&1
timeout 300 python scripts/emit_time.py > $O/emit_time_1GiB.txt 2>&1
HBS_EMIT_NALS=1677000 timeout 300 python scripts/emit_time.py > $O/emit_time_16GiB.txt 2>&1
timeout 1200 python scripts/nal_sweep.py --gib 2 --sizes 64,128,192,224,256,320,384,448,512,640,768,1024,2048,4096,10240,65536,524288 > $O/nal_sweep.txt 2>&1
rm -f $O/pair_time.txt; for i in 1 2 3 4 5 6; do timeout 300 python scripts/r4/pair_time.py >> $O/pair_time.txt 2>> $O/pair_time.err; done
timeout 400 python tests/tools/soak_emit_small.py > $O/soak_emit_small.txt 2>&1
timeout 500 python tests/tools/soak_gpu.py 240 37 > $O/soak_r04.txt 2>&1
HBS_EMIT_NALS=104858 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr_emit -- python3 scripts/emit_time.py > $O/tr_emit.txt 2>&1
f=$(find $O/tr_emit -name "*kernel_trace.csv" | head -1); python3 scripts/r4/trace_call.py $f k3t_check > $O/trace_emit_1GiB.txt; find $O/tr_emit -type f -delete; rm -f $O/tr_emit.txt
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r04p/pytest_gpu_final.txt 2>&1; tail -2 gpurun_out/r04p/pytest_gpu_final.txt
for f in emit_paths_16GiB emit_paths_16GiB_mixed emit_paths_16GiB_mixed_zeros emit_paths_16GiB_zero_heavy emit_time_1GiB emit_time_16GiB pair_time soak_emit_small soak_r04 trace_emit_1GiB; do echo "== $f"; tail -3 $O/$f.txt | cut -c1-260; done
