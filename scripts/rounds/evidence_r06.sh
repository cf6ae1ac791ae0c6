# round 6: the timing / soak / sweep evidence copied into profiles/r06/ (run on the GPU box from the repo root; everything bounded)
export HBS_PLAIN_ALLOC=1   # round 6: outputs from the plain allocator, as bench.py (placement is opt-in; scripts/pair_time.py measures both)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06e; mkdir -p $O
timeout 1200 python scripts/nal_sweep.py --gib 2 --sizes 64,128,192,224,256,320,384,448,512,640,768,1024,2048,4096,10240,65536,524288 > $O/nal_sweep.txt 2>&1
HBS_EMIT_NALS=1677000 timeout 600 python scripts/emit_paths.py > $O/emit_paths_16GiB.txt 2>&1
HBS_EMIT_NALS=1677000 HBS_EMIT_MIXED=1 timeout 600 python scripts/emit_paths.py > $O/emit_paths_16GiB_mixed.txt 2>&1
HBS_EMIT_NALS=1677000 HBS_EMIT_MIXED=2 timeout 900 python scripts/emit_paths.py > $O/emit_paths_16GiB_mixed_zeros.txt 2>&1
HBS_EMIT_NALS=1540000 timeout 600 python scripts/emit_paths.py 1 > $O/emit_paths_16GiB_zero_heavy.txt 2>&1
timeout 300 python scripts/scan_time.py --nals 104857 > $O/scan_time_1GiB.txt 2>&1
timeout 300 python scripts/scan_time.py > $O/scan_time_16GiB.txt 2>&1
timeout 300 python scripts/emit_time.py > $O/emit_time_1GiB.txt 2>&1
HBS_EMIT_NALS=1677000 timeout 300 python scripts/emit_time.py > $O/emit_time_16GiB.txt 2>&1
timeout 300 python scripts/mixed_time.py > $O/mixed_time.txt 2>&1
# the mixed stream at the bench's size: dense tiles counted ahead (default), counted in place, and round 4's library (a worktree of
# 823d19e built as build/variants/r04) on the same box; at 1 GiB (count-ahead is off below 4 GiB); a timeline per tile (diagnostic build)
rm -f $O/mixed_time_16GiB.txt
for m in 1 0; do echo "HBS_COUNT_AHEAD=$m" >> $O/mixed_time_16GiB.txt; HBS_COUNT_AHEAD=$m timeout 300 python scripts/mixed_time.py --nals 1677000 >> $O/mixed_time_16GiB.txt 2>&1; done
if [ -f build/variants/r04/libhbs.so ]; then echo "round 4's library" >> $O/mixed_time_16GiB.txt; HBS_LIB=build/variants/r04/libhbs.so timeout 300 python scripts/mixed_time.py --nals 1677000 >> $O/mixed_time_16GiB.txt 2>&1; fi
timeout 300 python scripts/mixed_time.py --nals 105000 > $O/mixed_time_1GiB.txt 2>&1
if [ -f build/diag/libhbs_diag.so ]; then HBS_LIB=build/diag/libhbs_diag.so timeout 300 python scripts/mixed_timeline.py > $O/mixed_timeline_4GiB.txt 2>&1; fi
for s in 512 1024 10240; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$s -- python3 scripts/nal_sweep.py --gib 2 --sizes $s > $O/st_$s.txt 2>&1
  f=$(find $O/st_$s -name "*kernel_stats.csv" | head -1); grep -v "at::native" $f > $O/idx5_stats_$s.csv; find $O/st_$s -type f -delete; rm -f $O/st_$s.txt
done
rm -f $O/pair_time.txt; for i in 1 2 3 4 5 6; do timeout 300 python scripts/pair_time.py >> $O/pair_time.txt 2>> $O/pair_time.err; done
timeout 300 python tests/tools/cli_time.py > $O/cli_time.txt 2>&1
timeout 600 python scripts/fix_time.py > $O/fix_time.txt 2>&1
timeout 400 python tests/tools/soak_gpu.py 200 11 > $O/soak_r06.txt 2>&1
timeout 500 python tests/tools/soak_emit_small.py 400 5 > $O/soak_emit_small.txt 2>&1
# the soak that found the two emit faults of this round (seed 66: iterations 47568 and 50920), again on the fixed library; buffers that end with their allocation
timeout 1500 python tests/tools/soak_gpu.py 1200 66 > $O/soak_long_seed66.txt 2>&1
timeout 1200 python tests/tools/edge_faults.py > $O/edge_faults.txt 2>&1
timeout 600 python tests/tools/fuzz_gpu_parse.py 1000 200 > $O/fuzz_gpu_parse.txt 2>&1
timeout 600 python tests/tools/fuzz_gpu_legacy.py > $O/fuzz_gpu_legacy.txt 2>&1
timeout 300 python scripts/config3_time.py > $O/config3_time.txt 2>&1
timeout 300 python scripts/config_1gib.py > $O/config_1gib.txt 2>&1
timeout 300 python scripts/scan_time.py --nals 209715 > $O/scan_time_2GiB.txt 2>&1
timeout 600 python scripts/emit_sweep.py --sizes 48,64,96,128,160,192,224,256,384 > $O/emit_sweep.txt 2>&1
timeout 600 python scripts/sweep_forced.py --sizes 64,128,192,256,384,512,768,1024 --kernels 0,2,4,6 > $O/sweep_forced.txt 2>&1
timeout 300 python scripts/experiments/index_stream_placement.py > $O/index_stream_placement.txt 2>&1
for f in mixed_time_16GiB mixed_time_1GiB mixed_timeline_4GiB nal_sweep emit_paths_16GiB emit_paths_16GiB_mixed emit_paths_16GiB_mixed_zeros emit_paths_16GiB_zero_heavy scan_time_1GiB scan_time_16GiB emit_time_1GiB emit_time_16GiB mixed_time pair_time cli_time fix_time soak_r06 soak_emit_small soak_long_seed66 edge_faults config3_time config_1gib scan_time_2GiB emit_sweep sweep_forced index_stream_placement fuzz_gpu_parse fuzz_gpu_legacy; do echo "== $f"; tail -3 $O/$f.txt | cut -c1-300; done
