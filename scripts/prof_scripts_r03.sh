# round 3: the dev-aid timings quoted in DESIGN.md, into gpurun_out/r03d (run on the GPU box from the repo root)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d; mkdir -p $O
timeout 300 python3 scripts/scan_time.py --reps 8 2>/dev/null > $O/scan_time.txt
timeout 300 python3 scripts/scan_time.py --reps 6 --mode 1 2>/dev/null > $O/scan_time_zero_heavy.txt
timeout 300 python3 scripts/scan_time.py --reps 8 --nals 104857 2>/dev/null > $O/scan_time_1GiB.txt
timeout 300 python3 scripts/mixed_time.py 2>/dev/null > $O/mixed_time.txt
timeout 300 python3 scripts/emit_time.py 2>/dev/null | tail -3 > $O/emit_time_1GiB.txt
HBS_EMIT_NALS=1677000 timeout 300 python3 scripts/emit_time.py 2>/dev/null | tail -3 > $O/emit_time_16GiB.txt
timeout 300 python3 scripts/emit_density.py 2>/dev/null > $O/emit_density.txt
timeout 300 python3 scripts/density_sweep.py 2>/dev/null > $O/density_sweep.txt
timeout 600 python3 scripts/nal_sweep.py --sizes 64,384,512,768,1024,1536,2048,4096,10240,65536,524288 2>/dev/null > $O/nal_sweep.txt
timeout 300 python3 scripts/parse_time.py 2>/dev/null > $O/parse_time.txt
timeout 200 python3 tests/tools/phase_timing4.py 0 2>/dev/null > $O/phase_timing4_final.txt
HBS4_NAL_MEAN=1024 timeout 200 python3 tests/tools/phase_timing4.py 0 2>/dev/null > $O/phase_timing4_final_1KiB_nals.txt
timeout 200 python3 tests/tools/phase_timing4.py 1 2>/dev/null > $O/phase_timing4_final_zero_heavy.txt
HBS_EMIT_NALS=1677000 timeout 300 python3 scripts/emit_paths.py 2>/dev/null > $O/emit_paths_16GiB.txt
HBS_EMIT_NALS=1677000 HBS_EMIT_MIXED=1 timeout 300 python3 scripts/emit_paths.py 2>/dev/null > $O/emit_paths_16GiB_mixed.txt
./build/ubench/perm_flag_check > $O/perm_flag_check.txt 2>&1
tail -n 4 $O/scan_time.txt $O/scan_time_zero_heavy.txt $O/scan_time_1GiB.txt $O/mixed_time.txt $O/emit_time_1GiB.txt $O/emit_time_16GiB.txt $O/density_sweep.txt | cut -c1-400
