#!/bin/bash
# round 3, GPU call 1: ceiling3 ubench, progressive-fetch A/B, phase timing at 16 GiB
mkdir -p gpurun_out/r03
cd $GRAFT_REPO_ROOT
timeout 300 ./build/ubench/ceiling3 16 > gpurun_out/r03/ceiling3.txt 2>&1
for v in base default base default; do
  if [ $v = default ]; then unset HBS_LIB; else export HBS_LIB=$PWD/build/variants/$v/libhbs.so; fi
  timeout 300 python scripts/scan_time.py --reps 8 >> gpurun_out/r03/scan_ab.txt 2>&1
done
unset HBS_LIB
HBS4_REPEAT=1024 timeout 300 python tests/tools/phase_timing4.py 0 1 > gpurun_out/r03/phase_prog.txt 2>&1
HBS4_REPEAT=1024 HBS4_FAKE_LB=1 timeout 300 python tests/tools/phase_timing4.py 0 1 > gpurun_out/r03/phase_prog_fakelb.txt 2>&1
tail -3 gpurun_out/r03/ceiling3.txt; cat gpurun_out/r03/scan_ab.txt | cut -c1-400
