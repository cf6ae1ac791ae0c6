#!/bin/bash
# round 3 dev call: ceiling3 ubench + A/B of library variants (names in $VARIANTS, "default" = the shipped library)
cd $GRAFT_REPO_ROOT
tag=${TAG:-x}
mkdir -p gpurun_out/r03
if [ -z "$SKIP_UBENCH" ]; then timeout 600 ./build/ubench/ceiling3 16 > gpurun_out/r03/ceiling3_$tag.txt 2>&1; fi
for v in $VARIANTS; do
  if [ $v = default ]; then unset HBS_LIB; else export HBS_LIB=$PWD/build/variants/$v/libhbs.so; fi
  timeout 300 python scripts/scan_time.py --reps 8 2>&1 | grep -v amdgpu.ids >> gpurun_out/r03/scan_ab_$tag.txt
done
unset HBS_LIB
cut -c1-420 gpurun_out/r03/scan_ab_$tag.txt
