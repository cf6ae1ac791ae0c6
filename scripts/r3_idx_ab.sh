#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
tag=${TAG:-i}
for v in $VARIANTS; do
  for w in $WAVES; do
    if [ $v = default ]; then unset HBS_LIB; else export HBS_LIB=$PWD/build/variants/$v/libhbs.so; fi
    HBS5_WAVES_PER_CU=$w timeout 200 python scripts/index5_time.py 5 2>&1 | grep -v amdgpu.ids >> gpurun_out/r03/idx_ab_$tag.txt
  done
done
cat gpurun_out/r03/idx_ab_$tag.txt
