#!/bin/bash
# round 3 dev call: A/B of the medium-tile threshold (library variants in $VARIANTS; "default" = the shipped library)
cd $GRAFT_REPO_ROOT
tag=${TAG:-x}
O=gpurun_out/r03/med_ab_$tag.txt; mkdir -p gpurun_out/r03; : > $O
for v in $VARIANTS; do
  if [ $v = default ]; then unset HBS_LIB; else export HBS_LIB=$PWD/build/variants/$v/libhbs.so; fi
  echo "== $v" >> $O
  timeout 300 python scripts/scan_time.py --reps 6 2>&1 | grep -v amdgpu.ids | cut -c1-330 >> $O
  timeout 300 python scripts/scan_time.py --reps 4 --mode 1 2>&1 | grep -v amdgpu.ids | cut -c1-330 >> $O
  timeout 300 python scripts/nal_sweep.py --sizes ${SIZES:-1024,1536,2048,3072} 2>&1 | grep -v amdgpu.ids | cut -c1-200 >> $O
done
unset HBS_LIB
if [ -n "$PHASES" ]; then
  for m in 1024 2048; do HBS_DIAG_LIB=build/variants/diag99/libhbs.so HBS4_NAL_MEAN=$m timeout 200 python tests/tools/phase_timing4.py 0 2>&1 | grep -v amdgpu.ids >> $O; done
fi
cat $O
