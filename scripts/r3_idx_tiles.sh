#!/bin/bash
# round 3 dev call: the index-only scan with other tile sizes (library variants), at 16 GiB, 1 GiB and on 2 GiB streams of small NALs
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03/idx_tiles.txt; mkdir -p gpurun_out/r03; : > $O
for v in $VARIANTS; do
  if [ $v = default ]; then unset HBS_LIB; else export HBS_LIB=$PWD/build/variants/$v/libhbs.so; fi
  echo "== $v" >> $O
  timeout 300 python scripts/index5_time.py 6 2>&1 | grep -v amdgpu.ids | cut -c1-200 >> $O
  for m in 1024 10240; do HBS5_NAL_MEAN=$m timeout 300 python scripts/index5_time.py 6 2>&1 | grep -v amdgpu.ids | cut -c1-200 >> $O; done
  timeout 300 python scripts/scan_time.py --reps 6 --nals 104857 2>&1 | grep -v amdgpu.ids | cut -c150-400 >> $O
done
cat $O
