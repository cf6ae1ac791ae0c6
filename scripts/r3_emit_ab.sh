#!/bin/bash
# round 3 dev call: K3 A/B (variants in $VARIANTS) at 16 GiB and 1 GiB + emit tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
tag=${TAG:-e}
for v in $VARIANTS; do
  if [ $v = default ]; then unset HBS_LIB; else export HBS_LIB=$PWD/build/variants/$v/libhbs.so; fi
  echo "== $v 16 GiB" >> gpurun_out/r03/emit_ab_$tag.txt
  HBS_EMIT_NALS=1677000 timeout 300 python scripts/emit_paths.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r03/emit_ab_$tag.txt
  echo "== $v 1 GiB" >> gpurun_out/r03/emit_ab_$tag.txt
  timeout 300 python scripts/emit_paths.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r03/emit_ab_$tag.txt
done
unset HBS_LIB
cat gpurun_out/r03/emit_ab_$tag.txt
python -m pytest tests/test_gpu_emit.py -x -q 2>&1 | tail -3
