cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_emit.py -x -q > $O/pytest_emit.txt 2>&1; tail -3 $O/pytest_emit.txt | cut -c1-300
for m in 1 2; do
echo "mixed $m"; HBS_DZ_TIMING=1 HBS_ONLY_TILES=1 HBS_LIB=build/variants/dzt/libhbs.so HBS_EMIT_NALS=1677000 HBS_EMIT_MIXED=$m timeout 600 python scripts/emit_paths.py 2>&1 | tail -2
done
HBS_EMIT_NALS=1677000 HBS_EMIT_MIXED=1 timeout 600 python scripts/emit_paths.py 2>&1 | tail -3
HBS_EMIT_NALS=1677000 HBS_EMIT_MIXED=2 timeout 600 python scripts/emit_paths.py 2>&1 | tail -3
HBS_ONLY_TILES=1 HBS_EMIT_NALS=1677000 timeout 600 python scripts/emit_paths.py 2>&1 | tail -1
