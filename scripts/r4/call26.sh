cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
for m in 1 2; do
echo "mixed $m"; HBS_DZ_TIMING=1 HBS_ONLY_TILES=1 HBS_LIB=build/variants/dzt/libhbs.so HBS_EMIT_NALS=1677000 HBS_EMIT_MIXED=$m timeout 600 python scripts/emit_paths.py 2>&1 | tail -2
done
