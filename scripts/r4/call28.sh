cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in dzt4 dzt5; do
echo "variant $v"; HBS_NO_CHECK=1 HBS_DZ_TIMING=1 HBS_ONLY_TILES=1 HBS_LIB=build/variants/$v/libhbs.so HBS_EMIT_NALS=1677000 HBS_EMIT_MIXED=1 timeout 600 python scripts/emit_paths.py 2>&1 | tail -2
done
