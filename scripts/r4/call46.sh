cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_emit.py -x -q 2>&1 | tail -2
for i in 1 2 3; do HBS_EMIT_NALS=1677000 timeout 300 python scripts/emit_time.py 2>&1 | tail -1; done
HBS_EMIT_NALS=1677000 HBS_EMIT_MIXED=1 HBS_ONLY_TILES=1 timeout 600 python scripts/emit_paths.py 2>&1 | tail -1
python scripts/r4/pair_time.py 2>/dev/null | tail -2 | cut -c1-120
