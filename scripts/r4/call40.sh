cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_emit.py -x -q 2>&1 | tail -3
HBS_EMIT_NALS=1677000 HBS_EMIT_MIXED=1 timeout 600 python scripts/emit_paths.py 2>&1 | tail -1
HBS_EMIT_NALS=1677000 HBS_EMIT_MIXED=2 HBS_ONLY_TILES=1 timeout 600 python scripts/emit_paths.py 2>&1 | tail -1
HBS_ONLY_TILES=1 HBS_EMIT_NALS=1677000 timeout 600 python scripts/emit_paths.py 2>&1 | tail -1
HBS_ONLY_TILES=1 HBS_EMIT_NALS=1540000 timeout 600 python scripts/emit_paths.py 1 2>&1 | tail -1
timeout 300 python scripts/emit_time.py 2>&1 | tail -2
