cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_emit.py -x -q 2>&1 | tail -2
timeout 300 python3 scripts/r4/emit_phase_mean.py 1024 2>&1 | grep -v amdgpu.ids
timeout 900 python scripts/nal_sweep.py --gib 2 --sizes 256,512,1024,2048,10240 2>&1 | grep mean_nal | cut -c1-40,230-330
for i in 1 2; do HBS_EMIT_NALS=1677000 timeout 300 python scripts/emit_time.py 2>&1 | tail -1; done
HBS_EMIT_NALS=1677000 HBS_EMIT_MIXED=1 HBS_ONLY_TILES=1 timeout 600 python scripts/emit_paths.py 2>&1 | tail -1
HBS_ONLY_TILES=1 HBS_EMIT_NALS=1540000 timeout 600 python scripts/emit_paths.py 1 2>&1 | tail -1
