cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_emit.py -x -q > $O/pytest_emit.txt 2>&1; tail -3 $O/pytest_emit.txt | cut -c1-300
HBS_EMIT_NALS=1677000 HBS_EMIT_MIXED=1 timeout 600 python scripts/emit_paths.py > $O/emit_paths_16GiB_mixed.txt 2>&1; tail -3 $O/emit_paths_16GiB_mixed.txt
HBS_EMIT_NALS=1677000 HBS_EMIT_MIXED=2 timeout 600 python scripts/emit_paths.py > $O/emit_paths_16GiB_mixed_zeros.txt 2>&1; tail -3 $O/emit_paths_16GiB_mixed_zeros.txt
