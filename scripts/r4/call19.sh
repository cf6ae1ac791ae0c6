cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
HBS_EMIT_NALS=1677000 timeout 600 python scripts/emit_paths.py > $O/emit_paths_16GiB.txt 2>&1; tail -4 $O/emit_paths_16GiB.txt
HBS_EMIT_NALS=1677000 HBS_EMIT_MIXED=1 timeout 600 python scripts/emit_paths.py > $O/emit_paths_16GiB_mixed.txt 2>&1; tail -4 $O/emit_paths_16GiB_mixed.txt
HBS_EMIT_NALS=1677000 HBS_EMIT_MIXED=2 timeout 600 python scripts/emit_paths.py > $O/emit_paths_16GiB_mixed_zeros.txt 2>&1; tail -4 $O/emit_paths_16GiB_mixed_zeros.txt
HBS_EMIT_NALS=1677000 timeout 600 python scripts/emit_paths.py 1 > $O/emit_paths_16GiB_zero_heavy.txt 2>&1; tail -4 $O/emit_paths_16GiB_zero_heavy.txt
timeout 900 python tests/tools/soak_gpu.py 300 4 > $O/soak_r04.txt 2>&1; tail -3 $O/soak_r04.txt
