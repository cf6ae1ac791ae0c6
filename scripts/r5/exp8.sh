cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05exp8; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_emit.py -x -q 2>&1 | tail -3
timeout 400 python tests/tools/soak_emit_small.py > $O/soak_emit_small.txt 2>&1; tail -2 $O/soak_emit_small.txt
timeout 600 python scripts/nal_sweep.py --gib 2 --sizes 64,128,192,224,256 > $O/sweep_small.txt 2>&1; grep mean_nal $O/sweep_small.txt | cut -c1-420
timeout 300 python scripts/k12_rounds.py > $O/k12_rounds.txt 2>&1; cat $O/k12_rounds.txt | grep tiles
