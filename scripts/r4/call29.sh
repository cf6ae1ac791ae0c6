cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_emit.py -x -q -k "tiny" 2>&1 | tail -2
timeout 900 python scripts/nal_sweep.py --gib 2 --sizes 64,128,256,384 > $O/nal_sweep_tiny.txt 2>&1; tail -4 $O/nal_sweep_tiny.txt | cut -c1-420
