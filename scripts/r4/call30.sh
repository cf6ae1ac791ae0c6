cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_emit.py -x -q > $O/pytest_emit.txt 2>&1; tail -5 $O/pytest_emit.txt | cut -c1-400
timeout 900 python scripts/nal_sweep.py --gib 2 --sizes 64,192,224,256,320,384,448,512,1024 > $O/nal_sweep_tiny.txt 2>&1; tail -9 $O/nal_sweep_tiny.txt | cut -c1-120,250-420
