cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_scan.py tests/test_gpu_index_parse.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -2
timeout 900 python scripts/nal_sweep.py --gib 2 --sizes 128,192,256,320,384,448,512,1024,10240 2>&1 | grep mean_nal | cut -c1-40,160-260
timeout 300 python scripts/scan_time.py 2>&1 | tail -1 | grep -o '"index_only".*'
timeout 300 python scripts/mixed_time.py 2>&1 | tail -1 | cut -c1-600
