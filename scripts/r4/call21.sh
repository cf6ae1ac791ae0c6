cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_scan.py tests/test_gpu_index_parse.py -x -q > $O/pytest_scan.txt 2>&1; tail -3 $O/pytest_scan.txt | cut -c1-300
timeout 900 python scripts/nal_sweep.py --gib 2 --sizes 512,640,768,1024,2048,10240 > $O/nal_sweep_mid.txt 2>&1; tail -7 $O/nal_sweep_mid.txt | cut -c1-420
timeout 300 python scripts/scan_time.py --nals 104857 > $O/scan_time_1GiB.txt 2>&1; tail -1 $O/scan_time_1GiB.txt | cut -c1-500
timeout 300 python scripts/scan_time.py > $O/scan_time_16GiB.txt 2>&1; tail -1 $O/scan_time_16GiB.txt | cut -c1-500
