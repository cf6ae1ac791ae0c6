cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_scan.py tests/test_gpu_emit.py tests/test_gpu_index_parse.py tests/test_gpu_shard.py -x -q > $O/pytest_call23.txt 2>&1; tail -3 $O/pytest_call23.txt | cut -c1-300
timeout 300 python scripts/emit_time.py > $O/emit_time_1GiB.txt 2>&1; tail -2 $O/emit_time_1GiB.txt
HBS_NO_SIDE_STREAMS=1 timeout 300 python scripts/emit_time.py 2>&1 | tail -2
timeout 300 python scripts/scan_time.py --nals 104857 > $O/scan_time_1GiB.txt 2>&1; tail -1 $O/scan_time_1GiB.txt | cut -c1-500
HBS_NO_SIDE_STREAMS=1 timeout 300 python scripts/scan_time.py --nals 104857 2>&1 | tail -1 | cut -c1-500
timeout 900 python scripts/nal_sweep.py --gib 2 --sizes 512,1024,10240 > $O/nal_sweep_mid.txt 2>&1; tail -3 $O/nal_sweep_mid.txt | cut -c1-420
HBS_EMIT_NALS=1677000 timeout 300 python scripts/emit_time.py > $O/emit_time_16GiB.txt 2>&1; tail -2 $O/emit_time_16GiB.txt
