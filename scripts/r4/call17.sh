cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 900 python scripts/nal_sweep.py --sizes 64,128,256,320,384,512 > $O/nal_sweep_tiny.txt 2>&1; cat $O/nal_sweep_tiny.txt | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['mean_nal_bytes'], d['nals'], 'extract', d['extract']['traffic_frac'], 'index', d['index_only']['read_frac'], 'emit ms', d['emit']['ms'], d['emit']['traffic_frac'])
    else: print(l.strip()[:300])
"
timeout 1200 python -m pytest tests/test_gpu_emit.py -x -q 2>&1 | tail -2
