#!/bin/bash
# round 3 dev call: parity + timing after a change to the scan kernels (exact element flags, medium tiles)
cd $GRAFT_REPO_ROOT
tag=${TAG:-x}
O=gpurun_out/r03; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_scan.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -4 > $O/exact_tests_$tag.txt
cat $O/exact_tests_$tag.txt
timeout 300 python tests/tools/soak_gpu.py ${SOAK_S:-120} 777 2>&1 | tail -3 | tee -a $O/exact_tests_$tag.txt
timeout 300 python bench.py --steps 5 --warmup 1 --cpu-sample-nals 0 --other-kernels 0 2>/dev/null | cut -c1-900 > $O/exact_bench_$tag.txt
timeout 300 python bench.py --mode 1 --steps 5 --warmup 1 --cpu-sample-nals 0 --other-kernels 0 2>/dev/null | cut -c1-900 >> $O/exact_bench_$tag.txt
timeout 600 python scripts/nal_sweep.py 2>&1 | grep -v amdgpu.ids > $O/exact_sweep_$tag.txt
python - <<PY
import json
for ln in open("$O/exact_bench_$tag.txt"):
    try:
        d = json.loads(ln if ln.rstrip().endswith("}") else ln[:ln.index(', "config"')] + "}")
    except Exception as e:
        print(ln[:300]); continue
    print(d["value"], d["ms_per_step"])
PY
grep -o '"roofline": {[^}]*}' $O/exact_bench_$tag.txt | cut -c1-200
cat $O/exact_sweep_$tag.txt | cut -c1-260
