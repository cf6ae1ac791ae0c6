# round 6, after the emit faults: every fuzzer and soak again with seeds nobody has run, longer than the evidence script does
# (run on the GPU box from the repo root; everything bounded)
O=gpurun_out/r06f; mkdir -p $O
timeout 900 python tests/tools/fuzz_gpu_parse.py 20000 500 > $O/fuzz_gpu_parse_20000.txt 2>&1
timeout 900 python tests/tools/fuzz_gpu_compact.py 21000 400 > $O/fuzz_gpu_compact_21000.txt 2>&1
timeout 900 python tests/tools/fuzz_gpu_index_parse.py 22000 400 > $O/fuzz_gpu_index_parse_22000.txt 2>&1
timeout 900 python tests/tools/fuzz_gpu_index_parts.py 60 23 > $O/fuzz_gpu_index_parts_23.txt 2>&1
timeout 900 python tests/tools/fuzz_gpu_legacy.py 24000 800 > $O/fuzz_gpu_legacy_24000.txt 2>&1
timeout 800 python tests/tools/soak_gpu.py 600 123 > $O/soak_gpu_seed123.txt 2>&1
timeout 500 python tests/tools/soak_emit_small.py 400 321 > $O/soak_emit_small_seed321.txt 2>&1
timeout 400 python tests/tools/canaries.py 300 77 > $O/canaries_seed77.txt 2>&1
for f in $O/*.txt; do echo "== $f"; tail -n 2 $f | cut -c1-300; done
