# the dev-aid timings quoted in DESIGN.md, into gpurun_out/r02 (run on the GPU box from the repo root)
O=gpurun_out/r02; mkdir -p $O
python3 scripts/scan_time.py > $O/scan_time.txt 2>/dev/null
python3 scripts/mixed_time.py > $O/mixed_time.txt 2>/dev/null
python3 scripts/emit_time.py 2>/dev/null | tail -3 > $O/emit_time_1GiB.txt
HBS_EMIT_NALS=1677000 python3 scripts/emit_time.py 2>/dev/null | tail -3 > $O/emit_time_16GiB.txt
python3 scripts/emit_real.py 2 2>/dev/null | tail -3 > $O/emit_real.txt
python3 scripts/emit_density.py 2>/dev/null > $O/emit_density.txt
python3 scripts/parse_time.py 2>/dev/null > $O/parse_time.txt
python3 scripts/legacy_time.py 2>/dev/null > $O/legacy_time.txt
python3 scripts/ingest_time.py 2>/dev/null > $O/ingest_time.txt
python3 tests/tools/phase_timing4.py 2>/dev/null > $O/phase_timing4.txt
HBS_EMIT_NALS=1677000 python3 scripts/emit_phase_cycles.py 2>/dev/null > $O/emit_phase_cycles.txt
HBS_EMIT_NALS=1677000 python3 scripts/emit_paths.py 2>/dev/null > $O/emit_paths_16GiB.txt
python3 scripts/emit_paths.py 2>/dev/null > $O/emit_paths_1GiB.txt
HBS_EMIT_NALS=1677000 python3 scripts/emit_phase_tiles.py 2>/dev/null > $O/emit_phase_tiles.txt
tail -n 3 $O/scan_time.txt $O/mixed_time.txt $O/phase_timing4.txt $O/emit_paths_16GiB.txt
