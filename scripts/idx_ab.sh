#!/bin/bash
# index-only A/B over libraries, alternating (dev aid): bash scripts/idx_ab.sh out.txt lib1 lib2 ...   ("default" = the shipped library)
out="$1"; shift
for round in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = default ]; then env -u HBS_LIB python3 scripts/idx_ab.py >> "$out" 2>&1
    else HBS_LIB=build/variants/$lib/libhbs.so python3 scripts/idx_ab.py >> "$out" 2>&1; fi
  done
done
