# round 5 experiment 1: 1 GiB calls -- K12 grid rounding, index-only tile height
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05exp1; mkdir -p $O
for g in 0 497 456 421 391; do
  if [ $g = 0 ]; then timeout 200 python scripts/scan_time.py --nals 104857 --reps 10 > $O/k12_g$g.txt 2>&1
  else HBS_GRID_BLOCKS=$g timeout 200 python scripts/scan_time.py --nals 104857 --reps 10 > $O/k12_g$g.txt 2>&1; fi
  echo "grid $g: $(tail -1 $O/k12_g$g.txt | cut -c1-400)"
done
for v in t64 t128; do
  HBS_LIB=build/variants/$v/libhbs.so timeout 200 python scripts/scan_time.py --nals 104857 --reps 10 > $O/idx_$v.txt 2>&1
  echo "$v 1GiB: $(tail -1 $O/idx_$v.txt | cut -c1-400)"
  HBS_LIB=build/variants/$v/libhbs.so timeout 200 python scripts/scan_time.py --reps 6 > $O/idx16_$v.txt 2>&1
  echo "$v 16GiB: $(tail -1 $O/idx16_$v.txt | cut -c1-400)"
done
for w in 8 16; do
  HBS5_WAVES_PER_CU=$w timeout 200 python scripts/scan_time.py --nals 104857 --reps 10 > $O/idx_w$w.txt 2>&1
  echo "waves $w 1GiB: $(tail -1 $O/idx_w$w.txt | cut -c1-400)"
  HBS5_WAVES_PER_CU=$w HBS_LIB=build/variants/t64/libhbs.so timeout 200 python scripts/scan_time.py --nals 104857 --reps 10 > $O/idx_t64_w$w.txt 2>&1
  echo "t64 waves $w 1GiB: $(tail -1 $O/idx_t64_w$w.txt | cut -c1-400)"
done
