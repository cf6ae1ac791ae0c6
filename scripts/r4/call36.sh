cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in default t128 t64; do
  if [ $v = default ]; then L=""; else L="build/variants/$v/libhbs.so"; fi
  echo "== $v"
  HBS_LIB=$L timeout 300 python scripts/scan_time.py --nals 104857 2>&1 | tail -1 | grep -o '"index_only".*'
  HBS_LIB=$L timeout 300 python scripts/scan_time.py --nals 209715 2>&1 | tail -1 | grep -o '"index_only".*'
  HBS_LIB=$L timeout 300 python scripts/scan_time.py 2>&1 | tail -1 | grep -o '"index_only".*'
done
