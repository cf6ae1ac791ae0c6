cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in default wl default wl; do
  if [ $v = default ]; then L=""; else L="build/variants/$v/libhbs.so"; fi
  echo "== $v"
  HBS_LIB=$L timeout 300 python scripts/scan_time.py 2>&1 | tail -1 | cut -c60-330
  HBS_LIB=$L timeout 300 python scripts/scan_time.py --mode 1 --nals 1540000 2>&1 | tail -1 | cut -c60-330
  HBS_LIB=$L HBS_PIN=4 timeout 300 python scripts/r4/pin_time.py 512,1024,2048 2>&1 | grep mean
done
