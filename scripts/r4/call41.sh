cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
for m in 0 1; do
N=1677000; [ $m = 1 ] && N=1540000
HBS_EMIT_NALS=$N timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr_emit$m -- python3 scripts/emit_time.py $m > $O/tr_emit$m.txt 2>&1
f=$(find $O/tr_emit$m -name "*kernel_trace.csv" | head -1); python3 scripts/r4/trace_call.py $f k3t_check | head -14; find $O/tr_emit$m -type f -delete
done
