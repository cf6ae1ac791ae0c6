cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr_emit -- python3 scripts/emit_time.py > $O/tr_emit.txt 2>&1
f=$(find $O/tr_emit -name "*kernel_trace.csv" | head -1); python3 scripts/r4/trace_call.py $f k3t_check > $O/trace_emit_1GiB.txt; find $O/tr_emit -type f -delete
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr_scan -- python3 scripts/scan_time.py --nals 104857 > $O/tr_scan.txt 2>&1
f=$(find $O/tr_scan -name "*kernel_trace.csv" | head -1); python3 scripts/r4/trace_call.py $f k_scan_prologue > $O/trace_scan_1GiB.txt; find $O/tr_scan -type f -delete
cat $O/trace_emit_1GiB.txt $O/trace_scan_1GiB.txt
