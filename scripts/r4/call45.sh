cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
HBS_EMIT_NALS=1677000 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr_emit0 -- python3 scripts/emit_time.py 0 > $O/tr_emit0.txt 2>&1
f=$(find $O/tr_emit0 -name "*kernel_trace.csv" | head -1); python3 scripts/r4/trace_call.py $f k3t_check | head -7; find $O/tr_emit0 -type f -delete; tail -1 $O/tr_emit0.txt
