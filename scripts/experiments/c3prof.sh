cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/c3prof
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/c3prof -o c3 -- python3 scripts/config3_time.py > gpurun_out/c3prof/out.txt 2>&1
tail -1 gpurun_out/c3prof/out.txt
