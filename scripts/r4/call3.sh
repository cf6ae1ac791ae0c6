cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
P=./build/ubench/placement
for i in 1 2 3 4 5 6 7 8 9 10; do timeout 120 $P probe >> $O/placement_probe.txt 2>&1; done
cat $O/placement_probe.txt
