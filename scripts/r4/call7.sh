cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
P=./build/ubench/placement
for i in 1 2 3; do timeout 300 $P probekernels >> $O/placement_probekernels.txt 2>&1; echo >> $O/placement_probekernels.txt; done
cat $O/placement_probekernels.txt
