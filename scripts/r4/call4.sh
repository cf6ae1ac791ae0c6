cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
P=./build/ubench/placement
for i in 1 2 3 4 5; do timeout 200 $P matrix >> $O/placement_matrix.txt 2>&1; echo >> $O/placement_matrix.txt; done
cat $O/placement_matrix.txt
