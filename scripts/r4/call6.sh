cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
P=./build/ubench/placement
for i in 1 2; do PAIRS_N=20 timeout 300 $P pairs >> $O/placement_pairs.txt 2>&1; echo >> $O/placement_pairs.txt; done
cat $O/placement_pairs.txt
