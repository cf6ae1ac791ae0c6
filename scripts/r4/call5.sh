cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
P=./build/ubench/placement
for i in 1 2 3; do timeout 200 $P delta >> $O/placement_delta.txt 2>&1; echo >> $O/placement_delta.txt; done
for i in 1 2 3; do timeout 200 $P chunks >> $O/placement_chunks.txt 2>&1; echo >> $O/placement_chunks.txt; done
cat $O/placement_delta.txt $O/placement_chunks.txt
