# round 4, GPU call 1: the tests touched so far, the counter list, and the placement probes (no torch in them)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
rocprofv3 --list-avail > $O/list_avail.txt 2>&1
P=./build/ubench/placement
for i in 1 2 3 4 5 6; do timeout 120 $P sep >> $O/placement_sep.txt 2>&1; done
for i in 1 2 3 4 5 6; do timeout 120 $P slab >> $O/placement_slab.txt 2>&1; done
timeout 200 $P realloc > $O/placement_realloc.txt 2>&1
timeout 300 $P skew > $O/placement_skew.txt 2>&1
timeout 300 $P idxskew > $O/placement_idxskew.txt 2>&1
for mb in 2 64 1024; do for i in 1 2 3; do VMM_CHUNK_MB=$mb timeout 200 $P vmm >> $O/placement_vmm.txt 2>&1; done; done
timeout 1500 python -m pytest tests/test_gpu_shard.py tests/test_gpu_index_parse.py "tests/test_gpu_scan.py::test_last_kernel_reports_what_ran_without_an_arena" "tests/test_gpu_parse.py::test_config3_4k30_100k_nals" -x -q > $O/pytest_call1.txt 2>&1
tail -5 $O/pytest_call1.txt
cat $O/placement_sep.txt $O/placement_slab.txt $O/placement_realloc.txt
