# round 4, GPU call 2: counters of fast and slow processes (the harness prints its own kernel time; no torch, program directly after --)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04/pmc_place; mkdir -p $O
P=./build/ubench/placement
run_set() {  # name mode counters...
  name=$1; mode=$2; shift 2
  for i in 1 2 3 4 5; do
    d=$O/${name}_${mode}_$i
    timeout 200 rocprofv3 --pmc "$@" --output-format csv -d $d -- $P $mode 1677000 3 > $d.out 2> $d.err
    f=$(find $d -name "*counter_collection.csv" | head -1)
    echo "== $name $mode $i: $(grep -h ' ms ' $d.out | tail -1)" >> $O/summary.txt
    python3 scripts/pmc_sum.py $d k_scan_extract4 >> $O/summary.txt 2>&1
    rm -rf $d
  done
}
run_set A sep TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum
run_set B sep TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum
run_set C sep TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WR_UNCACHED_32B_sum TCC_EA0_RDREQ_sum
run_set D sep TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum GRBM_UTCL2_BUSY
run_set E sep TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_GMI_32B_sum TCC_EA0_WRREQ_WRITE_GMI_32B_sum
run_set F sep TCC_HIT_sum TCC_MISS_sum TCC_NORMAL_WRITEBACK_sum TCC_NORMAL_EVICT_sum
cat $O/summary.txt
