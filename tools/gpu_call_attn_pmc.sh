# issue / wait counters of the short-sequence attention kernels (image shape of the bench)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=$GRAFT_REPO_ROOT/gpurun_out; T=${1:-p1}
cd /tmp; export TMPDIR=/tmp
timeout 120 rocprofv3 -L > $O/${T}_counters_avail.txt 2>&1
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/ap_$i -o p -- python $GRAFT_REPO_ROOT/tools/attn_one.py ${WHICH:-image} 6 > $O/${T}_pass$i.log 2>&1
  find /tmp/ap_$i -name "*counter_collection.csv" -exec cp {} $O/${T}_pass$i.csv \;
done
ls -la $O/${T}_pass*.csv 2>&1 | tail -5; grep -c . $O/${T}_counters_avail.txt
