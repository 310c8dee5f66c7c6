#!/bin/bash
# SQ counter passes over scripts/rr_pmc_workload.py for both wide-GEMM kernels (run on the GPU box).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/rr_pmc
mkdir -p $O
rm -f $O/summary.txt
cd /tmp; export TMPDIR=/tmp
for rr in ${RR_LIST:-1 0}; do
  export DDMP_GEMM_RR=$rr
  echo "=== DDMP_GEMM_RR=$rr" >> $O/summary.txt
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_WAVES" \
             "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o r -- python3 $R/scripts/rr_pmc_workload.py > $O/p$i.log 2>&1
    DB=$(ls $O/p$i/*/*results.db $O/p$i/*results.db 2>/dev/null | head -1)
    if [ -n "$DB" ]; then python3 $R/scripts/pmc_dump.py "$DB" gemm_ >> $O/summary.txt; else echo "pass $i ($set): no database" >> $O/summary.txt; tail -3 $O/p$i.log >> $O/summary.txt; fi
    rm -rf $O/p$i
  done
done
cat $O/summary.txt
