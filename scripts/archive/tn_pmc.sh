#!/bin/bash
# SQ counter passes over scripts/tn_pmc_workload.py (run on the GPU box): gpurun -- 'bash scripts/tn_pmc.sh'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/tn_pmc
mkdir -p $O
rm -f $O/summary.txt
cd /tmp; export TMPDIR=/tmp
for which in ${WHICH_LIST:-bnbwd plain}; do
  export TN_WHICH=$which
  echo "=== $which  DDMP_TN_PP=${DDMP_TN_PP:-0}" >> $O/summary.txt
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_WAVES"; do
    i=$((i+1))
    rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o r -- python3 $R/scripts/tn_pmc_workload.py > $O/p$i.log 2>&1
    DB=$(ls $O/p$i/*/*results.db $O/p$i/*results.db 2>/dev/null | head -1)
    if [ -n "$DB" ]; then python3 $R/scripts/pmc_dump.py "$DB" gemm_tn >> $O/summary.txt; else echo "pass $i ($set): no database" >> $O/summary.txt; tail -3 $O/p$i.log >> $O/summary.txt; fi
    rm -rf $O/p$i
  done
done
cat $O/summary.txt
