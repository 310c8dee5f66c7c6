#!/bin/bash
# Counter passes over scripts/gemm_pmc_workload.py (run on the GPU box).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/gemm_pmc
mkdir -p $O
rm -f $O/summary.txt
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_MEM_VIOLATIONS" \
           "SQ_INSTS_VALU_CVT SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INSTS_SALU" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o r -- python3 $R/scripts/gemm_pmc_workload.py > $O/p$i.log 2>&1
  DB=$(ls $O/p$i/*/*results.db $O/p$i/*results.db 2>/dev/null | head -1)
  if [ -n "$DB" ]; then python3 $R/scripts/pmc_dump.py "$DB" gemm_ >> $O/summary.txt; else echo "pass $i ($set): no database" >> $O/summary.txt; tail -3 $O/p$i.log >> $O/summary.txt; fi
  rm -rf $O/p$i
done
cat $O/summary.txt
