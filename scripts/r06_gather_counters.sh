#!/bin/bash
# Counter passes over scripts/r06_gather_counters_workload.py (run on the GPU box).  usage: r06_gather_counters.sh [C]  (round 6: the round-5 passes of scripts/archive/spmm_pmc.sh on the final tree)
C=${1:-512}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/spmm_pmc_$C
mkdir -p $O
rm -f $O/summary.txt
cd /tmp; export TMPDIR=/tmp
i=0
for set in "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
           "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCP_TOTAL_READ_sum TCP_TAGRAM0_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o r -- python3 $R/scripts/r06_gather_counters_workload.py $C > $O/p$i.log 2>&1
  DB=$(ls $O/p$i/*/*results.db $O/p$i/*results.db 2>/dev/null | head -1)
  if [ -n "$DB" ]; then python3 $R/scripts/pmc_dump.py "$DB" spmm >> $O/summary.txt; else echo "pass $i ($set): no database" >> $O/summary.txt; tail -3 $O/p$i.log >> $O/summary.txt; fi
  rm -rf $O/p$i
done
cat $O/summary.txt
