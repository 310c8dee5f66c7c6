#!/bin/bash
# round 5: evidence on the final tree (run once on the GPU box)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash scripts/collect_evidence.sh r05 f32 > gpurun_out/r05_collect_f32.log 2>&1
bash scripts/collect_evidence.sh r05 bf16 > gpurun_out/r05_collect_bf16.log 2>&1
bash scripts/spmm_pmc.sh 512 > gpurun_out/r05_gather_tcp_tcc_counters_raw.txt 2>&1
tail -2 gpurun_out/r05_collect_f32.log gpurun_out/r05_collect_bf16.log
