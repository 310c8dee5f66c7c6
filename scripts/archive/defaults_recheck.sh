#!/bin/bash
# Are the default routes still the fastest on the current kernels?  Each A/B switch against the default, interleaved, one box.
#   gpurun -- 'bash scripts/defaults_recheck.sh'
cd "$(dirname "$0")/.."
B="python3 bench.py --no-cpu-baseline --extras 0 --bf16-extra 0 --mode-ab 0 --steps 20 --warmup 5"
for rep in 1 2; do
for cfg in "X=1" "DDMP_RR_PM2_WIDE=1" "DDMP_GEMM_BNRED=0" "DDMP_GEMM_BNRED_NARROW=0" "DDMP_BNBWD_NARROW=0" "DDMP_BNBWD_L0=0" "DDMP_SPMM_BNBWD=0" "DDMP_TAIL_FUSE=0" "DDMP_PREP_WEIGHTS=0" "DDMP_GEMM_WS=0"; do
  env $cfg timeout 300 $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %.3f ms/step' % ('$cfg', d['ms_per_step']))"
done; done
