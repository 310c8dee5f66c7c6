#!/bin/bash
# A/B of the wide wgrad kernels: gpurun -- 'bash scripts/tn_ab.sh > gpurun_out/tn_ab.txt 2>&1'
cd "$(dirname "$0")/.."
for pp in 0 1; do echo "== round-4 kernel DDMP_TN_PP=$pp (0: same segment order on all waves; 1: waves 0-3 | 4-7 opposite)"; DDMP_TN_PP=$pp python3 scripts/tn_ablate.py; done
echo "== round-3 panel kernel (DDMP_TN_RM=0)"; DDMP_TN_RM=0 python3 scripts/tn_ablate.py
