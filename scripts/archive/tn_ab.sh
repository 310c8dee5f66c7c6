#!/bin/bash
# A/B of the wide wgrad kernels: gpurun -- 'bash scripts/tn_ab.sh > gpurun_out/tn_ab.txt 2>&1'
cd "$(dirname "$0")/.."
for pp in 0 1; do echo "== round-4 kernel DDMP_TN_PP=$pp (0: same segment order on all waves; 1: waves 0-3 | 4-7 opposite)"; DDMP_TN_PP=$pp python3 scripts/tn_ablate.py; done
# (the round-3 panel kernel is no longer dispatched for these shapes; its figures in profiles/r04_tn_kernel_ab.txt are from the
#  build that still had both behind DDMP_TN_RM)
