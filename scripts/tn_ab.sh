#!/bin/bash
# A/B of the wide wgrad kernels: gpurun -- 'bash scripts/tn_ab.sh > gpurun_out/tn_ab.txt 2>&1'
cd "$(dirname "$0")/.."
echo "== round-4 kernel (row-major staging), same segment order on all waves (DDMP_TN_PP=0)"; DDMP_TN_PP=0 python3 scripts/tn_ablate.py
echo "== round-4 kernel, the two waves of a SIMD in opposite segment order (DDMP_TN_PP=1)"; DDMP_TN_PP=1 python3 scripts/tn_ablate.py
echo "== round-3 panel kernel (DDMP_TN_RM=0)"; DDMP_TN_RM=0 python3 scripts/tn_ablate.py
