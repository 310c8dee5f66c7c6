#!/bin/bash
# round 6: bfloat16 features on the LDS-patch gather -- the fused reduction (bit 1) and C = 128 (bit 2) with the new chunk set-up, against today's selection
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
rm -f $O/r06_b16_ab.txt
for rep in 1 2; do
  for m in 0 3; do
    export DDMP_B16_PATCH_AB=$m
    echo "== DDMP_B16_PATCH_AB=$m (round $rep)" >> $O/r06_b16_ab.txt
    timeout 600 python3 scripts/microbench.py spmm --dtype bf16 --order rcb --rotate 3 --widths 512,256,128 --iters 12 2>/dev/null | grep "^spmm" >> $O/r06_b16_ab.txt
  done
done
unset DDMP_B16_PATCH_AB
cat $O/r06_b16_ab.txt
