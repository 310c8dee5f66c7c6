#!/bin/bash
# round 6: after the prologue change, the LDS-patch gather forced at the narrow widths (DDMP_SPMM_PATCH=1) against the default selection (C >= 256)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
rm -f $O/r06_pn_ab.txt
for rep in 1 2; do
  for m in default forced; do
    if [ $m = forced ]; then export DDMP_SPMM_PATCH=1; else unset DDMP_SPMM_PATCH; fi
    echo "== $m (round $rep)" >> $O/r06_pn_ab.txt
    timeout 600 python3 scripts/microbench.py spmm --order rcb --rotate 3 --widths 128,64 --iters 12 2>/dev/null | grep "^spmm" >> $O/r06_pn_ab.txt
  done
done
cat $O/r06_pn_ab.txt
