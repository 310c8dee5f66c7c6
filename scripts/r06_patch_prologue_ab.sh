#!/bin/bash
# round 6: the LDS-patch gather's prologue (one round trip to the tables, entry weights precomputed per graph) against the library
# before it (experiments/tmp/libddmp_old.so = the tree of profiles/r06_bench_line.json), interleaved on ONE box; parity of the forms first.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
( timeout 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_irregular.py -q -m gpu -p no:cacheprovider -x -k "spmm or gather or irregular and not under_switch" 2>&1 | tail -5 ) > $O/r06_pp_tests.txt 2>&1
cat $O/r06_pp_tests.txt
for rep in 1 2; do
  for lib in old new; do
    if [ $lib = old ]; then export DDMP_LIB=$R/experiments/tmp/libddmp_old.so; else unset DDMP_LIB; fi
    echo "== $lib (round $rep)" >> $O/r06_pp_ab.txt
    timeout 600 python3 scripts/microbench.py spmm --order rcb --rotate 3 --widths ${WIDTHS:-512,256,128,64,32} --iters 12 2>/dev/null | grep "^spmm" >> $O/r06_pp_ab.txt
  done
done
unset DDMP_LIB
cat $O/r06_pp_ab.txt
