#!/bin/bash
# round 6: graph-sized entry tables in the LDS-patch gather (more workgroups per CU) against the build before (experiments/tmp/libddmp_new1.so)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
rm -f $O/r06_pe_ab.txt
for rep in 1 2; do
  for lib in new1 new2; do
    if [ $lib = new1 ]; then export DDMP_LIB=$R/experiments/tmp/libddmp_new1.so; else unset DDMP_LIB; fi
    echo "== $lib (round $rep)" >> $O/r06_pe_ab.txt
    timeout 600 python3 scripts/microbench.py spmm --order rcb --rotate 3 --widths 512,256,128 --iters 12 2>/dev/null | grep "^spmm" >> $O/r06_pe_ab.txt
  done
done
unset DDMP_LIB
cat $O/r06_pe_ab.txt
B="--no-cpu-baseline --extras 0 --bf16-extra 0 --profile-steps 0 --mode-ab 0 --parity 0 --steps 20 --warmup 5"
for rep in 1 2 3; do
  for lib in new1 new2; do
    if [ $lib = new1 ]; then export DDMP_LIB=$R/experiments/tmp/libddmp_new1.so; else unset DDMP_LIB; fi
    ms=$(timeout 300 python3 bench.py $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['loss'])")
    echo "1M faces, $lib: ms_per_step, loss = $ms" | tee -a $O/r06_pe_ab.txt
  done
done
