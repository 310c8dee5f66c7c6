#!/bin/bash
# round 6: the lean gather with TWO waves per workgroup (four rows per lane) at C <= 64 against four waves (DDMP_TMP_LEAN_NW2=0, a temporary knob)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
( timeout 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_irregular.py -q -m gpu -p no:cacheprovider -x -k "spmm or gather or irregular and not under_switch" 2>&1 | tail -3 ) > $O/r06_nw2_tests.txt 2>&1
cat $O/r06_nw2_tests.txt
rm -f $O/r06_nw2_ab.txt
for rep in 1 2; do
  for m in 0 64; do
    export DDMP_TMP_LEAN_NW2=$m
    echo "== two waves up to C = $m (round $rep)" >> $O/r06_nw2_ab.txt
    timeout 600 python3 scripts/microbench.py spmm --order rcb --rotate 3 --widths 64,32 --iters 12 2>/dev/null | grep "^spmm" >> $O/r06_nw2_ab.txt
  done
done
cat $O/r06_nw2_ab.txt
B="--no-cpu-baseline --extras 0 --bf16-extra 0 --profile-steps 0 --mode-ab 0 --parity 0 --steps 20 --warmup 5"
for rep in 1 2 3; do
  for m in 0 64; do
    export DDMP_TMP_LEAN_NW2=$m
    ms=$(timeout 300 python3 bench.py $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['loss'])")
    echo "1M faces, two waves up to C = $m: ms_per_step, loss = $ms" | tee -a $O/r06_nw2_ab.txt
  done
done
