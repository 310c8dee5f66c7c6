#!/bin/bash
# round 6: split chunks (oversized patches walked as two 32-row halves in the same launch: no heavy list on regular meshes) against the same library with
# the split switched off in the table builder (DDMP_TMP_NOSPLIT, a temporary knob: heavy list + second launch as before), interleaved on one box
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
rm -f $O/r06_split_ab.txt
B="--no-cpu-baseline --extras 0 --bf16-extra 0 --profile-steps 0 --mode-ab 0 --parity 0 --steps 20 --warmup 5"
for rep in 1 2 3; do
  for m in heavy split; do
    if [ $m = heavy ]; then export DDMP_TMP_NOSPLIT=1; else unset DDMP_TMP_NOSPLIT; fi
    ms=$(timeout 300 python3 bench.py $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['loss'])")
    echo "1M faces, $m: ms_per_step, loss = $ms" | tee -a $O/r06_split_ab.txt
  done
done
for m in heavy split; do
  if [ $m = heavy ]; then export DDMP_TMP_NOSPLIT=1; else unset DDMP_TMP_NOSPLIT; fi
  echo "== $m" >> $O/r06_split_ab.txt
  timeout 600 python3 scripts/microbench.py spmm --order rcb --rotate 3 --widths 512,256,128 --iters 12 2>/dev/null | grep "^spmm" >> $O/r06_split_ab.txt
done
unset DDMP_TMP_NOSPLIT
for F in 125000; do
  for m in heavy split; do
    if [ $m = heavy ]; then export DDMP_TMP_NOSPLIT=1; else unset DDMP_TMP_NOSPLIT; fi
    ms=$(timeout 300 python3 bench.py $B --faces $F 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['loss'])")
    echo "$F faces, $m: ms_per_step, loss = $ms" | tee -a $O/r06_split_ab.txt
  done
done
tail -22 $O/r06_split_ab.txt
