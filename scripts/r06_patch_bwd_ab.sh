#!/bin/bash
# round 6: the BatchNorm backward on the gather as a form of the LDS-patch kernel (C <= 256) against the lean kernel's form (DDMP_SPMM_PATCH_BWD_MAXC=0)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
( timeout 900 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_irregular.py -q -m gpu -p no:cacheprovider -x -k "spmm or gather or irregular and not under_switch" 2>&1 | tail -5 ) > $O/r06_pb_tests.txt 2>&1
cat $O/r06_pb_tests.txt
rm -f $O/r06_pb_ab.txt
for rep in 1 2; do
  for m in lean patch; do
    if [ $m = lean ]; then export DDMP_SPMM_PATCH_BWD_MAXC=0; else export DDMP_SPMM_PATCH_BWD_MAXC=512; fi
    echo "== $m (round $rep)" >> $O/r06_pb_ab.txt
    timeout 600 python3 scripts/microbench.py spmm --order rcb --rotate 3 --widths 512,256,128,64 --iters 12 2>/dev/null | grep "^spmm" | sed 's/plain.*reduction/.../' >> $O/r06_pb_ab.txt
  done
done
unset DDMP_SPMM_PATCH_BWD_MAXC
cat $O/r06_pb_ab.txt
B="--no-cpu-baseline --extras 0 --bf16-extra 0 --profile-steps 0 --mode-ab 0 --parity 0 --steps 20 --warmup 5"
for rep in 1 2 3; do
  for m in lean patch; do
    if [ $m = lean ]; then export DDMP_SPMM_PATCH_BWD_MAXC=0; else unset DDMP_SPMM_PATCH_BWD_MAXC; fi
    ms=$(timeout 300 python3 bench.py $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['loss'])")
    echo "1M faces, bwd form on $m: ms_per_step, loss = $ms" | tee -a $O/r06_pb_ab.txt
  done
done
