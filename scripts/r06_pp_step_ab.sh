#!/bin/bash
# round 6: the step with the gather's new chunk set-up (+ LDS-patch kernel from C = 128) against the library before it, interleaved on one box
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
( timeout 1200 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_irregular.py tests/test_gpu_path.py -q -m gpu -p no:cacheprovider -x -k "not under_switch and not teacher_forced and not long_horizon" 2>&1 | tail -5 ) > $O/r06_pp_tests2.txt 2>&1
cat $O/r06_pp_tests2.txt
rm -f $O/r06_pp_step.txt
B="--no-cpu-baseline --extras 0 --bf16-extra 0 --profile-steps 0 --mode-ab 0 --parity 0 --steps 20 --warmup 5"
for rep in 1 2 3; do
  for lib in old new; do
    if [ $lib = old ]; then export DDMP_LIB=$R/experiments/tmp/libddmp_old.so; else unset DDMP_LIB; fi
    ms=$(timeout 300 python3 bench.py $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['loss'])")
    echo "1M faces, $lib: ms_per_step, loss = $ms" | tee -a $O/r06_pp_step.txt
  done
done
unset DDMP_LIB
ms=$(timeout 300 python3 bench.py $B --faces 125000 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['loss'])")
echo "125k faces, new: $ms" | tee -a $O/r06_pp_step.txt
