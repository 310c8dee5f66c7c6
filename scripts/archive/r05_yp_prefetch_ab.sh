#!/bin/bash
# Round 5, late: lean gather with the fused reduction -- the Yp row requested before the gather loop (A/B against the commit before:
# experiments/ab_prev exported by hand: git archive HEAD~0 ... ; here simply the narrow widths before / after in one run is not
# possible, so: forms at the narrow widths + the step, to be compared with profiles/r05_gather_forms.txt section A and the last step figures)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
B="--no-cpu-baseline --extras 0 --bf16-extra 0 --profile-steps 0 --mode-ab 0 --parity 0 --irregular 0 --steps 20 --warmup 5"
J='import sys,json; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print(d["ms_per_step"])'
{
for lib in prev tree prev tree; do
  if [ $lib = prev ]; then export DDMP_LIB=$PWD/build_abl/libddmp_prev.so; else unset DDMP_LIB; fi
  echo "== $lib: gather forms, narrow widths"; python3 scripts/microbench.py spmm --order rcb --widths 128,64,32 2>&1 | grep -v amdgpu | grep -E "spmm|reduce"
  echo "== $lib: step"; python3 bench.py $B 2>/dev/null | python3 -c "$J"
done
} > gpurun_out/r05_yp_prefetch_ab.txt 2>&1
cat gpurun_out/r05_yp_prefetch_ab.txt
