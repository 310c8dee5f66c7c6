#!/bin/bash
# Round 5, late: the BatchNorm backward on the gather on the LDS-patch kernel (DDMP_SPMM_PATCH_FORMS bit 4) against the lean kernel
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/r05_bwd_patch_ab.txt
B="--no-cpu-baseline --extras 0 --bf16-extra 0 --profile-steps 0 --mode-ab 0 --parity 0 --irregular 0 --steps 20 --warmup 5"
J='import sys,json; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print(d["ms_per_step"])'
{
for fl in 0 10; do
  FLIP=$fl DDMP_SPMM_PATCH_FORMS=15 python3 scripts/bwd_patch_check.py 2>&1 | grep -v amdgpu
  FLIP=$fl python3 scripts/bwd_patch_check.py 2>&1 | grep -v amdgpu
done
timeout 900 python3 -m pytest tests/test_gpu_irregular.py tests/test_gpu_kernels.py -x -q -m gpu -k "gather or spmm or csr" 2>&1 | tail -3
for rep in 1 2 3; do
  echo "== step, lean BWD (FORMS=15)"; DDMP_SPMM_PATCH_FORMS=15 python3 bench.py $B 2>/dev/null | python3 -c "$J"
  echo "== step, default"; python3 bench.py $B 2>/dev/null | python3 -c "$J"
done
} > $O 2>&1
cat $O
