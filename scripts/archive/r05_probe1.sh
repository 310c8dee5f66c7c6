#!/bin/bash
# round 5, first GPU pass: irregular-graph parity, then the gather under Morton / RCB numbering, regular / flipped meshes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 1500 python3 -m pytest tests/test_gpu_irregular.py -x -q -m gpu 2>&1 | tail -25 ) > gpurun_out/r05_irregular_tests.txt
( timeout 900 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "spmm or tail_fused" 2>&1 | tail -15 ) > gpurun_out/r05_kernel_spmm_tests.txt
{
for order in morton rcb; do
  echo "== regular torus, $order order, default selection"; timeout 300 python3 scripts/microbench.py spmm --order $order --widths 512,256,128 2>&1 | grep "^spmm\|prologue\|statistics"
done
echo "== regular torus, rcb order, DDMP_SPMM_PATCH=1 (LDS-patch forced)"; DDMP_SPMM_PATCH=1 timeout 300 python3 scripts/microbench.py spmm --order rcb --widths 512,256,128 2>&1 | grep "^spmm\|prologue\|statistics"
echo "== regular torus, rcb order, DDMP_SPMM_PATCH=0 (lean)"; DDMP_SPMM_PATCH=0 timeout 300 python3 scripts/microbench.py spmm --order rcb --widths 512,256,128,64,32 2>&1 | grep "^spmm\|prologue\|statistics"
echo "== flipped torus (10 rounds + hub 24), rcb order, default"; timeout 400 python3 scripts/microbench.py spmm --order rcb --flip 10 --widths 512,256,128 2>&1 | grep "^spmm\|prologue\|statistics"
echo "== flipped torus, rcb order, DDMP_SPMM_PATCH=1"; DDMP_SPMM_PATCH=1 timeout 400 python3 scripts/microbench.py spmm --order rcb --flip 10 --widths 512,256 2>&1 | grep "^spmm\|prologue\|statistics"
echo "== flipped torus, rcb order, DDMP_SPMM_PATCH=0"; DDMP_SPMM_PATCH=0 timeout 400 python3 scripts/microbench.py spmm --order rcb --flip 10 --widths 512,256 2>&1 | grep "^spmm\|prologue\|statistics"
} > gpurun_out/r05_gather_orders.txt 2>&1
tail -5 gpurun_out/r05_irregular_tests.txt gpurun_out/r05_kernel_spmm_tests.txt
