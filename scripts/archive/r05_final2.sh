#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 1500 python3 -m pytest tests/test_gpu_irregular.py tests/test_gpu_kernels.py tests/test_gpu_bf16.py -x -q -m gpu -k "not under_switch" 2>&1 | tail -8 ) > gpurun_out/r05e_tests.txt
{
echo "== regular torus, rcb order, default selection (heavy-chunk launches split over slab groups)"; timeout 300 python3 scripts/microbench.py spmm --order rcb --widths 512,256,128,64,32 2>&1 | grep "^spmm\|prologue\|statistics"
echo "== flipped torus (10 rounds + hub 24), rcb order, default"; timeout 400 python3 scripts/microbench.py spmm --order rcb --flip 10 --widths 512,256,128 2>&1 | grep "^spmm\|prologue\|statistics"
echo "== regular torus, rcb order, DDMP_SPMM_PATCH=0 (lean kernel only)"; DDMP_SPMM_PATCH=0 timeout 300 python3 scripts/microbench.py spmm --order rcb --widths 512,256 2>&1 | grep "^spmm\|prologue\|statistics"
echo "== flipped torus, rcb order, DDMP_SPMM_PATCH=0"; DDMP_SPMM_PATCH=0 timeout 400 python3 scripts/microbench.py spmm --order rcb --flip 10 --widths 512,256 2>&1 | grep "^spmm\|prologue\|statistics"
echo "== regular torus, MORTON order, default"; timeout 300 python3 scripts/microbench.py spmm --order morton --widths 512,256 2>&1 | grep "^spmm\|prologue\|statistics"
} > gpurun_out/r05e_gather_forms.txt 2>&1
tail -3 gpurun_out/r05e_tests.txt
if grep -q "passed" gpurun_out/r05e_tests.txt && ! grep -q "failed" gpurun_out/r05e_tests.txt; then bash scripts/r05_evidence.sh; fi
