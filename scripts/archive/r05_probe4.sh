#!/bin/bash
# round 5, fourth GPU pass: precomputed entry weights, bf16 on the LDS-patch kernel
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 1500 python3 -m pytest tests/test_gpu_irregular.py tests/test_gpu_kernels.py tests/test_gpu_bf16.py -x -q -m gpu -k "not under_switch" 2>&1 | tail -8 ) > gpurun_out/r05d_tests.txt
{
echo "== regular torus, rcb order, default selection (entry weights precomputed)"; timeout 300 python3 scripts/microbench.py spmm --order rcb --widths 512,256,128,64,32 2>&1 | grep "^spmm\|prologue\|statistics"
echo "== bf16, regular torus, rcb order, default"; timeout 300 python3 scripts/microbench.py spmm --order rcb --dtype bf16 --widths 512,256,128 2>&1 | grep "^spmm\|prologue\|statistics"
echo "== bf16, regular torus, rcb order, DDMP_SPMM_PATCH=1"; DDMP_SPMM_PATCH=1 timeout 300 python3 scripts/microbench.py spmm --order rcb --dtype bf16 --widths 512,256,128 2>&1 | grep "^spmm\|prologue\|statistics"
} > gpurun_out/r05d_gather_forms.txt 2>&1
timeout 900 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --mode-ab 0 --irregular 0 --kernel-table gpurun_out/r05d_kernel_table.json > gpurun_out/r05d_bench.json 2> gpurun_out/r05d_bench.err
tail -3 gpurun_out/r05d_tests.txt
