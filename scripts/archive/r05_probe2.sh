#!/bin/bash
# round 5, second GPU pass: new gather forms (statistics on the LDS-patch kernel, sorted rows), parity, step time
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 1500 python3 -m pytest tests/test_gpu_irregular.py -x -q -m gpu -k "not under_switch" 2>&1 | tail -15 ) > gpurun_out/r05b_irregular_tests.txt
( timeout 900 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "spmm or tail_fused" 2>&1 | tail -15 ) > gpurun_out/r05b_kernel_spmm_tests.txt
( timeout 1500 python3 -m pytest tests/test_gpu_path.py -x -q -m gpu -k "bnf or irregular or gcnconv" 2>&1 | tail -15 ) > gpurun_out/r05b_path_tests.txt
( timeout 1500 python3 -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "float64" 2>&1 | tail -15 ) > gpurun_out/r05b_fullsize_tests.txt
{
echo "== regular torus, rcb order, default selection (round 5)"; timeout 300 python3 scripts/microbench.py spmm --order rcb --widths 512,256,128 2>&1 | grep "^spmm\|prologue\|statistics"
echo "== flipped torus (10 rounds + hub 24), rcb order, default"; timeout 400 python3 scripts/microbench.py spmm --order rcb --flip 10 --widths 512,256,128 2>&1 | grep "^spmm\|prologue\|statistics"
echo "== regular torus, rcb order, DDMP_SPMM=row (round-1 row kernel), narrow widths"; DDMP_SPMM=row timeout 300 python3 scripts/microbench.py spmm --order rcb --widths 128,64,32 2>&1 | grep "^spmm\|prologue\|statistics"
echo "== regular torus, rcb order, default, narrow widths"; timeout 300 python3 scripts/microbench.py spmm --order rcb --widths 128,64,32 2>&1 | grep "^spmm\|prologue\|statistics"
} > gpurun_out/r05b_gather_forms.txt 2>&1
timeout 900 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --mode-ab 0 --bf16-extra 0 --kernel-table gpurun_out/r05b_kernel_table.json > gpurun_out/r05b_bench.json 2> gpurun_out/r05b_bench.err
for f in gpurun_out/r05b_*_tests.txt; do echo $f; tail -3 $f; done
