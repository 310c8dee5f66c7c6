#!/bin/bash
# Round 5: bf16 transform-first dgrads with the BatchNorm-backward reductions in the epilogue (ddmp_gemm_nn_bnred_bf16):
# tests, then interleaved A/B of the bf16 step against DDMP_BF16_GEMM_BNRED=0 (plain dgrad + bn_bwd_reduce pass).
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/r05_bf16_bnred_ab.txt
B="--dtype bf16 --no-cpu-baseline --extras 0 --bf16-extra 0 --profile-steps 0 --mode-ab 0 --parity 0 --irregular 0 --steps 20 --warmup 5"
{
timeout 900 python3 -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "bnred or nn_bf16 or statistics" 2>&1 | tail -5
for rep in 1 2 3; do
  echo "== bf16 step, DDMP_BF16_GEMM_BNRED=0"; DDMP_BF16_GEMM_BNRED=0 python3 bench.py $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'])"
  echo "== bf16 step, default"; python3 bench.py $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'])"
done
} > $O 2>&1
cat $O
