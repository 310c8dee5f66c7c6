#!/bin/bash
# sample the shader clock while a GEMM microbenchmark runs (diagnostic)
cd "$(dirname "$0")/.."
python scripts/microbench.py gemm --iters ${1:-300} > /tmp/mb.log 2>&1 &
pid=$!
sleep 8
for i in $(seq 1 12); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.5; done
wait $pid
grep -E "K=512 M=512" /tmp/mb.log
