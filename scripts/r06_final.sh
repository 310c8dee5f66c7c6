#!/bin/bash
# round 6, last call: the whole GPU suite + smoke() on the final tree, and the partitioned bench path at one rank with the exchange stream
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 3000 python3 -m pytest tests -q -m gpu -p no:cacheprovider --durations=15 -x ) > gpurun_out/r06_full_gpu_suite.txt 2>&1
tail -30 gpurun_out/r06_full_gpu_suite.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
B="--no-cpu-baseline --extras 0 --bf16-extra 0 --profile-steps 0 --mode-ab 0 --parity 0 --steps 10 --warmup 3 --faces 125000"
DDMP_FORCE_DIST=1 DDMP_DIST_SPLIT=1 DDMP_COMM_LOOPBACK=1 timeout 300 python3 bench.py $B 2>gpurun_out/r06_final_dist.err | cut -c1-300
tail -3 gpurun_out/r06_final_dist.err
