#!/bin/bash
# round 6: the overlap mode at ONE rank with every exchange through RCCL (DDMP_COMM_LOOPBACK=1): what the exchange stream costs when there
# is nothing to hide (no halo rows at one rank) -- interleaved; and the new row-slice tests + the multi tests.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
B="--no-cpu-baseline --extras 0 --bf16-extra 0 --profile-steps 0 --mode-ab 0 --parity 0 --steps 30 --warmup 5 --faces 125000 --gate-open 0"
for rep in 1 2; do
  for sp in 0 1; do
    ms=$(DDMP_FORCE_DIST=1 DDMP_COMM_LOOPBACK=1 DDMP_DIST_SPLIT=$sp timeout 300 python3 bench.py $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['loss'])")
    echo "faces 125000 dist+loopback, DDMP_DIST_SPLIT=$sp: ms_per_step, loss = $ms" | tee -a gpurun_out/r06_p5_overlap_cost.txt
  done
done
( time python3 -m pytest tests/test_gpu_irregular.py tests/test_gpu_multi.py tests/test_gpu_dist.py -q -m gpu -p no:cacheprovider -x -k "row_slices or rccl or split" ) > gpurun_out/r06_p5_tests.txt 2>&1
tail -8 gpurun_out/r06_p5_tests.txt
