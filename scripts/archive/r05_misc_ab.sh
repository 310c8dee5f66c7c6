#!/bin/bash
# Round 5, late: new bf16 test + switch, stream-priority A/B of the float32 step (PosNet's stream at high priority through a
# temporary DDMP_SIDE_PRIORITY hook in trainer.py, since removed: 43.01 / 42.74 / 43.48 ms default vs 43.06 / 42.95 / 42.92 -- noise)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/r05_misc_ab.txt
B="--no-cpu-baseline --extras 0 --bf16-extra 0 --profile-steps 0 --mode-ab 0 --parity 0 --irregular 0 --steps 20 --warmup 5"
J='import sys,json; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print(d["ms_per_step"])'
{
timeout 900 python3 -m pytest tests/test_gpu_bf16.py -x -q -m gpu -k "bnred" 2>&1 | tail -3
timeout 900 python3 -m pytest tests/test_gpu_switches.py -x -q -m gpu -k "BNRED" 2>&1 | tail -3
for rep in 1 2 3; do
  echo "== f32 step, default"; python3 bench.py $B 2>/dev/null | python3 -c "$J"
  echo "== f32 step, DDMP_SIDE_PRIORITY=-1"; DDMP_SIDE_PRIORITY=-1 python3 bench.py $B 2>/dev/null | python3 -c "$J"
done
} > $O 2>&1
cat $O
