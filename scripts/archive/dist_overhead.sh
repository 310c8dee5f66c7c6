#!/bin/bash
# The partitioned path at world size 1 (DDMP_FORCE_DIST=1) against the single-device path at the per-rank sizes of a 1M-face
# mesh on 8 and 4 GPUs (125k / 250k faces): eager RCCL path, RCCL loopback (every exchange really goes through RCCL), and the
# whole iteration captured into one hipGraph.   gpurun -- 'bash scripts/dist_overhead.sh > gpurun_out/dist_overhead.txt 2>&1'
cd "$(dirname "$0")/.."
B="--no-cpu-baseline --extras 0 --bf16-extra 0 --profile-steps 0 --mode-ab 0 --parity 0 --steps 30 --warmup 5"
for F in 125000 250000; do
  for cfg in "single" "dist" "dist+loopback" "dist+loopback+graph" "dist+graph"; do
    env_=""
    case $cfg in
      dist) env_="DDMP_FORCE_DIST=1";;
      dist+loopback) env_="DDMP_FORCE_DIST=1 DDMP_COMM_LOOPBACK=1";;
      dist+loopback+graph) env_="DDMP_FORCE_DIST=1 DDMP_COMM_LOOPBACK=1 DDMP_DIST_GRAPH=1";;
      dist+graph) env_="DDMP_FORCE_DIST=1 DDMP_DIST_GRAPH=1";;
    esac
    ms=$(env $env_ timeout 240 python3 bench.py --faces $F $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['loss'])")
    echo "faces $F  $cfg: ms_per_step, loss = $ms"
  done
done
