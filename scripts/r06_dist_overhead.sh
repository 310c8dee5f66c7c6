#!/bin/bash
# The partitioned path at world size 1 (DDMP_FORCE_DIST=1) against the single-device path at the per-rank sizes of a 1M-face mesh on
# 8 and 4 GPUs (125k / 250k faces): eager RCCL path (two communicators, two streams), RCCL loopback (every exchange really goes
# through RCCL), the whole iteration captured into one hipGraph -- on one stream when RCCL calls are inside the capture, on two where
# none is (one rank, no loopback).  (The two-stream capture WITH RCCL calls -- one communicator, PosNet's stream forked inside the
# capture -- was measured once in round 6 through a switch that has since been removed: SIGSEGV, profiles/r06_dist_overhead.txt.)   gpurun -- 'bash scripts/r06_dist_overhead.sh > gpurun_out/r06_dist_overhead.txt 2>&1'
cd "$(dirname "$0")/.."
B="--no-cpu-baseline --extras 0 --bf16-extra 0 --profile-steps 0 --mode-ab 0 --parity 0 --steps 30 --warmup 5 --gate-open 0"
for F in ${FACES:-125000 250000}; do
  for cfg in "single" "dist" "dist+loopback" "dist+loopback+graph" "dist+graph" "single"; do
    env_="DDMP_NOP=1"
    case $cfg in
      dist) env_="DDMP_FORCE_DIST=1";;
      dist+loopback) env_="DDMP_FORCE_DIST=1 DDMP_COMM_LOOPBACK=1";;
      dist+loopback+graph) env_="DDMP_FORCE_DIST=1 DDMP_COMM_LOOPBACK=1 DDMP_DIST_GRAPH=1";;     # RCCL inside the capture: ONE stream
      dist+graph) env_="DDMP_FORCE_DIST=1 DDMP_DIST_GRAPH=1";;                               # no RCCL call at one rank: two streams
    esac
    ms=$(env $env_ timeout 240 python3 bench.py --faces $F $B 2>gpurun_out/r06_dist_err.txt | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['loss'])")
    echo "faces $F  $cfg: ms_per_step, loss = $ms"
    [ -z "$ms" ] && tail -5 gpurun_out/r06_dist_err.txt
  done
done
