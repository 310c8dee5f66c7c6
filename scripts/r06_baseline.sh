#!/bin/bash
# round 6, first call: the round-5 tree's GPU suite with per-test durations (what to cut: VERDICT r5 weak-12), the 13k-face mesh and
# the 125k-face shard single / partitioned (the figures VERDICT r5 next-1 is judged against) on ONE box.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python3 scripts/small_mesh_probe.py > gpurun_out/r06_base_small.txt 2>&1
B="--no-cpu-baseline --extras 0 --bf16-extra 0 --profile-steps 0 --mode-ab 0 --parity 0 --steps 30 --warmup 5"
for cfg in "single" "dist" "dist+loopback"; do
  env_=""
  case $cfg in
    dist) env_="DDMP_FORCE_DIST=1";;
    dist+loopback) env_="DDMP_FORCE_DIST=1 DDMP_COMM_LOOPBACK=1";;
  esac
  ms=$(env $env_ timeout 240 python3 bench.py --faces 125000 $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['loss'])")
  echo "faces 125000  $cfg: ms_per_step, loss = $ms" >> gpurun_out/r06_base_small.txt
done
cat gpurun_out/r06_base_small.txt
timeout 3000 python3 -m pytest tests -q -m gpu -p no:cacheprovider --durations=120 2>&1 | tail -160 > gpurun_out/r06_base_suite_durations.txt
tail -3 gpurun_out/r06_base_suite_durations.txt
