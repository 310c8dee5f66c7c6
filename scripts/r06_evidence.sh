#!/bin/bash
# round 6: evidence on the final tree (run once on the GPU box): bench line + rocprofv3 kernel stats (two-stream and serialised) + PMC HBM
# traffic for float32 and bfloat16 features (scripts/collect_evidence.sh), the launch-bound sizes, the partitioned path at one rank.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash scripts/collect_evidence.sh r06 f32 > gpurun_out/r06_collect_f32.log 2>&1
bash scripts/collect_evidence.sh r06 bf16 > gpurun_out/r06_collect_bf16.log 2>&1
tail -2 gpurun_out/r06_collect_f32.log gpurun_out/r06_collect_bf16.log
python3 scripts/small_mesh_probe.py 2>&1 | grep -v amdgpu > gpurun_out/r06_small_mesh_final.txt
B="--no-cpu-baseline --extras 0 --bf16-extra 0 --profile-steps 0 --mode-ab 0 --parity 0 --steps 30 --warmup 5"
for go in 0 1; do
  ms=$(timeout 240 python3 bench.py --faces 125000 $B --gate-open $go 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['loss'])")
  echo "faces 125000 single, --gate-open $go: ms_per_step, loss = $ms" >> gpurun_out/r06_small_mesh_final.txt
done
cat gpurun_out/r06_small_mesh_final.txt
