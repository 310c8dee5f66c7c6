#!/bin/bash
# round 6, last call: evidence on the final kernel sources (collect_evidence.sh: bench line, rocprofv3 stats, PMC traffic; f32 and bf16), the launch-bound
# sizes, then the whole GPU suite + smoke() on the same tree
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash scripts/collect_evidence.sh r06 f32 > gpurun_out/r06_collect_f32.log 2>&1
bash scripts/collect_evidence.sh r06 bf16 > gpurun_out/r06_collect_bf16.log 2>&1
python3 scripts/small_mesh_probe.py 2>&1 | grep -v amdgpu > gpurun_out/r06_small_mesh_final.txt
B="--no-cpu-baseline --extras 0 --bf16-extra 0 --profile-steps 0 --mode-ab 0 --parity 0 --steps 30 --warmup 5"
for F in 125000 250000 500000; do
  ms=$(timeout 240 python3 bench.py --faces $F $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['loss'])")
  echo "faces $F single (gate open): ms_per_step, loss = $ms" >> gpurun_out/r06_small_mesh_final.txt
done
cat gpurun_out/r06_small_mesh_final.txt
( time timeout 3000 python3 -m pytest tests -q -m gpu -p no:cacheprovider --durations=15 -x ) > gpurun_out/r06_full_gpu_suite.txt 2>&1
tail -4 gpurun_out/r06_full_gpu_suite.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
