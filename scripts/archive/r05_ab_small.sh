#!/bin/bash
# same-box A/B of the round-4 tree against this tree at small sizes.  Needs the round-4 tree exported next to this one first:
#   mkdir -p experiments/ab_r04 && git archive 162167b | tar -x -C experiments/ab_r04      (not committed)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( cd experiments/ab_r04 && make -C dual-dmp_amd/csrc -j32 > /dev/null 2>&1; ls -la dual-dmp_amd/csrc/libddmp_hip.so )
B="--no-cpu-baseline --extras 0 --bf16-extra 0 --profile-steps 0 --mode-ab 0 --parity 0 --steps 30 --warmup 5"
{
for rep in 1 2; do
  echo "== round-4 tree, 13k faces (rep $rep)"; ( cd experiments/ab_r04 && python3 scripts/small_mesh_probe.py 2>&1 | grep -v amdgpu )
  echo "== this tree, 13k faces (rep $rep)"; python3 scripts/small_mesh_probe.py 2>&1 | grep -v amdgpu
  for F in 125000; do
    echo "== round-4 tree, $F faces single device"; ( cd experiments/ab_r04 && python3 bench.py --faces $F $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'])" )
    echo "== this tree, $F faces single device"; python3 bench.py --faces $F $B --irregular 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'])"
  done
done
} > gpurun_out/r05_ab_small.txt 2>&1
cat gpurun_out/r05_ab_small.txt
