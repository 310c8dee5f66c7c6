#!/bin/bash
# Round 5, conversion of the f16x3 operands by v_fma_mix (ddmp_common.h f16_split_quad): same-box A/B against the library built
# from HEAD (build_abl/libddmp_head.so, not committed): per-form times, output bits, then the step.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/r05_split_ab.txt
B="--no-cpu-baseline --extras 0 --bf16-extra 0 --profile-steps 0 --mode-ab 0 --parity 0 --irregular 0 --steps 20 --warmup 5"
{
DDMP_LIB=$PWD/build_abl/libddmp_head.so python3 scripts/split_ab_probe.py 2>&1 | grep -v amdgpu > gpurun_out/split_head.txt
python3 scripts/split_ab_probe.py 2>&1 | grep -v amdgpu > gpurun_out/split_tree.txt
paste -d'\n' gpurun_out/split_head.txt gpurun_out/split_tree.txt
echo "== bit-identical outputs?"
diff <(awk '{print $2,$3,$4,$NF}' gpurun_out/split_head.txt) <(awk '{print $2,$3,$4,$NF}' gpurun_out/split_tree.txt) && echo "all checksums equal"
for rep in 1 2 3; do
  echo "== step, HEAD library"; DDMP_LIB=$PWD/build_abl/libddmp_head.so python3 bench.py $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'])"
  echo "== step, this tree"; python3 bench.py $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'])"
done
} > $O 2>&1
cat $O
