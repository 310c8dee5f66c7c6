#!/bin/bash
# round 6, call 3: (a) the single-launch partial reduction for <= 2560 groups at the 125k-face shard (DDMP_FP_MID=0/1, interleaved);
# (b) the 13k-face mesh with the large-mesh routes' row thresholds lowered (are the thresholds right for launch-bound sizes?)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
B="--no-cpu-baseline --extras 0 --bf16-extra 0 --profile-steps 0 --mode-ab 0 --parity 0 --steps 30 --warmup 5 --gate-open 0"
for rep in 1 2 3; do
  for m in 0 1; do
    ms=$(DDMP_FP_MID=$m timeout 240 python3 bench.py --faces 125000 $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['loss'])")
    echo "faces 125000 single, DDMP_FP_MID=$m: ms_per_step, loss = $ms" | tee -a $O/r06_p2_fp_mid.txt
  done
done
run13() {
  echo "== $*" | tee -a $O/r06_p2_small_thresholds.txt
  env "$@" python3 scripts/small_mesh_probe.py 2>&1 | grep "graph=1" | tee -a $O/r06_p2_small_thresholds.txt
}
run13 DDMP_NOP=1
run13 DDMP_RR_MIN_ROWS=1000
run13 DDMP_PANEL_MIN_ROWS=1000 DDMP_RR_MIN_ROWS=1000
run13 DDMP_TN_PANEL_MIN_ROWS=1000
run13 DDMP_PANEL_MIN_ROWS=1000 DDMP_RR_MIN_ROWS=1000 DDMP_TN_PANEL_MIN_ROWS=1000
run13 DDMP_NOP=1
