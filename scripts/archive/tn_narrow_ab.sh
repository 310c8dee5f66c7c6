#!/bin/bash
# The 128 <-> 256 wgrads on the f16x3 panels of 256 x 128 / 128 x 256 (default) against the tiled bf16x6 kernel they replace
# (DDMP_TN_NARROW_PANELS=0): us per call at 1M rows, and the step.   gpurun -- 'bash scripts/tn_narrow_ab.sh'
cd "$(dirname "$0")/.."
for v in 1 0; do echo "== DDMP_TN_NARROW_PANELS=$v"; DDMP_TN_NARROW_PANELS=$v TN_SHAPES="256x128 128x256" python3 scripts/tn_ablate.py; done
B="python3 bench.py --no-cpu-baseline --extras 0 --bf16-extra 0 --mode-ab 0 --steps 20 --warmup 5"
for rep in 1 2; do for v in 0 1; do
  DDMP_TN_NARROW_PANELS=$v timeout 300 $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DDMP_TN_NARROW_PANELS=$v  %.3f ms/step  gemm_tn %.3f ms  loss %.6f' % (d['ms_per_step'], d['kernel_ms_per_step']['gemm_tn'], d['loss']))"
done; done
