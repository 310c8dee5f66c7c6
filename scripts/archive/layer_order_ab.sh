#!/bin/bash
# A/B of the layer order on equal-width layers (512->512, 256->256) and of the gather's statistics epilogue, same box:
#   gpurun -- 'bash scripts/layer_order_ab.sh > gpurun_out/layer_order_ab.txt 2>&1'
cd "$(dirname "$0")/.."
B="python3 bench.py --no-cpu-baseline --extras 0 --bf16-extra 0 --mode-ab 0 --parity 0 --steps 20 --warmup 5"
for rep in 1 2; do
for cfg in "agg 0" "transform 0" "transform 1" "agg 1"; do
  set -- $cfg
  DDMP_EQUAL_WIDTH=$1 DDMP_SPMM_STATS=$2 timeout 300 $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('equal-width=%-9s spmm_stats=%s  %.3f ms/step  loss %.6f  %s' % ('$1','$2',d['ms_per_step'],d['loss'],d['kernel_ms_per_step']))"
done
done
