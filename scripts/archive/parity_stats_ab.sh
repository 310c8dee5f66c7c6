#!/bin/bash
# parity_1m (incl. the float64 forward) with the gather's statistics epilogue on (default) and off, same box:
#   gpurun -- 'bash scripts/parity_stats_ab.sh > gpurun_out/parity_stats_ab.txt 2>&1'
cd "$(dirname "$0")/.."
for st in 1 0; do
  echo "== DDMP_SPMM_STATS=$st"
  DDMP_SPMM_STATS=$st timeout 900 python3 bench.py --cpu-iters 1 --extras 0 --bf16-extra 0 --mode-ab 0 --steps 5 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
p=d['parity_1m']
print(json.dumps({k:p[k] for k in p if k not in ('what','later_iterations_note','bounds')}, indent=1))
print('cpu:', d['cpu_baseline']['sample'][-150:])"
done
