#!/bin/bash
# the step on this box (the pool's boxes differ by a few percent: the GEMM family runs against the socket's power cap): 20 timed steps, gate open,
# with the family times of the profiled pass.   gpurun -- 'bash scripts/r06_box_probe.sh'   (one box per call; appended by hand to profiles/r06_box_probe.txt)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python3 bench.py --no-cpu-baseline --extras 0 --bf16-extra 0 --mode-ab 0 --parity 0 --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
g = d['roofline']; s = d['roofline_gather']
print('step %.3f ms (gate open) | gemm %.2f ms frac %.3f / 8d %.3f | gather %.2f ms frac %.3f / 8d %.3f | copy probe %s' % (d['ms_per_step'], g['ms_per_step'], g['frac'], g['frac_survey_8d'], s['ms_per_step'], s['frac'], s['frac_survey_8d'], d['device_copy']['best_GBs']))
" | tee gpurun_out/r06_box_probe_line.txt
