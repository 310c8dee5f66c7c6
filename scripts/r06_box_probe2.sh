#!/bin/bash
# round 6, final tree: the step and the two roofline fractions on whatever box this call lands on (one line per call; profiles/r06_box_probe_final.txt)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
mkdir -p gpurun_out
timeout 600 python3 bench.py --no-cpu-baseline --mode-ab 0 --parity 0 --bf16-extra 0 --extras 0 --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1])
r, g = d['roofline'], d['roofline_gather']
print('step %.3f ms | GEMM %.2f ms frac %.4f (8d %.4f) | gather %.2f ms frac %.4f (8d %.4f; face %.4f vertex %.4f) | best copy %.0f GB/s, local-graph ceiling %.0f / %.0f GB/s, gather = %.3f of ceiling'
      % (d['ms_per_step'], r['ms_per_step'], r['frac'], r['frac_survey_8d'], g['ms_per_step'], g['frac'], g['frac_survey_8d'], g['by_fan_in']['4']['frac'],
         g['by_fan_in']['7']['frac'], g['device_copy_GBs'], g['by_fan_in']['4']['local_graph_ceiling_GBs'], g['by_fan_in']['7']['local_graph_ceiling_GBs'],
         g['ceiling']['frac_of_ceiling']))" | tee -a gpurun_out/r06_box_probe_final.txt
