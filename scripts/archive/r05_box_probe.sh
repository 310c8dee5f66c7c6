#!/bin/bash
# one line per call: the step and its two families on whatever box this call got (run several times: the pool's boxes differ)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python3 bench.py --no-cpu-baseline --bf16-extra 0 --mode-ab 0 --parity 0 --irregular 0 --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
g=d['roofline_gather']; r=d['roofline']; c=d.get('device_copy',{}).get('ddmp_copy_probe_GBs',{})
print('step %.2f ms  gemm %.2f ms (%.3f)  gather %.2f ms (%.3f; face %.3f vertex %.3f)  copy probe %s / %s GB/s' % (d['ms_per_step'], r['ms_per_step'], r['frac'], g['ms_per_step'], g['frac'], g['by_fan_in']['4']['frac'], g['by_fan_in']['7']['frac'], c.get('plain'), c.get('nontemporal')))
" | tee -a gpurun_out/r05_box_probe.txt
