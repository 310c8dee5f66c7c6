B="python3 bench.py --no-cpu-baseline --extras 0 --bf16-extra 0 --mode-ab 0 --steps 20 --warmup 5"
for rep in 1 2; do
for cfg in "MAXNNZ=8 NE=1" "MAXNNZ=5 NE=1" "MAXNNZ=5 NE=0"; do
  set -- $cfg
  env DDMP_SPMM_PATCH_${1} DDMP_SPMM_PATCH_${2} timeout 300 $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
g=d['roofline_gather']
print('$cfg  %.3f ms/step  spmm %.3f ms  gather frac %.4f  %s' % (d['ms_per_step'], d['kernel_ms_per_step']['spmm'], g['frac'], {k:v['frac'] for k,v in g['by_fan_in'].items()}))"
done
done
