#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time python3 -m pytest tests/test_gpu_bench.py -q -m gpu -p no:cacheprovider -x ) > gpurun_out/r06_p6_test.txt 2>&1
tail -5 gpurun_out/r06_p6_test.txt
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_line_v2.json 2> gpurun_out/r06_bench_v2.err
python3 -c "
import json
d=json.loads(open('gpurun_out/r06_bench_line_v2.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['frac'], d['roofline_gather']['frac'], d['roofline']['traffic'])
print(json.dumps(d['parity_1m']['iter2_teacher_forced']))
print({k:d['parity_1m'][k] for k in ('rel','max_abs_dpos','max_abs_dnorm','mad_delta_deg','ok')})
"
