#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time python3 -m pytest tests/test_gpu_multi.py -q -m gpu -p no:cacheprovider -x -k "LOOPBACK or extra" ) > gpurun_out/r06_p3_multi.txt 2>&1
tail -15 gpurun_out/r06_p3_multi.txt
FACES=125000 bash scripts/r06_dist_overhead.sh > gpurun_out/r06_dist_overhead.txt 2>&1
cat gpurun_out/r06_dist_overhead.txt
