#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time python3 -m pytest tests/test_gpu_switches.py tests/test_gpu_kernels.py tests/test_gpu_bf16.py tests/test_gpu_dist.py tests/test_gpu_multi.py -q -m gpu -p no:cacheprovider -x ) > gpurun_out/r06_p4.txt 2>&1
( time python3 -m pytest tests/test_gpu_path.py -q -m gpu -p no:cacheprovider -x -k "fusion or layer_order or graph_replay or timed_configuration or smoke" ) >> gpurun_out/r06_p4.txt 2>&1
grep -E "passed|failed|error" gpurun_out/r06_p4.txt; tail -30 gpurun_out/r06_p4.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
