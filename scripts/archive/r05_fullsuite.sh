#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 5400 python3 -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -40 > gpurun_out/r05_full_gpu_suite.txt
tail -5 gpurun_out/r05_full_gpu_suite.txt
