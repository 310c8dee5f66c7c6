#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( time timeout 3000 python3 -m pytest tests -q -m gpu -p no:cacheprovider --durations=25 -x ) > gpurun_out/r06_suite.txt 2>&1
tail -45 gpurun_out/r06_suite.txt
