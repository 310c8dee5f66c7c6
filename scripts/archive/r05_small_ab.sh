#!/bin/bash
# Round 5, late: single-stage partial reduction for few groups (fpartials_single_kernel): 13k-face iteration + the tests that run small meshes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
python3 scripts/small_mesh_probe.py 2>&1 | grep -v amdgpu
python3 scripts/small_mesh_probe.py 2>&1 | grep -v amdgpu
timeout 1500 python3 -m pytest tests/test_gpu_path.py tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -4
} > gpurun_out/r05_small_ab.txt 2>&1
cat gpurun_out/r05_small_ab.txt
