#!/bin/bash
# round 5: side figures on the final tree -- the partitioned path at world size 1, the 13k-face mesh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash scripts/dist_overhead.sh > gpurun_out/r05_dist_overhead.txt 2>&1
python3 scripts/small_mesh_probe.py > gpurun_out/r05_small_mesh.txt 2>&1
cat gpurun_out/r05_dist_overhead.txt gpurun_out/r05_small_mesh.txt | grep -v amdgpu.ids
