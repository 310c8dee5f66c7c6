#!/bin/bash
cd "$(dirname "$0")/.."
bash scripts/r05_fullsuite.sh
bash scripts/r05_evidence.sh
python3 scripts/small_mesh_probe.py > gpurun_out/r05_small_mesh.txt 2>&1
grep -v amdgpu gpurun_out/r05_small_mesh.txt
