#!/bin/bash
# What would fewer W bytes per flop buy the wide f16x3 GEMM?  A/B builds of gemm_rr_kernel with parts of its work removed
# (results are WRONG in these builds; DDMP_RR_ABLATE bits: 1 = half the W fragment reads from LDS, 2 = half the W copies
# L2 -> LDS, 4 = two MFMA products instead of three), each timed in a loop with rocm-smi power / clock samples.
#   gpurun -- 'bash scripts/rr_ablation.sh > gpurun_out/rr_ablation.txt 2>&1'
cd "$(dirname "$0")/.."
for ABL in 0 1 2 3 4 7; do
  touch dual-dmp_amd/csrc/gemm.hip
  make -C dual-dmp_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -DDDMP_RR_ABLATE=$ABL" >/dev/null 2>&1 || { echo build failed; exit 1; }
  echo "== DDMP_RR_ABLATE=$ABL"
  python3 scripts/rr_ablation_probe.py
done
touch dual-dmp_amd/csrc/gemm.hip
make -C dual-dmp_amd/csrc >/dev/null 2>&1
