#!/bin/bash
# Diagnostic builds of libddmp_hip.so with -D flags on gemm_tn_rm.hip (timing-only ablations of the wide wgrad; results invalid).
#   scripts/build_ablation_tn.sh NAME "-DDDMP_TN_ABLATE=3" [NAME2 "-D..."] ...   ->  build_abl/libddmp_NAME.so
set -e
cd "$(dirname "$0")/.."
mkdir -p build_abl
make -s -C dual-dmp_amd/csrc
others=$(ls dual-dmp_amd/csrc/*.o | grep -v '/gemm_tn_rm.o')
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $flags -c dual-dmp_amd/csrc/gemm_tn_rm.hip -o build_abl/gemm_tn_rm_$name.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_abl/libddmp_$name.so build_abl/gemm_tn_rm_$name.o $others ) &
done
wait
ls build_abl/*.so
