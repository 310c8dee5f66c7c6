#!/bin/bash
# Library with the experimental LDS-patch SpMM compiled in: build_abl/libddmp_exp.so (A/B runs: DDMP_LIB=... DDMP_SPMM_PATCH=1)
set -e
cd "$(dirname "$0")/.."
mkdir -p build_abl/exp
cd dual-dmp_amd/csrc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -DDDMP_WITH_SPMM_PATCH -I."
for f in spmm spmm_b16 graph; do /opt/rocm/bin/hipcc $FL -c $f.hip -o ../../build_abl/exp/$f.o & done
/opt/rocm/bin/hipcc $FL -c experimental/spmm_patch.hip -o ../../build_abl/exp/spmm_patch.o &
wait
make -s
others=$(ls *.o | grep -v -E '^(spmm|spmm_b16|graph)\.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_abl/libddmp_exp.so $others ../../build_abl/exp/*.o
ls -la ../../build_abl/libddmp_exp.so
