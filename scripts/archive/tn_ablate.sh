#!/bin/bash
# gpurun -- 'bash scripts/tn_ablate.sh > gpurun_out/tn_ablate.txt 2>&1'   (builds: scripts/build_ablation_tn.sh rN "-DDDMP_TN_ABLATE=N")
cd "$(dirname "$0")/.."
python3 scripts/tn_ablate.py
for v in r1 r2 r4 r8 r6 r14 r13 r9; do
  [ -f build_abl/libddmp_$v.so ] && DDMP_LIB=$PWD/build_abl/libddmp_$v.so python3 scripts/tn_ablate.py
done
python3 scripts/tn_ablate.py
