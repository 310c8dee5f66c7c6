#!/bin/bash
# round-4 probes on the GPU box: narrow wgrads, gather with / without the LDS-patch selection
cd "$(dirname "$0")/.."
echo "== narrow wgrads (1M rows)"; python3 scripts/tn_narrow_probe.py 2>&1 | grep -v amdgpu.ids
echo "== gather, default selection"; python3 scripts/microbench.py spmm --order morton 2>&1 | grep -v amdgpu.ids | grep "^spmm\|prologue"
echo "== gather, DDMP_SPMM_PATCH=0"; DDMP_SPMM_PATCH=0 python3 scripts/microbench.py spmm --order morton 2>&1 | grep -v amdgpu.ids | grep "^spmm\|prologue"
