#!/bin/bash
# 13k-face mesh: ms per iteration (eager / graph / graph + two streams) and the kernel list of three eager iterations
#   gpurun -- 'bash scripts/small_mesh_trace.sh > gpurun_out/small_mesh.txt 2>&1'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python3 scripts/small_mesh_probe.py 2>&1 | grep -v amdgpu.ids
cd /tmp; export TMPDIR=/tmp
MODE=trace rocprofv3 --kernel-trace --stats -d $O/small_prof -o small -- python3 $R/scripts/small_mesh_probe.py > $O/small_prof.log 2>&1
DB=$(ls $O/small_prof/*/*results.db $O/small_prof/*results.db 2>/dev/null | head -1)
cd $R
python3 scripts/rocpd_summary.py "$DB" --top 70
rm -rf $O/small_prof
