#!/bin/bash
# the per-rank size of an 8-GPU run of the 1M mesh (125k faces) on one device, final tree: rocprofv3 --kernel-trace --stats of 5 eager one-stream iterations
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p125 -o bench -- python3 $R/bench.py --faces 125000 --steps 5 --warmup 3 --no-cpu-baseline --profile-steps 0 --extras 0 --bf16-extra 0 --graph 0 --overlap 0 > $R/gpurun_out/r06_125k_prof.log 2>&1
cd $R
DB=$(ls gpurun_out/p125/*/*results.db gpurun_out/p125/*results.db 2>/dev/null | head -1)
python3 scripts/rocpd_summary.py "$DB" --top 60 --between-markers 5 > gpurun_out/r06_shard_125k_kernel_stats_serialised.txt 2>&1
rm -rf gpurun_out/p125
head -12 gpurun_out/r06_shard_125k_kernel_stats_serialised.txt | cut -c1-170
