#!/bin/bash
# the per-rank size of an 8-GPU run of the 1M mesh (125k faces) on one device: where does the time go?
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python3 bench.py --faces 125000 --steps 30 --warmup 5 --no-cpu-baseline --mode-ab 0 --bf16-extra 0 --irregular 0 --extras 0 --kernel-table gpurun_out/r05_125k_kernel_table.json > gpurun_out/r05_125k_bench.json 2> gpurun_out/r05_125k_bench.err
python3 bench.py --faces 250000 --steps 30 --warmup 5 --no-cpu-baseline --mode-ab 0 --bf16-extra 0 --irregular 0 --extras 0 --kernel-table gpurun_out/r05_250k_kernel_table.json > gpurun_out/r05_250k_bench.json 2> gpurun_out/r05_250k_bench.err
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p125 -o bench -- python3 $R/bench.py --faces 125000 --steps 5 --warmup 3 --no-cpu-baseline --profile-steps 0 --extras 0 --bf16-extra 0 --graph 0 --overlap 0 > $R/gpurun_out/r05_125k_prof.log 2>&1
cd $R
DB=$(ls gpurun_out/p125/*/*results.db gpurun_out/p125/*results.db 2>/dev/null | head -1)
python3 scripts/rocpd_summary.py "$DB" --top 70 --between-markers 5 > gpurun_out/r05_125k_kernel_stats_serialised.txt 2>&1
rm -rf gpurun_out/p125
head -3 gpurun_out/r05_125k_kernel_stats_serialised.txt | cut -c1-200
