#!/bin/bash
# Run on the GPU box (gpurun): bench line + kernel table, rocprofv3 kernel trace, two PMC passes.  TAG = r01c ...
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python bench.py --kernel-table $O/${TAG}_kernel_table.json > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
tail -1 $O/${TAG}_bench.json | cut -c1-600
cd /tmp; export TMPDIR=/tmp
PROF="--steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 --graph 0 --overlap 0"
rocprofv3 --kernel-trace --stats -d $O/${TAG}_prof -o bench -- python3 $R/bench.py $PROF > $O/${TAG}_prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${TAG}_pmc_F -o r -- python3 $R/bench.py $PROF > $O/${TAG}_pmc_F.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/${TAG}_pmc_W -o r -- python3 $R/bench.py $PROF > $O/${TAG}_pmc_W.log 2>&1
ls -la $O/${TAG}_prof $O/${TAG}_pmc_F $O/${TAG}_pmc_W
