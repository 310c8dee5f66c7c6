#!/bin/bash
# Run on the GPU box (gpurun): bench line + kernel table, rocprofv3 kernel trace, two PMC passes (FETCH_SIZE / WRITE_SIZE
# in separate runs, kernel trace only), and the summaries that get committed under profiles/.
#   usage: collect_evidence.sh TAG [f32|bf16]
TAG=${1:-rXX}
DT=${2:-f32}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
SUF=""; [ "$DT" = "bf16" ] && SUF="_bf16"
python3 bench.py --dtype $DT --steps 20 --warmup 5 --kernel-table $O/${TAG}${SUF}_bench_kernel_table_hip_events.json > $O/${TAG}${SUF}_bench_line.json 2> $O/${TAG}${SUF}_bench.err
tail -1 $O/${TAG}${SUF}_bench_line.json | cut -c1-400
cd /tmp; export TMPDIR=/tmp
PROF="--dtype $DT --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 --graph 0 --overlap 0 --extras 0"
rocprofv3 --kernel-trace --stats -d $O/${TAG}${SUF}_prof -o bench -- python3 $R/bench.py $PROF > $O/${TAG}${SUF}_prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${TAG}${SUF}_pmc_F -o r -- python3 $R/bench.py $PROF > $O/${TAG}${SUF}_pmc_F.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/${TAG}${SUF}_pmc_W -o r -- python3 $R/bench.py $PROF > $O/${TAG}${SUF}_pmc_W.log 2>&1
cd $R
DB=$(ls $O/${TAG}${SUF}_prof/*/*results.db $O/${TAG}${SUF}_prof/*results.db 2>/dev/null | head -1)
python3 scripts/rocpd_summary.py "$DB" --top 60 > $O/${TAG}${SUF}_bench_kernel_stats.txt 2>> $O/${TAG}${SUF}_prof.log
FD=$(ls $O/${TAG}${SUF}_pmc_F/*/*results.db $O/${TAG}${SUF}_pmc_F/*results.db 2>/dev/null | head -1)
WD=$(ls $O/${TAG}${SUF}_pmc_W/*/*results.db $O/${TAG}${SUF}_pmc_W/*results.db 2>/dev/null | head -1)
python3 scripts/pmc_traffic.py "$FD" "$WD" "bench.py $PROF (4 eager iterations per pass)" $DT 4 > $O/${TAG}${SUF}_pmc_hbm_traffic.json 2>> $O/${TAG}${SUF}_prof.log
# second bench run with this tree's PMC file in place, so that the line carries roofline.traffic (the first line is kept)
mkdir -p $R/profiles
cp $O/${TAG}${SUF}_pmc_hbm_traffic.json $R/profiles/
mv $O/${TAG}${SUF}_bench_line.json $O/${TAG}${SUF}_bench_line_first_run.json
python3 bench.py --dtype $DT --steps 20 --warmup 5 --kernel-table $O/${TAG}${SUF}_bench_kernel_table_hip_events.json > $O/${TAG}${SUF}_bench_line.json 2>> $O/${TAG}${SUF}_bench.err
# keep the merge-back small: the raw databases stay on the box
rm -rf $O/${TAG}${SUF}_prof $O/${TAG}${SUF}_pmc_F $O/${TAG}${SUF}_pmc_W
ls -la $O | grep ${TAG}${SUF}
