#!/bin/bash
# Run on the GPU box (gpurun), ONCE per round on the final tree: the bench line + kernel table, a rocprofv3 kernel trace of the
# bench command in the configuration the driver times (one replayed hipGraph per iteration, PosNet on a second stream) cut to
# its timed region (bench.py brackets it with ddmp_trace_marker_kernel: no priming iteration, no set-up in the statistics),
# two PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs, kernel trace only) cut the same way, and the summaries that get
# committed under profiles/.
#   usage: collect_evidence.sh TAG [f32|bf16]
TAG=${1:-rXX}
DT=${2:-f32}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
SUF=""; [ "$DT" = "bf16" ] && SUF="_bf16"
STEPS=5
PROF="--dtype $DT --steps $STEPS --warmup 3 --no-cpu-baseline --profile-steps 0 --extras 0 --bf16-extra 0"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/${TAG}${SUF}_prof -o bench -- python3 $R/bench.py $PROF > $O/${TAG}${SUF}_prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${TAG}${SUF}_pmc_F -o r -- python3 $R/bench.py $PROF > $O/${TAG}${SUF}_pmc_F.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/${TAG}${SUF}_pmc_W -o r -- python3 $R/bench.py $PROF > $O/${TAG}${SUF}_pmc_W.log 2>&1
cd $R
DB=$(ls $O/${TAG}${SUF}_prof/*/*results.db $O/${TAG}${SUF}_prof/*results.db 2>/dev/null | head -1)
python3 scripts/rocpd_summary.py "$DB" --top 60 --between-markers $STEPS > $O/${TAG}${SUF}_bench_kernel_stats.txt 2>> $O/${TAG}${SUF}_prof.log
FD=$(ls $O/${TAG}${SUF}_pmc_F/*/*results.db $O/${TAG}${SUF}_pmc_F/*results.db 2>/dev/null | head -1)
WD=$(ls $O/${TAG}${SUF}_pmc_W/*/*results.db $O/${TAG}${SUF}_pmc_W/*results.db 2>/dev/null | head -1)
python3 scripts/pmc_traffic.py "$FD" "$WD" "bench.py $PROF: the $STEPS timed iterations between the trace markers (replayed hipGraph, two streams)" $DT $STEPS > $O/${TAG}${SUF}_pmc_hbm_traffic.json 2>> $O/${TAG}${SUF}_prof.log
# the bench line itself, with this tree's PMC file in place so that it carries roofline.traffic
mkdir -p $R/profiles
cp $O/${TAG}${SUF}_pmc_hbm_traffic.json $R/profiles/
python3 bench.py --dtype $DT --steps 20 --warmup 5 --kernel-table $O/${TAG}${SUF}_bench_kernel_table_hip_events.json > $O/${TAG}${SUF}_bench_line.json 2> $O/${TAG}${SUF}_bench.err
tail -1 $O/${TAG}${SUF}_bench_line.json | cut -c1-400
# keep the merge-back small: the raw databases stay on the box
rm -rf $O/${TAG}${SUF}_prof $O/${TAG}${SUF}_pmc_F $O/${TAG}${SUF}_pmc_W
ls -la $O | grep ${TAG}${SUF}
