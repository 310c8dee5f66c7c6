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
# the same iterations SERIALISED (eager launches on one stream: per-kernel durations are not stretched by a second stream's
# kernels sharing the CUs) -- the summary whose averages the roofline fractions can be recomputed from
rocprofv3 --kernel-trace --stats -d $O/${TAG}${SUF}_prof_ser -o bench -- python3 $R/bench.py $PROF --graph 0 --overlap 0 > $O/${TAG}${SUF}_prof_ser.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${TAG}${SUF}_pmc_F -o r -- python3 $R/bench.py $PROF > $O/${TAG}${SUF}_pmc_F.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/${TAG}${SUF}_pmc_W -o r -- python3 $R/bench.py $PROF > $O/${TAG}${SUF}_pmc_W.log 2>&1
cd $R
DB=$(ls $O/${TAG}${SUF}_prof/*/*results.db $O/${TAG}${SUF}_prof/*results.db 2>/dev/null | head -1)
{ echo "# CONFIGURATION: the one the driver times -- one replayed hipGraph per iteration, PosNet on a second stream beside NormalNet."
  echo "# Kernels of the two streams share the CUs: per-kernel durations here are STRETCHED by co-residency and sum to more than the"
  echo "# wall time; the bench line's roofline fractions do NOT follow from this file but from the serialised one next to it"
  echo "# (${TAG}${SUF}_bench_kernel_stats_serialised.txt) and, independently, from bench.py's own HIP-event pass (eager, one stream)."
  python3 scripts/rocpd_summary.py "$DB" --top 60 --between-markers $STEPS; } > $O/${TAG}${SUF}_bench_kernel_stats.txt 2>> $O/${TAG}${SUF}_prof.log
DBS=$(ls $O/${TAG}${SUF}_prof_ser/*/*results.db $O/${TAG}${SUF}_prof_ser/*results.db 2>/dev/null | head -1)
{ echo "# CONFIGURATION: the same $STEPS iterations launched EAGERLY on ONE stream (bench.py --graph 0 --overlap 0): no two kernels"
  echo "# overlap, the average duration of a kernel here is what roofline.achieved = algorithmic bytes / duration is computed from"
  echo "# (bench.py measures the same thing with HIP events in its profiled pass: *_bench_kernel_table_hip_events.json)."
  python3 scripts/rocpd_summary.py "$DBS" --top 60 --between-markers $STEPS; } > $O/${TAG}${SUF}_bench_kernel_stats_serialised.txt 2>> $O/${TAG}${SUF}_prof.log
FD=$(ls $O/${TAG}${SUF}_pmc_F/*/*results.db $O/${TAG}${SUF}_pmc_F/*results.db 2>/dev/null | head -1)
WD=$(ls $O/${TAG}${SUF}_pmc_W/*/*results.db $O/${TAG}${SUF}_pmc_W/*results.db 2>/dev/null | head -1)
python3 scripts/pmc_traffic.py "$FD" "$WD" "bench.py $PROF: the $STEPS timed iterations between the trace markers (replayed hipGraph, two streams)" $DT $STEPS > $O/${TAG}${SUF}_pmc_hbm_traffic.json 2>> $O/${TAG}${SUF}_prof.log
# the bench line itself, with this tree's PMC file in place so that it carries roofline.traffic
mkdir -p $R/profiles
cp $O/${TAG}${SUF}_pmc_hbm_traffic.json $R/profiles/
NOCPU=""; [ "$DT" = "bf16" ] && NOCPU="--no-cpu-baseline"      # (the CPU baseline / parity leg belongs to the f32 line)
python3 bench.py --dtype $DT --steps 20 --warmup 5 $NOCPU --kernel-table $O/${TAG}${SUF}_bench_kernel_table_hip_events.json > $O/${TAG}${SUF}_bench_line.json 2> $O/${TAG}${SUF}_bench.err
tail -1 $O/${TAG}${SUF}_bench_line.json | cut -c1-400
# keep the merge-back small: the raw databases stay on the box
rm -rf $O/${TAG}${SUF}_prof $O/${TAG}${SUF}_prof_ser $O/${TAG}${SUF}_pmc_F $O/${TAG}${SUF}_pmc_W
ls -la $O | grep ${TAG}${SUF}
