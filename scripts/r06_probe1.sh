#!/bin/bash
# round 6, call 2: the new / restructured tests, the 13k-face iteration under rocprofv3 (eager kernel list + timeline of the replayed graph),
# the LDS-patch gather below 64k rows at the 125k-face shard.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
( time python3 -m pytest tests/test_gpu_path.py -q -m gpu -p no:cacheprovider -k "long_horizon or teacher_forced" -x --durations=10 ) > $O/r06_p1_tests.txt 2>&1
( time python3 -m pytest tests/test_gpu_irregular.py -q -m gpu -p no:cacheprovider -k "under_switch" -x ) >> $O/r06_p1_tests.txt 2>&1
tail -30 $O/r06_p1_tests.txt
B="--no-cpu-baseline --extras 0 --bf16-extra 0 --profile-steps 0 --mode-ab 0 --parity 0 --steps 30 --warmup 5 --gate-open 0"
for mr in 65536 16384; do
  for rep in 1 2; do
    ms=$(DDMP_SPMM_PATCH_MIN_ROWS=$mr timeout 240 python3 bench.py --faces 125000 $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['loss'])")
    echo "faces 125000 single, DDMP_SPMM_PATCH_MIN_ROWS=$mr: ms_per_step, loss = $ms" | tee -a $O/r06_p1_patch_min_rows.txt
  done
done
cd /tmp; export TMPDIR=/tmp
MODE=trace rocprofv3 --kernel-trace --stats -d $O/small_prof -o small -- python3 $R/scripts/small_mesh_probe.py > $O/small_prof.log 2>&1
DB=$(ls $O/small_prof/*/*results.db $O/small_prof/*results.db 2>/dev/null | head -1)
python3 $R/scripts/rocpd_summary.py "$DB" --top 70 > $O/r06_p1_small_kernels.txt
python3 $R/scripts/r06_timeline.py "$DB" --tail-frac 0.3 >> $O/r06_p1_small_kernels.txt
rm -rf $O/small_prof
for S in 1 2; do
MODE=trace_graph STREAMS=$S rocprofv3 --kernel-trace -d $O/small_prof -o small -- python3 $R/scripts/small_mesh_probe.py > $O/small_prof.log 2>&1
DB=$(ls $O/small_prof/*/*results.db $O/small_prof/*results.db 2>/dev/null | head -1)
echo "== replayed graph, $S stream(s)" >> $O/r06_p1_small_timeline.txt
python3 $R/scripts/r06_timeline.py "$DB" --tail-frac 0.5 >> $O/r06_p1_small_timeline.txt
rm -rf $O/small_prof
done
cat $O/r06_p1_small_timeline.txt
