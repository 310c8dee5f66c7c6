#!/bin/bash
# round 6: gate-open against gate-closed iterations as the MAIN timed region of separate bench.py processes, interleaved (the in-run "other gate" figure of the
# evidence run came out 1 ms above the open one: measurement order or a real effect?)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
rm -f $O/r06_gate_check.txt
B="--no-cpu-baseline --bf16-extra 0 --profile-steps 0 --mode-ab 0 --parity 0 --steps 20 --warmup 5"
for rep in 1 2 3; do
  for g in 1 0; do
    timeout 300 python3 bench.py $B --gate-open $g 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('--gate-open $g: main', d['ms_per_step'], 'open', d.get('gate_open_ms_per_step'), 'closed', d.get('gate_closed_ms_per_step'))" | tee -a $O/r06_gate_check.txt
  done
done
