#!/bin/bash
# Which overlap pays: PosNet beside NormalNet (two streams, default) or the weight gradients beside the dgrad chain (DDMP_ASYNC_WGRAD=1,
# one main stream + one wgrad stream; the combination of both does not survive hipStreamEndCapture), each as one replayed hipGraph.
#   gpurun -- 'bash scripts/stream_ab.sh > gpurun_out/stream_ab.txt 2>&1'
cd "$(dirname "$0")/.."
B="python3 bench.py --no-cpu-baseline --extras 0 --bf16-extra 0 --mode-ab 0 --steps 20 --warmup 5"
run() { echo -n "$1: "; shift; env "$@" timeout 300 $B ${EXTRA} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms/step  loss %.6f' % (d['ms_per_step'], d['loss']))"; }
for rep in 1 2; do
EXTRA="" run "two streams (default)        " X=1
EXTRA="--overlap 0" run "one stream                   " X=1
EXTRA="--overlap 0" run "one stream + wgrad stream    " DDMP_ASYNC_WGRAD=1
EXTRA="--overlap 0 --graph 0" run "eager, one stream            " X=1
EXTRA="--overlap 0 --graph 0" run "eager, one + wgrad stream    " DDMP_ASYNC_WGRAD=1
EXTRA="--graph 0" run "eager, two streams + 2 wgrad " DDMP_ASYNC_WGRAD=1
done
