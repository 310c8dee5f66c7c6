#!/bin/bash
# Upper bound of the strong-scaling curve from ONE device: the per-rank shard of the 1M-face mesh on N = 2 / 4 / 8 GPUs (500k / 250k / 125k faces)
# through (a) the single-device path and (b) the partitioned path at one rank with every exchange through RCCL (loopback), without and with the
# overlap mode -- what a rank would take if its peers' halo rows cost nothing.  T(1M, one device) / (N x T(shard)) bounds the efficiency.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
B="--no-cpu-baseline --extras 0 --bf16-extra 0 --profile-steps 0 --mode-ab 0 --parity 0 --steps 20 --warmup 5"
run() { env "$@" timeout 400 python3 bench.py --faces $F $B 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'])"; }
for F in 1000000 500000 250000 125000; do
  a=$(run X=1); b=$(run DDMP_FORCE_DIST=1 DDMP_COMM_LOOPBACK=1 DDMP_DIST_SPLIT=0); c=$(run DDMP_FORCE_DIST=1 DDMP_COMM_LOOPBACK=1 DDMP_DIST_SPLIT=1)
  echo "faces $F: single $a ms | partitioned (one rank, RCCL loopback) $b ms | ... overlap mode $c ms" | tee -a gpurun_out/r06_scaling_bound.txt
done
