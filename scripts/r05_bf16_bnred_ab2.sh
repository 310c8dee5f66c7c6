cd /root/repo
B="--dtype bf16 --no-cpu-baseline --extras 0 --bf16-extra 0 --profile-steps 0 --mode-ab 0 --parity 0 --irregular 0 --steps 20 --warmup 5"
J='import sys,json; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print(d["ms_per_step"])'
for rep in 1 2 3; do
  echo "== off";  DDMP_BF16_GEMM_BNRED=0 python3 bench.py $B 2>/dev/null | python3 -c "$J"
  echo "== M>=128"; python3 bench.py $B 2>/dev/null | python3 -c "$J"
  echo "== M>=256"; DDMP_BF16_GEMM_BNRED_MINM=256 python3 bench.py $B 2>/dev/null | python3 -c "$J"
done
