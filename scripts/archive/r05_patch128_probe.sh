cd /root/repo
for m in default 1; do
  echo "== f32 widths 128,64, DDMP_SPMM_PATCH=$m"
  if [ $m = default ]; then python3 scripts/microbench.py spmm --order rcb --widths 128,64 2>&1 | grep -v amdgpu
  else DDMP_SPMM_PATCH=1 python3 scripts/microbench.py spmm --order rcb --widths 128,64 2>&1 | grep -v amdgpu; fi
done
