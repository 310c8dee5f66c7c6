#!/usr/bin/env python3
"""bf16 transform-first dgrad: fused reductions epilogue (ddmp_gemm_nn_bnred_bf16) vs plain dgrad + bn_bwd_reduce, us per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dual_dmp_amd import ops
dev = torch.device("cuda:0")
BF = torch.bfloat16


def timeit(fn, it=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for n in (1000000, 500000):
    for M, K in ((256, 512), (128, 256), (64, 128)):
        a = torch.randn(n, M, device=dev).to(BF); yp = torch.randn(n, K, device=dev).to(BF)
        w = torch.randn(M, K, device=dev) / M ** 0.5
        bn4 = torch.rand(4, K, device=dev) + 0.5
        sums = torch.zeros(2 * K, dtype=torch.float64, device=dev)
        out = torch.empty(n, K, device=dev, dtype=BF)
        t_plain = timeit(lambda: ops.gemm_nn(a, w, out=out))
        t_red = timeit(lambda: ops.bn_bwd_reduce(out, yp, bn4, sums2=sums))
        t_fused = timeit(lambda: ops.gemm_nn_bnred(a, w, yp, bn4, sums, out=out)) if ops.gemm_nn_bnred_supported(M, K, n, BF) else float("nan")
        print("n=%7d  %3d -> %3d   dgrad %6.0f + reduce %6.0f = %6.0f us   fused %6.0f us" % (n, M, K, t_plain, t_red, t_plain + t_red, t_fused), flush=True)
