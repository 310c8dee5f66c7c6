#!/usr/bin/env python3
"""Where are the errors of a panel GEMM in an f16 mode?  (development aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dual_dmp_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
n, K, M = 65536, 256, 256
A = torch.randn(n, K, device=dev); W = torch.randn(M, K, device=dev) / K ** 0.5
ref = (A.double() @ W.double().T)
for mode in (6, 13, 14):
    ops.set_gemm_mode(mode)
    for rep in range(3):
        Y = ops.gemm_nt(A, W)
        e = (Y.double() - ref).abs()
        rel = float((Y.double() - ref).norm() / ref.norm())
        bad = e > 1e-4
        print("mode", mode, "rep", rep, "rel %.2e" % rel, "max %.2e" % float(e.max()), "n_bad", int(bad.sum()))
        if bad.any():
            r, c = bad.nonzero(as_tuple=True)
            print("  rows mod 256 hist (64-row groups):", torch.bincount((r % 256) // 64, minlength=4).tolist(),
                  " cols (32-col groups):", torch.bincount(c // 32, minlength=8).tolist(),
                  " row tiles:", torch.unique(r // 256)[:10].tolist(), "n_tiles", int(torch.unique(r // 256).numel()))
            print("  sample:", [(int(r[i]), int(c[i]), float(Y[r[i], c[i]]), float(ref[r[i], c[i]])) for i in range(0, min(len(r), 5))])
