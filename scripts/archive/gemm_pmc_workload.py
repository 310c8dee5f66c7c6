"""Workload for counter passes on the float32 row-panel GEMMs: NT / NN / TN at 1M rows, 512x512 and 256x256 (3 launches each).
  rocprofv3 --pmc <counters> --kernel-trace -d gpurun_out/x -o r -- python3 scripts/gemm_pmc_workload.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dual_dmp_amd import ops          # noqa: E402

dev = torch.device("cuda:0")
n = 1000000
for K, M in ((512, 512), (256, 256)):
    A = torch.randn(n, K, device=dev)
    W = torch.randn(M, K, device=dev) / K ** 0.5
    G = torch.randn(n, M, device=dev)
    Y = torch.empty(n, M, device=dev)
    X = torch.empty(n, K, device=dev)
    dW = torch.empty(M, K, device=dev)
    for _ in range(3):
        ops.gemm_nt(A, W, out=Y)
        ops.gemm_nn(G, W, out=X)
        ops.gemm_tn(G, A, out=dW)
    torch.cuda.synchronize()
