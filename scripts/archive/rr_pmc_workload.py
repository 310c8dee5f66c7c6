"""Workload for counter passes on the wide f16x3 forward GEMM: NT 1M x 512 x 512 and 1M x 256 x 256 with caller scale slots
(no pre-pass), 3 launches each.  DDMP_GEMM_RR=0|1 selects the row-panel / row-register kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dual_dmp_amd import ops          # noqa: E402

dev = torch.device("cuda:0")
n = 1000000
ops.set_gemm_mode(13)
for K, M in ((512, 512), (256, 256)):
    A = torch.randn(n, K, device=dev)
    W = torch.randn(M, K, device=dev) / K ** 0.5
    Y = torch.empty(n, M, device=dev)
    sc, sh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev)
    slots = torch.zeros(1, 4, device=dev)
    ops.gemm_next_scales(slots[0], None, prime=True)
    ops.gemm_nt(A, W, out=Y, pro=(sc, sh))
    ops.gemm_scales_roll(slots)
    for _ in range(3):
        ops.gemm_next_scales(slots[0], None)
        ops.gemm_nt(A, W, out=Y, pro=(sc, sh))
    torch.cuda.synchronize()
