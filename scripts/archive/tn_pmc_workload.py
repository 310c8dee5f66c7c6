"""Workload for counter passes on the wide f16x3 wgrad: 1M x 512 x 512, BatchNorm backward on the G load (3 launches) and plain
(3 launches), caller scale slots (no pre-pass)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dual_dmp_amd import ops          # noqa: E402

dev = torch.device("cuda:0")
n, M, K = 1000000, 512, 512
dz = torch.randn(n, M, device=dev); yb = torch.randn(n, M, device=dev); z = torch.randn(n, K, device=dev)
bn4 = torch.rand(4, M, device=dev) + 0.5; c10 = torch.rand(2, M, device=dev) * 0.1
dW = torch.empty(M, K, device=dev)
slots = torch.zeros(2, 4, device=dev)
which = os.environ.get("TN_WHICH", "bnbwd")
call = (lambda: ops.gemm_tn_bnbwd(dz, yb, z, bn4, c10, out=dW)) if which == "bnbwd" else (lambda: ops.gemm_tn(dz, z, out=dW))
ops.gemm_next_scales(slots[0], slots[1], prime=True); call(); ops.gemm_scales_roll(slots)
for _ in range(3):
    ops.gemm_next_scales(slots[0], slots[1]); call()
torch.cuda.synchronize()
