#!/usr/bin/env python3
"""Time the panel NT GEMM in mode 13 with caller scale slots (no pre-pass), for ablation builds (DDMP_LIB=...)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dual_dmp_amd import ops
dev = torch.device("cuda:0")
n = 1000000
mode = int(os.environ.get("MODE", "13"))
ops.set_gemm_mode(mode)
out = []
for K, M in ((512, 512), (256, 256), (256, 512)):
    A = torch.randn(n, K, device=dev); W = torch.randn(M, K, device=dev) / K ** 0.5
    Y = torch.empty(n, M, device=dev)
    slots = torch.zeros(1, 4, device=dev)
    if mode == 13:
        ops.gemm_next_scales(slots[0], None, prime=True)
    ops.gemm_nt(A, W, out=Y)
    if mode == 13:
        ops.gemm_scales_roll(slots)
    def fn():
        if mode == 13:
            ops.gemm_next_scales(slots[0], None)
        ops.gemm_nt(A, W, out=Y)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record(); torch.cuda.synchronize()
    out.append("%dx%d %6.0f us" % (K, M, e0.elapsed_time(e1) / 5 * 1e3))
print(os.environ.get("DDMP_LIB", "default").split("/")[-1], "mode", mode, " | ".join(out), flush=True)
