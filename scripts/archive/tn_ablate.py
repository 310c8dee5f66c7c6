#!/usr/bin/env python3
"""Time the wide f16x3 wgrad (caller scale slots, no pre-pass) for ablation builds: DDMP_LIB=build_abl/libddmp_tN.so."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dual_dmp_amd import ops
dev = torch.device("cuda:0")
n = int(os.environ.get("ROWS", "1000000"))
out = []
shapes = [tuple(int(x) for x in t.split("x")) for t in os.environ.get("TN_SHAPES", "512x512 256x256 512x256").split()]
for M, K in shapes:
    dz = torch.randn(n, M, device=dev); yb = torch.randn(n, M, device=dev); z = torch.randn(n, K, device=dev)
    bn4 = torch.rand(4, M, device=dev) + 0.5; c10 = torch.rand(2, M, device=dev) * 0.1
    sc = torch.rand(K, device=dev) + 0.5; sh = torch.randn(K, device=dev)
    dW = torch.empty(M, K, device=dev)
    slots = torch.zeros(2, 4, device=dev)
    for name, call in (("bnbwd", lambda: ops.gemm_tn_bnbwd(dz, yb, z, bn4, c10, out=dW)),
                       ("pro", lambda: ops.gemm_tn(dz, z, out=dW, pro=(sc, sh))),
                       ("plain", lambda: ops.gemm_tn(dz, z, out=dW))):
        slots.zero_()
        ops.gemm_next_scales(slots[0], slots[1], prime=True); call(); ops.gemm_scales_roll(slots)
        def fn():
            ops.gemm_next_scales(slots[0], slots[1]); call()
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record(); torch.cuda.synchronize()
        out.append("%dx%d %s %5.0f" % (M, K, name, e0.elapsed_time(e1) / 5 * 1e3))
print("%-18s" % os.environ.get("DDMP_LIB", "default").split("/")[-1], " | ".join(out), flush=True)
