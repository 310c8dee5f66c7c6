#!/usr/bin/env python3
"""Accuracy (vs float64 on a row sample) and time of the GEMM arithmetic modes 6 (bf16x6), 13 (f16x3), 14 (f16x4), 3.
usage: f16_probe.py [--rows N] [--scale S]   (S multiplies the operands: range check of the f16 scaling)"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dual_dmp_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1000000)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--modes", type=str, default="6,13,14,3")
a = ap.parse_args()
dev = torch.device("cuda:0")
n = a.rows
torch.manual_seed(0)


def timeit(fn, iters=a.iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def rel(x, ref):
    return float((x.double() - ref).norm() / ref.norm())


for K, M in ((512, 512), (256, 256), (256, 512), (512, 256)):
    A = torch.randn(n, K, device=dev) * a.scale
    A[:, : K // 4] *= 1e-3                                        # columns of very different magnitude
    W = torch.randn(M, K, device=dev) / K ** 0.5
    G = torch.randn(n, M, device=dev) * (a.scale * 1e-4) * torch.rand(n, 1, device=dev) ** 4      # gradient-like rows
    Yb = torch.randn(n, M, device=dev)
    sc = torch.rand(K, device=dev) + 0.5; sh = torch.randn(K, device=dev)
    bn4 = torch.stack([torch.rand(M, device=dev) + 0.5, torch.randn(M, device=dev) * 0.1, torch.zeros(M, device=dev), torch.zeros(M, device=dev)]).contiguous()
    c10 = torch.stack([torch.randn(M, device=dev) * 1e-6 * a.scale, torch.randn(M, device=dev) * 1e-6 * a.scale]).contiguous()
    Y = torch.empty(n, M, device=dev); X = torch.empty(n, K, device=dev); dW = torch.empty(M, K, device=dev)
    S = slice(0, 4096)
    Ad, Wd, Gd = A[S].double(), W.double(), G[S].double()
    ref_nt = Ad @ Wd.T
    ref_ntp = torch.nn.functional.leaky_relu(Ad * sc.double() + sh.double(), 0.01) @ Wd.T
    ref_nn = Gd @ Wd
    ref_tn = G.double().T @ A.double()
    fl = 2.0 * n * K * M
    fused_ok = ops.gemm_bnbwd_supported(M, K, n)
    for mode in [int(m) for m in a.modes.split(",")]:
        ops.set_gemm_mode(mode)
        out = []
        for name, fn, chk in (("nt", lambda: ops.gemm_nt(A, W, out=Y), lambda: rel(Y[S], ref_nt)),
                              ("nt+pro", lambda: ops.gemm_nt(A, W, out=Y, pro=(sc, sh)), lambda: rel(Y[S], ref_ntp)),
                              ("nn", lambda: ops.gemm_nn(G, W, out=X), lambda: rel(X[S], ref_nn)),
                              ("tn", lambda: ops.gemm_tn(G, A, out=dW), lambda: rel(dW, ref_tn))):
            us = timeit(fn)
            out.append("%s %6.0f us %5.0f TF err %.2e" % (name, us, fl / us / 1e6, chk()))
        if fused_ok:
            us = timeit(lambda: ops.gemm_nn_bnbwd(G, Yb, W, bn4, c10, out=X))
            us2 = timeit(lambda: ops.gemm_tn_bnbwd(G, Yb, A, bn4, c10, out=dW))
            out.append("nn_bnbwd %6.0f us  tn_bnbwd %6.0f us" % (us, us2))
        print("K=%3d M=%3d mode %2d | " % (K, M, mode) + " | ".join(out), flush=True)
    ops.set_gemm_mode(6)
