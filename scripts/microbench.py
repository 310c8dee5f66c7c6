#!/usr/bin/env python3
"""Kernel micro-benchmarks at bench scale (HIP-event timing); also the target of rocprofv3 --pmc runs.
usage: microbench.py [gemm|spmm|bn|all] [--rows N] [--iters K]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dual_dmp_amd import ops, synth
from dual_dmp_amd.mesh import Mesh

ap = argparse.ArgumentParser()
ap.add_argument("what", nargs="?", default="all")
ap.add_argument("--rows", type=int, default=1000000)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--order", default="native")
ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
ap.add_argument("--flip", type=int, default=0, help="rounds of random edge flips (irregular valence)")
ap.add_argument("--widths", default="512,256,128,64,32")
ap.add_argument("--rotate", type=int, default=1, help="spmm: cycle through this many (input, output) buffer sets so that narrow "
                "widths are not served from the 256 MB MALL (a 1M x 32 float tensor is 128 MB)")
a = ap.parse_args()
dev = torch.device("cuda:0")
n = a.rows
DT = torch.bfloat16 if a.dtype == "bf16" else torch.float32
ES = 2 if a.dtype == "bf16" else 4


def timeit(fn, iters=a.iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3     # us


if a.what in ("gemm", "all"):
    for K, M in ((512, 512), (256, 256), (256, 512), (512, 256), (128, 256)):
        A = torch.randn(n, K, device=dev); W = torch.randn(M, K, device=dev) / K ** 0.5
        G = torch.randn(n, M, device=dev)
        sc = torch.rand(K, device=dev) + 0.5; sh = torch.randn(K, device=dev)
        Y = torch.empty(n, M, device=dev); X = torch.empty(n, K, device=dev); dW = torch.empty(M, K, device=dev)
        fl = 2.0 * n * K * M
        for name, fn in (("nt", lambda: ops.gemm_nt(A, W, out=Y)), ("nt+pro", lambda: ops.gemm_nt(A, W, out=Y, pro=(sc, sh))),
                         ("nn", lambda: ops.gemm_nn(G, W, out=X)), ("tn", lambda: ops.gemm_tn(G, A, out=dW)),
                         ("tn+pro", lambda: ops.gemm_tn(G, A, out=dW, pro=(sc, sh)))):
            us = timeit(fn)
            print("gemm_%-7s K=%3d M=%3d  %8.0f us  %6.1f TF-eq" % (name, K, M, us, fl / us / 1e6))

if a.what in ("spmm", "all"):
    nu = int(round((n / 2.0) ** 0.5)) if a.what == "spmm_v" else None
    # face graph of a torus with n faces (deg 3+1) and vertex graph with n/2 verts (deg 6+1)
    nv_ = int(round((n / 4.0) ** 0.5)); nu_ = n // (2 * nv_)
    v, f = synth.torus(nu_, nv_)
    if a.flip:
        f = synth.flip_edges(v, f, rounds=a.flip, seed=1)
        f = synth.add_hub(v, f, 1234, 24)
    if a.order == "rcb":
        v, f = synth.rcb_relabel(v, f)
    elif a.order == "random":
        v, f = synth.permute_vertices(v, f, 0); f = synth.permute_faces(f, 0)
    elif a.order == "morton":
        v, f = synth.morton_relabel(v, f)
    m = Mesh(vs=v, faces=f)
    e = torch.tensor(m.edges.T, dtype=torch.long); ei = torch.cat([e, e[[1, 0]]], 1).to(dev)
    fi = torch.from_numpy(m.f_edges).to(dev)
    for gname, idx, nn_ in (("face", fi, len(f)), ("vert", ei, len(v))):
        g = ops.graph_for(idx, nn_)
        for C in [int(c) for c in a.widths.split(",")]:
            if a.rotate > 1:                                     # cold-cache figures: every launch works on another buffer set
                R = a.rotate
                Xs = [torch.randn(nn_, C, device=dev).to(DT) for _ in range(R)]
                Ys = [torch.empty(nn_, C, device=dev, dtype=DT) for _ in range(R)]
                Yps = [torch.randn(nn_, C, device=dev).to(DT) for _ in range(R)]
                sc = torch.rand(C, device=dev) + 0.5; sh = torch.randn(C, device=dev)
                bn4 = torch.rand(4, C, device=dev) + 0.5; c10 = torch.rand(2, C, device=dev) * 0.1
                sums = torch.empty(2 * C, dtype=torch.float64, device=dev); ref0 = torch.zeros(C, device=dev)
                k = [0]

                def rot(fn):
                    def go():
                        i = k[0] % R; k[0] += 1
                        fn(Xs[i], Ys[i], Yps[i])
                    return go
                it = max(a.iters, 2 * R)
                t = {name: timeit(rot(fn), it) for name, fn in (
                    ("plain", lambda X, Y, Yp: ops.spmm(g, X, out=Y)),
                    ("prologue", lambda X, Y, Yp: ops.spmm(g, X, out=Y, pro=(sc, sh))),
                    ("statistics", lambda X, Y, Yp: ops.spmm_stats(g, X, Y, ref0, sums, bias=sh)),
                    ("reduction", lambda X, Y, Yp: ops.spmm_bnred(g, X, Y, Yp, bn4, sums)),
                    ("bn-backward", lambda X, Y, Yp: ops.spmm_bnbwd(g, X, Yp, bn4, c10, Y)))}
                b2, b3 = 2.0 * nn_ * C * ES, 3.0 * nn_ * C * ES
                print("spmm %s N=%d C=%3d  cold (%d buffer sets): " % (gname, nn_, C, R) + "   ".join(
                    "%s %5.0f us (%.2f TB/s)" % (nm, us, (b3 if nm in ("reduction", "bn-backward") else b2) / us / 1e6) for nm, us in t.items()), flush=True)
                del Xs, Ys, Yps
                continue
            X = torch.randn(nn_, C, device=dev).to(DT); Y = torch.empty(nn_, C, device=dev, dtype=DT)
            us = timeit(lambda: ops.spmm(g, X, out=Y))
            alg = 2.0 * nn_ * C * ES + 4.0 * g.nnz + 8.0 * nn_
            print("spmm %s N=%d C=%3d  %8.0f us  %7.1f GB/s alg (%.1f%% of 8 TB/s)  gather-logical %.1f GB/s" % (
                gname, nn_, C, us, alg / us / 1e3, alg / us / 1e3 / 80.0, (g.nnz * C * 4.0 + nn_ * C * 4.0) / us / 1e3))
            if C >= 32:
                sc = torch.rand(C, device=dev) + 0.5; sh = torch.randn(C, device=dev)
                us_p = timeit(lambda: ops.spmm(g, X, out=Y, pro=(sc, sh)))
                sums0 = torch.empty(2 * C, dtype=torch.float64, device=dev); ref0 = torch.zeros(C, device=dev)
                us_t = timeit(lambda: ops.spmm_stats(g, X, Y, ref0, sums0, bias=sh))
                bn4 = torch.rand(4, C, device=dev) + 0.5; sums = torch.empty(2 * C, dtype=torch.float64, device=dev)
                Yp = torch.randn(nn_, C, device=dev).to(DT)
                us_r = timeit(lambda: ops.spmm_bnred(g, X, Y, Yp, bn4, sums))
                c10 = torch.rand(2, C, device=dev) * 0.1
                us_b = timeit(lambda: ops.spmm_bnbwd(g, X, Yp, bn4, c10, Y))
                dY = torch.empty_like(X)
                us_a = timeit(lambda: ops.bn_bwd_apply(X, Yp, bn4, c10, dY, sums))
                us_s = timeit(lambda: ops.bn_bwd_reduce(X, Yp, bn4, sums2=sums))
                print("     +statistics %8.0f us (x%.2f)" % (us_t, us_t / us))
                print("     +prologue %8.0f us (x%.2f)   +bn-backward reduce %8.0f us (x%.2f; separate pass %.0f us)   "
                      "bn-backward on the gather %8.0f us (apply pass %.0f us + plain)" % (us_p, us_p / us, us_r, us_r / us, us_s, us_b, us_a))

if a.what in ("bn", "all"):
    for C in (512, 256):
        Y = torch.randn(n, C, device=dev); dZ = torch.randn(n, C, device=dev); dY = torch.empty(n, C, device=dev)
        sums = torch.empty(2 * C, dtype=torch.float64, device=dev)
        bn4 = torch.rand(4, C, device=dev) + 0.5; c10 = torch.rand(2, C, device=dev)
        for name, fn, by in (("stats", lambda: ops.bn_stats(Y, sums=sums), 4.0), ("bwd_reduce", lambda: ops.bn_bwd_reduce(dZ, Y, bn4, sums2=sums), 8.0),
                             ("bwd_apply", lambda: ops.bn_bwd_apply(dZ, Y, bn4, c10, dY, sums), 12.0)):
            us = timeit(fn)
            print("bn_%-10s C=%3d %8.0f us  %7.1f GB/s" % (name, C, us, by * n * C / us / 1e3))
