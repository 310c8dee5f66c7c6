#!/usr/bin/env python3
"""Can a compute-bound GEMM and the HBM-bound kernels share the GPU?  wgrad (TN panel) on one stream, SpMM + BatchNorm
reduction on another: concurrent wall time vs the sum of the two alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dual_dmp_amd import ops, synth
from dual_dmp_amd.mesh import Mesh
dev = torch.device("cuda:0")
n, C = 1000000, 512
v, f = synth.torus(1000, 500)
v, f = synth.morton_relabel(v, f)
m = Mesh(vs=v, faces=f)
fi = torch.from_numpy(m.f_edges).to(dev)
g = ops.graph_for(fi, len(f))
G = torch.randn(n, C, device=dev); Z = torch.randn(n, C, device=dev); dW = torch.empty(C, C, device=dev)
X = torch.randn(n, C, device=dev); Y = torch.empty(n, C, device=dev)
bn4 = torch.rand(4, C, device=dev) + 0.5; sums = torch.empty(2 * C, dtype=torch.float64, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def gemm():
    ops.gemm_tn(G, Z, out=dW)


def mem():
    ops.spmm(g, X, out=Y)
    ops.bn_bwd_reduce(Y, X, bn4, sums2=sums)


def wall(fns, iters=10):
    for st, fn in fns:
        with torch.cuda.stream(st):
            fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for st in (s1, s2):
        st.wait_stream(torch.cuda.current_stream())
    for _ in range(iters):
        for st, fn in fns:
            with torch.cuda.stream(st):
                fn()
    for st in (s1, s2):
        torch.cuda.current_stream().wait_stream(st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


a = wall([(s1, gemm)]); b = wall([(s2, mem)]); c = wall([(s1, gemm), (s2, mem)])
print("wgrad 512x512 alone %.0f us | spmm+bn_reduce alone %.0f us | concurrent %.0f us (sum %.0f, max %.0f)" % (a, b, c, a + b, max(a, b)))
