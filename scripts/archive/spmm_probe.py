#!/usr/bin/env python3
"""SpMM structure probes: identity graph (self loop only: the kernel as a plain copy), chain graph (2 neighbours, perfectly
local), and the mesh graphs -- to separate the kernel's structural ceiling from the cost of the gathers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dual_dmp_amd import ops
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000


def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


i = torch.arange(n - 1)
graphs = {"identity": torch.zeros((2, 0), dtype=torch.long),
          "chain(2 nbrs)": torch.stack([torch.cat([i, i + 1]), torch.cat([i + 1, i])]),
          "ring+3 (4 nbrs)": torch.stack([torch.cat([i, i + 1, i[:-2], i[:-2] + 3]), torch.cat([i + 1, i, i[:-2] + 3, i[:-2]])])}
for name, ei in graphs.items():
    g = ops.graph_for(ei.to(dev), n)
    for C in (512, 256, 128):
        X = torch.randn(n, C, device=dev); Y = torch.empty(n, C, device=dev)
        us = timeit(lambda: ops.spmm(g, X, out=Y))
        alg = 2.0 * n * C * 4 + 4.0 * g.nnz + 8.0 * n
        print("%-16s N=%d C=%3d nnz/row=%.1f  %8.0f us  %7.1f GB/s alg" % (name, n, C, g.nnz / n, us, alg / us / 1e3))
X = torch.randn(n, 512, device=dev); Y = torch.empty_like(X)
us = timeit(lambda: Y.copy_(X))
print("torch copy N=%d C=512 %8.0f us %7.1f GB/s" % (n, us, 2.0 * n * 512 * 4 / us / 1e3))
