"""Workload for counter passes on the gather: 3 launches each of the plain SpMM (C = 512, RCB order: the LDS-patch kernel) on the face graph
(1M rows, 4 entries per row) and on the vertex graph (0.5M rows, 7 entries per row) of the bench torus.
  rocprofv3 --pmc <counters> --kernel-trace -d gpurun_out/x -o r -- python3 scripts/spmm_pmc_workload.py [C]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dual_dmp_amd import ops, synth          # noqa: E402
from dual_dmp_amd.mesh import Mesh           # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
v, f = synth.torus(1000, 500)
v, f = synth.rcb_relabel(v, f)               # (the engines' numbering since round 5; Morton before)
m = Mesh(vs=v, faces=f)
e = torch.tensor(m.edges.T, dtype=torch.long)
ei = torch.cat([e, e[[1, 0]]], 1).to(dev)
fi = torch.from_numpy(m.f_edges).to(dev)
for idx, n in ((fi, len(f)), (ei, len(v))):
    g = ops.graph_for(idx, n)
    X = torch.randn(n, C, device=dev)
    Y = torch.empty(n, C, device=dev)
    for _ in range(3):
        ops.spmm(g, X, out=Y)
    torch.cuda.synchronize()
