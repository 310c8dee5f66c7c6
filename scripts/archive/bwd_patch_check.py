#!/usr/bin/env python3
"""BatchNorm backward on the gather (ddmp_spmm_bnbwd_f32) on the 1M-face torus, both graphs, C = 256 / 512: us per call and a checksum
of the output BITS.  Run once per selection (DDMP_SPMM_PATCH_FORMS=15: lean kernel | unset: LDS-patch kernel): same bits expected."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dual_dmp_amd import ops, synth, _lib
from dual_dmp_amd.mesh import Mesh
dev = torch.device("cuda:0")
flip = int(os.environ.get("FLIP", "0"))
v, f = synth.torus(1000, 500)
if flip:
    f = synth.flip_edges(v, f, rounds=flip, seed=1)
    f = synth.add_hub(v, f, 1000, 24)
order = ops.rcb_order_host(v, 64).astype(np.int64); inv = np.empty_like(order); inv[order] = np.arange(len(order))
v, f = v[order], inv[f]
f = f[ops.rcb_order_host(v[f].mean(1), 64)]
m = Mesh(vs=v, faces=f)
e = torch.tensor(m.edges.T, dtype=torch.long)
graphs = {"vert": (torch.cat([e, e[[1, 0]]], 1), len(v)), "face": (torch.from_numpy(m.f_edges), len(f))}
tag = os.environ.get("DDMP_SPMM_PATCH_FORMS", "default")
for name, (ei, n) in graphs.items():
    eig = ei.to(dev)
    g = ops.graph_for(eig, n)
    for C in (256, 512):
        torch.manual_seed(C)
        dz = torch.randn(n, C, device=dev); yb = torch.randn(n, C, device=dev) * 2 + 0.5
        bn4 = torch.rand(4, C, device=dev) + 0.5; c10 = torch.randn(2, C, device=dev) * 0.1
        out = torch.empty(n, C, device=dev)
        ops.spmm_bnbwd(g, dz, yb, bn4, c10, out); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.spmm_bnbwd(g, dz, yb, bn4, c10, out)
        e1.record(); torch.cuda.synchronize()
        vv = out.view(torch.int32).to(torch.int64).flatten()
        w = (torch.arange(vv.numel(), device=dev, dtype=torch.int64) % 1021) + 1
        sel = _lib.lib().ddmp_spmm_patch_selected(g._h, C, 0, 1, 3)
        print("forms=%-8s flip=%d %s C=%d  patch=%d  %6.0f us  %016x" % (tag, flip, name, C, sel, e0.elapsed_time(e1) / 5 * 1e3,
                                                                     int((vv * w).sum().item()) & 0xFFFFFFFFFFFFFFFF), flush=True)
        del dz, yb, out
