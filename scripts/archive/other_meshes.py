#!/usr/bin/env python3
"""The training iteration on other synthetic meshes than the bench torus: icosphere (12 valence-5 vertices), an open
grid (boundary: faces with -1 neighbours, valence 2..6) and a CAD-like cube -- time per iteration and MAD progress."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dual_dmp_amd import synth
from dual_dmp_amd.datamaker import dataset_from_meshes
from dual_dmp_amd.networks import PosNet, NormalNet
from dual_dmp_amd.trainer import FusedTrainer
from dual_dmp_amd.loss import mad
from dual_dmp_amd.mesh import Mesh
dev = torch.device("cuda:0")


def mad_of(pos, noisy, gt):
    o = Mesh.__new__(Mesh); o.vs, o.faces = np.asarray(pos, dtype=np.float64), noisy.faces
    Mesh.compute_face_normals(o); return float(mad(o.fn, gt.fn))


for name, (v, f) in (("icosphere-8", synth.icosphere(8)), ("open grid 600x500", synth.open_grid(600, 500)),
                     ("cube-cad-160", synth.cube_cad(160))):
    v, f = synth.permute_vertices(v, f, 1)                     # arbitrary numbering: the engine relabels internally
    gt, noisy, smooth = synth.make_triplet(v, f)
    data = dataset_from_meshes(noisy, smooth); data.to(dev)
    torch.manual_seed(0)
    tr = FusedTrainer(PosNet(dev), NormalNet(dev), data, noisy, use_graph=True, overlap=True)
    for _ in range(5):
        tr.step().item()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        loss = tr.step().item()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("%-18s V=%7d F=%7d  %7.2f ms/iteration  loss %.4f  MAD %.3f -> %.3f deg" % (
        name, len(v), len(f), dt * 1e3, loss, float(mad(noisy.fn, gt.fn)), mad_of(tr.pos.cpu().numpy(), noisy, gt)))
