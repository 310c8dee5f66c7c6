#!/usr/bin/env python3
"""MAD every 100 epochs of 1000 free-running HIP iterations with float32 and with bfloat16 features (BASELINE.json configs[1]'s arithmetic), same initial
weights: a CAD-like mesh (13,068 faces) and a non-CAD one (icosphere-5, 20,480 faces).  usage: mad_bf16_13k.py [iters=1000]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from dual_dmp_amd import synth
from dual_dmp_amd.datamaker import dataset_from_meshes
from dual_dmp_amd.networks import PosNet, NormalNet
from dual_dmp_amd.trainer import FusedTrainer
from dual_dmp_amd.loss import mad
from dual_dmp_amd.mesh import Mesh

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda:0")


def mad_of(pos, noisy, gt):
    o = Mesh.__new__(Mesh)
    o.vs, o.faces = np.asarray(pos, dtype=np.float64), noisy.faces
    Mesh.compute_face_normals(o)
    return float(mad(o.fn, gt.fn))


for name, (v, f) in (("cube-cad-33", synth.cube_cad(33)), ("icosphere-5", synth.icosphere(5))):
    gt, noisy, smooth = synth.make_triplet(v, f)
    out = {}
    for dt in (torch.float32, torch.bfloat16):
        data = dataset_from_meshes(noisy, smooth)
        torch.manual_seed(7)
        posnet, normnet = PosNet(dev, dtype=dt), NormalNet(dev, dtype=dt)
        data.to(dev)
        tr = FusedTrainer(posnet, normnet, data, noisy)
        t0 = time.perf_counter()
        for ep in range(1, iters + 1):
            loss = tr.step().item()
            if ep % 100 == 0:
                out.setdefault(ep, []).append((loss, mad_of(tr.pos.float().cpu().numpy(), noisy, gt)))
        out.setdefault("s", []).append(time.perf_counter() - t0)
    print("%s: V=%d F=%d, noisy input MAD %.4f deg; %d iterations: float32 %.1f s, bf16 features %.1f s" % (
        name, len(v), len(f), float(mad(noisy.fn, gt.fn)), iters, out["s"][0], out["s"][1]))
    print("  epoch   f32 loss  f32 MAD deg |  bf16 loss  bf16 MAD deg")
    for ep in sorted(k for k in out if k != "s"):
        (l0, m0), (l1, m1) = out[ep]
        print("  %5d  %9.5f  %9.4f   | %9.5f  %9.4f" % (ep, l0, m0, l1, m1))
