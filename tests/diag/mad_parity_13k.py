#!/usr/bin/env python3
"""MAD along a training run at the reference's own mesh size (13,068 faces: the size of BASELINE.json configs[0]) and iteration
counts: the HIP path for `hip_iters` iterations (main.py's default is 1000) and the CPU oracle in float32 for `oracle_iters`, from
identical initial weights, MAD every 100 epochs as main.py:117-127 evaluates it.  Free-running training is chaotic under Adam
(DESIGN.md 5): what can agree is the LEVEL the two runs settle at, not digits.
    usage: mad_parity_13k.py [oracle_iters=500] [hip_iters=1000] [threads=32]"""
import importlib.util, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
o_iters = int(sys.argv[1]) if len(sys.argv) > 1 else 500
h_iters = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
torch.set_num_threads(int(sys.argv[3]) if len(sys.argv) > 3 else 32)
spec = importlib.util.spec_from_file_location("ddmp_oracle", os.path.join(ROOT, "oracle", "ddmp_oracle.py"))
oracle = importlib.util.module_from_spec(spec); sys.modules["ddmp_oracle"] = oracle; spec.loader.exec_module(oracle)
from dual_dmp_amd import synth
from dual_dmp_amd.datamaker import dataset_from_meshes
from dual_dmp_amd.networks import PosNet, NormalNet
from dual_dmp_amd.trainer import FusedTrainer
from dual_dmp_amd.loss import mad
from dual_dmp_amd.mesh import Mesh

dev = torch.device("cuda:0")


def mad_of(pos, noisy, gt):
    o = Mesh.__new__(Mesh)
    o.vs, o.faces = np.asarray(pos, dtype=np.float64), noisy.faces
    Mesh.compute_face_normals(o)
    return float(mad(o.fn, gt.fn))


v, f = synth.cube_cad(33)
gt, noisy, smooth = synth.make_triplet(v, f)
data = dataset_from_meshes(noisy, smooth)
torch.manual_seed(7)
sd_p, sd_n = oracle.PosNetRef().state_dict(), oracle.NormalNetRef().state_dict()
print("cube-cad-33: V=%d F=%d, noisy input MAD %.4f deg, smoothed input %.4f deg" % (len(v), len(f), float(mad(noisy.fn, gt.fn)), float(mad(smooth.fn, gt.fn))), flush=True)

posnet, normnet = PosNet(dev), NormalNet(dev)
posnet.load_state_dict(sd_p); normnet.load_state_dict(sd_n)
data.to(dev)
tr = FusedTrainer(posnet, normnet, data, noisy)
t0 = time.perf_counter()
hip = {}
for ep in range(1, h_iters + 1):
    loss = tr.step().item()
    if ep % 100 == 0:
        hip[ep] = (loss, mad_of(tr.pos.cpu().numpy(), noisy, gt))
print("HIP: %d iterations in %.1f s" % (h_iters, time.perf_counter() - t0), flush=True)

rp, rn = oracle.PosNetRef(), oracle.NormalNetRef()
rp.load_state_dict(sd_p); rn.load_state_dict(sd_n)
odata = oracle.OracleDataset(noisy, smooth)
args = oracle.StepArgs()
op = torch.optim.Adam(rp.parameters(), lr=args.pos_lr); on = torch.optim.Adam(rn.parameters(), lr=args.norm_lr)
t0 = time.perf_counter()
orc = {}
for ep in range(1, o_iters + 1):
    loss, p, n, _ = oracle.train_step(rp, rn, op, on, odata, noisy, args, ep)
    if ep % 100 == 0:
        orc[ep] = (float(loss), mad_of(p.detach().double().numpy(), noisy, gt))
        print("  oracle epoch %d: %.1f s so far" % (ep, time.perf_counter() - t0), flush=True)
print("oracle float32 (%d threads): %d iterations in %.1f s" % (torch.get_num_threads(), o_iters, time.perf_counter() - t0))
print("epoch    HIP loss   HIP MAD deg | oracle loss  oracle MAD deg")
for ep in sorted(hip):
    o = orc.get(ep)
    print("%5d  %10.6f  %9.4f   | %s" % (ep, hip[ep][0], hip[ep][1], ("%10.6f  %9.4f" % o) if o else "        -          -"))
