#!/usr/bin/env python3
"""MAD after a full (short) training run: HIP path vs the CPU oracle in float32 and float64 from identical initial
weights, on small synthetic meshes.  Adam makes the trajectories chaotic (they separate ~10x per iteration), so the
comparison is between END RESULTS: the oracle's own float32-vs-float64 gap is the yardstick."""
import importlib.util, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import types
import numpy as np, torch
torch.set_num_threads(8)
spec = importlib.util.spec_from_file_location("ddmp_oracle", os.path.join(ROOT, "oracle", "ddmp_oracle.py"))
oracle = importlib.util.module_from_spec(spec); sys.modules["ddmp_oracle"] = oracle; spec.loader.exec_module(oracle)
from dual_dmp_amd import synth
from dual_dmp_amd.datamaker import dataset_from_meshes
from dual_dmp_amd.networks import PosNet, NormalNet
from dual_dmp_amd.trainer import FusedTrainer
from dual_dmp_amd.loss import mad
from dual_dmp_amd.mesh import Mesh

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 150
dev = torch.device("cuda:0")


def mad_of(pos, noisy, gt):
    o = Mesh.__new__(Mesh)
    o.vs, o.faces = np.asarray(pos, dtype=np.float64), noisy.faces
    Mesh.compute_face_normals(o)
    return float(mad(o.fn, gt.fn))


for name, (v, f) in (("icosphere-3", synth.icosphere(3)), ("cube-cad-6", synth.cube_cad(6))):
    gt, noisy, smooth = synth.make_triplet(v, f)
    data = dataset_from_meshes(noisy, smooth)
    torch.manual_seed(7)
    sd_p, sd_n = oracle.PosNetRef().state_dict(), oracle.NormalNetRef().state_dict()
    res = {}
    for dt in (torch.float32, torch.float64):
        rp, rn = oracle.PosNetRef(), oracle.NormalNetRef()
        rp.load_state_dict(sd_p); rn.load_state_dict(sd_n)
        odata, omesh = oracle.OracleDataset(noisy, smooth), noisy
        if dt == torch.float64:
            rp.double(); rn.double()
            for k in ("z1", "z2", "x_pos"):
                setattr(odata, k, getattr(odata, k).double())
            omesh = types.SimpleNamespace(vs=noisy.vs, fn=noisy.fn, faces=noisy.faces, f2f=noisy.f2f,
                                          v2v_mat=noisy.v2v_mat.double(), v_dims=noisy.v_dims.double())
        args = oracle.StepArgs()
        op = torch.optim.Adam(rp.parameters(), lr=args.pos_lr); on = torch.optim.Adam(rn.parameters(), lr=args.norm_lr)
        for ep in range(1, iters + 1):
            loss, p, n, _ = oracle.train_step(rp, rn, op, on, odata, omesh, args, ep)
        res[str(dt)] = (loss, mad_of(p.detach().double().numpy(), noisy, gt))
    posnet, normnet = PosNet(dev), NormalNet(dev)
    posnet.load_state_dict(sd_p); normnet.load_state_dict(sd_n)
    data.to(dev)
    tr = FusedTrainer(posnet, normnet, data, noisy)
    for ep in range(iters):
        loss = tr.step().item()
    res["hip"] = (loss, mad_of(tr.pos.cpu().numpy(), noisy, gt))
    print("%-12s V=%d F=%d  noisy MAD %.3f deg;  after %d iterations:" % (name, len(v), len(f), float(mad(noisy.fn, gt.fn)), iters))
    for k, (l, m) in res.items():
        print("    %-14s loss %.6f   MAD %.4f deg" % (k, l, m))
