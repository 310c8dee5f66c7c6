import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from dual_dmp_amd import ops, synth
from dual_dmp_amd.datamaker import dataset_from_meshes
from dual_dmp_amd.networks import PosNet, NormalNet
from dual_dmp_amd.trainer import FusedTrainer
from dual_dmp_amd.loss import mad
from dual_dmp_amd.mesh import Mesh
dev = torch.device("cuda:0")
v, f = synth.torus(380, 190)
gt, noisy, smooth = synth.make_triplet(v, f)
def mad_of(pos):
    o = Mesh.__new__(Mesh); o.vs, o.faces = np.asarray(pos, dtype=np.float64), noisy.faces
    Mesh.compute_face_normals(o); return float(mad(o.fn, gt.fn))
modes = [int(m) for m in sys.argv[1].split(",")] if len(sys.argv) > 1 else [6, 0]
seeds = [int(m) for m in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1]
import itertools
for seed, mode in itertools.product(seeds, modes):
    ops.set_gemm_mode(mode)
    data = dataset_from_meshes(noisy, smooth); data.to(dev)
    torch.manual_seed(seed)
    tr = FusedTrainer(PosNet(dev), NormalNet(dev), data, noisy, use_graph=(mode != 0), overlap=(mode != 0))
    ls = []
    for it in range(150):
        ls.append(tr.step().item())
        if it % 10 == 9:
            tr.check_scales()                                   # f16x3: no operand outgrew its scale (raises)
    print("seed %d mode %2d (fused routes: %s): loss @1 %.5f @10 %.5f @50 %.5f @100 %.5f @150 %.5f  MAD %.4f" % (seed, mode, any(tr.neng.fuse_bnbwd), ls[0], ls[9], ls[49], ls[99], ls[149], mad_of(tr.pos.cpu().numpy())))
