"""Free-running MAD / loss statistics over several seeds: HIP path vs the float32 CPU oracle (test infrastructure)."""
import importlib.util, os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
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
dev=torch.device("cuda:0"); iters=120
def mad_of(pos, noisy, gt):
    o = Mesh.__new__(Mesh); o.vs, o.faces = np.asarray(pos, dtype=np.float64), noisy.faces
    Mesh.compute_face_normals(o); return float(mad(o.fn, gt.fn))
v,f = synth.cube_cad(6)
gt, noisy, smooth = synth.make_triplet(v, f)
data = dataset_from_meshes(noisy, smooth)
rows=[]
for seed in range(5):
    torch.manual_seed(100+seed)
    sd_p, sd_n = oracle.PosNetRef().state_dict(), oracle.NormalNetRef().state_dict()
    rp, rn = oracle.PosNetRef(), oracle.NormalNetRef(); rp.load_state_dict(sd_p); rn.load_state_dict(sd_n)
    odata = oracle.OracleDataset(noisy, smooth); args = oracle.StepArgs()
    op = torch.optim.Adam(rp.parameters(), lr=args.pos_lr); on = torch.optim.Adam(rn.parameters(), lr=args.norm_lr)
    for ep in range(1, iters+1): lo, p, n, _ = oracle.train_step(rp, rn, op, on, odata, noisy, args, ep)
    mo = mad_of(p.detach().double().numpy(), noisy, gt)
    posnet, normnet = PosNet(dev), NormalNet(dev); posnet.load_state_dict(sd_p); normnet.load_state_dict(sd_n)
    d2 = dataset_from_meshes(noisy, smooth); d2.to(dev)
    tr = FusedTrainer(posnet, normnet, d2, noisy)
    for ep in range(iters): lh = tr.step().item()
    mh = mad_of(tr.pos.cpu().numpy(), noisy, gt)
    rows.append((lo, mo, lh, mh)); print("seed %d: oracle f32 loss %.4f MAD %.3f | hip loss %.4f MAD %.3f" % (seed, lo, mo, lh, mh), flush=True)
a=np.array(rows); print("mean: oracle loss %.4f MAD %.3f (std %.3f) | hip loss %.4f MAD %.3f (std %.3f)" % (a[:,0].mean(), a[:,1].mean(), a[:,1].std(), a[:,2].mean(), a[:,3].mean(), a[:,3].std()))
