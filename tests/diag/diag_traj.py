"""Diagnostic: how far do free-running trajectories drift?  HIP-f32 vs oracle-f32 vs oracle-f64."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import __graft_entry__ as ge
oracle = ge._load_oracle()
from dual_dmp_amd import synth
from dual_dmp_amd.datamaker import dataset_from_meshes
from dual_dmp_amd.networks import PosNet, NormalNet
from dual_dmp_amd.trainer import FusedTrainer

dev = torch.device("cuda:0")
v, f = synth.icosphere(3)
gt, noisy, smooth = synth.make_triplet(v, f)
data = dataset_from_meshes(noisy, smooth)
torch.manual_seed(11)
sd_pos, sd_norm = oracle.PosNetRef().state_dict(), oracle.NormalNetRef().state_dict()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10

def run_oracle(dtype):
    odata = oracle.OracleDataset(noisy, smooth)
    pn, nn_ = oracle.PosNetRef(), oracle.NormalNetRef()
    pn.load_state_dict(sd_pos); nn_.load_state_dict(sd_norm)
    if dtype == torch.float64:
        pn.double(); nn_.double()
        for k in ("z1", "z2", "x_pos"):
            setattr(odata, k, getattr(odata, k).double())
    mesh = noisy
    if dtype == torch.float64:
        import types
        mesh = types.SimpleNamespace(vs=noisy.vs, fn=noisy.fn, faces=noisy.faces, f2f=noisy.f2f,
                                     v2v_mat=noisy.v2v_mat.double(), v_dims=noisy.v_dims.double())
    args = oracle.StepArgs()
    op = torch.optim.Adam(pn.parameters(), lr=0.01); on = torch.optim.Adam(nn_.parameters(), lr=0.01)
    out = []
    for ep in range(1, steps + 1):
        out.append(oracle.train_step(pn, nn_, op, on, odata, mesh, args, ep))
    return out

h32 = run_oracle(torch.float32)
h64 = run_oracle(torch.float64)
posnet, normnet = PosNet(dev), NormalNet(dev)
posnet.load_state_dict(sd_pos); normnet.load_state_dict(sd_norm)
tr = FusedTrainer(posnet, normnet, data, noisy)
print("step | loss64        | dloss hip/o32 (rel to o64) | max|dpos| hip-o64  o32-o64  hip-o32 | max|dnorm| hip-o64 o32-o64")
for s in range(steps):
    loss = tr.step().item()
    p, n = tr.pos.cpu().double(), tr.norm.cpu().double()
    l64, p64, n64, _ = h64[s]; l32, p32, n32, _ = h32[s]
    print("%3d  | %.8f | %.2e %.2e | %.2e %.2e %.2e | %.2e %.2e" % (
        s + 1, l64, abs(loss - l64) / l64, abs(l32 - l64) / l64,
        (p - p64.double()).abs().max(), (p32.double() - p64.double()).abs().max(), (p - p32.double()).abs().max(),
        (n - n64.double()).abs().max(), (n32.double() - n64.double()).abs().max()))
fn = lambda P: oracle.face_normals_np(P.numpy().astype(np.float64), noisy.faces)[0]
print("MAD deg: hip %.5f  o32 %.5f  o64 %.5f  (noisy input %.5f)" % (
    oracle.mad_np(fn(tr.pos.cpu()), gt.fn), oracle.mad_np(fn(h32[-1][1]), gt.fn), oracle.mad_np(fn(h64[-1][1].float()), gt.fn),
    oracle.mad_np(noisy.fn, gt.fn)))
