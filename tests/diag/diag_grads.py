"""Diagnostic: per-parameter gradient error of the HIP engine vs the float64 oracle, with selectable
perturbations of the (initially trivial) conv biases / BN affine parameters."""
import sys, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import __graft_entry__ as ge
oracle = ge._load_oracle()
from dual_dmp_amd import synth
from dual_dmp_amd.datamaker import dataset_from_meshes
from dual_dmp_amd.networks import PosNet, NormalNet

dev = torch.device("cuda:0")
v, f = synth.icosphere(3)
gt, noisy, smooth = synth.make_triplet(v, f)
data = dataset_from_meshes(noisy, smooth)
odata = oracle.OracleDataset(noisy, smooth)
for k in ("z1", "z2", "x_pos"):
    setattr(odata, k, getattr(odata, k).double())
rel = lambda a, b: float((a.double().cpu() - b.double().cpu()).norm() / (b.double().cpu().norm() + 1e-30))
for mode in sys.argv[1:] or ["none", "convbias", "beta", "gamma", "all"]:
    torch.manual_seed(5)
    ref = oracle.PosNetRef()
    with torch.no_grad():
        for i in range(1, 13):
            if mode in ("convbias", "all"):
                getattr(ref, "conv%d" % i).bias.normal_(std=0.1)
            if mode in ("gamma", "all"):
                getattr(ref, "bn%d" % i).weight.uniform_(0.5, 1.5)
            if mode in ("beta", "all"):
                getattr(ref, "bn%d" % i).bias.normal_(std=0.1)
    net = PosNet(dev)
    net.load_state_dict(ref.state_dict())
    r64 = copy.deepcopy(ref).double()
    dout = torch.randn(len(noisy.vs), 3)
    o = r64(odata); o.backward(dout.double())
    oh = net(data); oh.backward(dout.to(dev))
    got = net.named_views(grads=True)
    errs = {n: rel(got[n], p.grad) for n, p in r64.named_parameters() if not (n.startswith("conv") and n.endswith(".bias"))}
    print("mode=%-8s out %.1e | " % (mode, rel(oh, o)) + " ".join("%s:%.0e" % (n.replace(".lin.weight", "W").replace(".weight", "g").replace(".bias", "b"), e) for n, e in errs.items()))

if "gamma" in (sys.argv[1:] or []):
    r32 = copy.deepcopy(ref)
    od32 = oracle.OracleDataset(noisy, smooth)
    o32 = r32(od32); o32.backward(dout)
    for name in ("bn6.bias", "bn6.weight", "bn7.bias"):
        g64 = dict(r64.named_parameters())[name].grad
        g32 = dict(r32.named_parameters())[name].grad.double()
        gh = got[name].double().cpu()
        eh, e32 = (gh - g64).abs(), (g32 - g64).abs()
        idx = torch.argsort(eh, descending=True)[:5]
        print(name, "|g64| max %.3e  L2 %.3e ; hip err max %.3e (L2 %.3e) ; o32 err max %.3e (L2 %.3e)" % (
            g64.abs().max(), g64.norm(), eh.max(), eh.norm(), e32.max(), e32.norm()))
        print("   worst channels", idx.tolist(), "g64", g64[idx].tolist(), "hip", gh[idx].tolist(), "o32", g32[idx].tolist())
        gam = dict(r64.named_parameters())["bn6.weight"].detach()
        print("   gamma at worst", gam[idx].tolist())
    eng = net._engine
    print("bn6 stats: rstd max %.3e min %.3e ; mean absmax %.3e" % (eng.bn4[5][3].max(), eng.bn4[5][3].min(), eng.bn4[5][2].abs().max()))
