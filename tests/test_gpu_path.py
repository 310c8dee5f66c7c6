"""HIP path vs golden vectors (captured from the reference) and vs the CPU oracle:
losses (values, dtypes, gradients), GCNConv drop-in, PosNet / NormalNet (fused + modular),
and the training step trajectory + MAD.

Tolerances: single forward rel-L2 <= 1e-4 after 12 layers; gradients rel-L2 <= 1e-3;
loss scalars |d| <= 1e-5 rel; 10-step trajectory max-abs <= 1e-3 (unit mean edge); MAD |d| <= 1e-3 deg.
"""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
NAMES = ["ico2", "grid4", "cube3", "grid7x5"]


def relerr(a, b):
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _golden_mesh(golden_dir, name):
    gm = np.load(os.path.join(golden_dir, "mesh_%s.npz" % name))
    return types.SimpleNamespace(vs=gm["vs"], faces=gm["faces"], edges=gm["edges"], f2f=gm["f2f"], fn=gm["fn"])


# ------------------------------------------------------------------------------------ losses
@pytest.mark.parametrize("name", NAMES)
def test_losses_match_reference_golden(dev, golden_dir, name):
    from dual_dmp_amd import loss as L
    gl = np.load(os.path.join(golden_dir, "loss_%s.npz" % name))
    m = _golden_mesh(golden_dir, name)

    def leaf(a):
        return torch.from_numpy(a).to(dev).requires_grad_(True)

    pos, nrm = leaf(gl["pos"]), leaf(gl["norm"])
    l = L.pos_rec_loss(pos, m.vs)
    assert l.dtype == torch.float64
    np.testing.assert_allclose(l.item(), gl["pos_rec"], rtol=1e-9)
    (g,) = torch.autograd.grad(l, pos)
    assert g.dtype == torch.float32 and relerr(g, gl["pos_rec_dpos"]) < 1e-5

    l = L.mesh_laplacian_loss(pos, m)
    assert l.dtype == torch.float32
    np.testing.assert_allclose(l.item(), gl["lap"], rtol=1e-5)
    (g,) = torch.autograd.grad(l, pos)
    assert relerr(g, gl["lap_dpos"]) < 1e-5

    l = L.norm_rec_loss(nrm, m.fn)
    assert l.dtype == torch.float64
    np.testing.assert_allclose(l.item(), gl["norm_rec"], rtol=1e-9)
    (g,) = torch.autograd.grad(l, nrm)
    assert relerr(g, gl["norm_rec_dnorm"]) < 1e-6

    for loop in (1, 5):
        l, new_fn = L.fn_bnf_loss(pos, nrm, m, loop=loop)
        assert l.dtype == torch.float32
        np.testing.assert_allclose(l.item(), gl["bnf%d" % loop], rtol=2e-5)
        assert relerr(new_fn, gl["bnf%d_newfn" % loop]) < 1e-5
        (g,) = torch.autograd.grad(l, nrm)
        assert relerr(g, gl["bnf%d_dnorm" % loop]) < 1e-4, loop

    l = L.pos_norm_loss(pos, nrm, m)
    np.testing.assert_allclose(l.item(), gl["pos_norm"], rtol=1e-5)
    gp, gn = torch.autograd.grad(l, [pos, nrm])
    assert relerr(gp, gl["pos_norm_dpos"]) < 1e-5 and relerr(gn, gl["pos_norm_dnorm"]) < 1e-5

    # the weighted sum exactly as main.py:106 (float64 by promotion) and its gradients
    tot = (3.0 * L.pos_rec_loss(pos, m.vs) + 4.0 * L.mesh_laplacian_loss(pos, m) + 4.0 * L.norm_rec_loss(nrm, m.fn)
           + 4.0 * L.fn_bnf_loss(pos, nrm, m, loop=1)[0] + 1.0 * L.pos_norm_loss(pos, nrm, m))
    assert tot.dtype == torch.float64
    np.testing.assert_allclose(tot.item(), gl["total"], rtol=1e-5)
    gp, gn = torch.autograd.grad(tot, [pos, nrm])
    assert relerr(gp, gl["total_dpos"]) < 1e-4 and relerr(gn, gl["total_dnorm"]) < 1e-4

    # fused engine: same numbers without autograd
    eng = L.LossEngine(m, dev, bnfloop=1)
    buf, dpos, dnorm = eng.forward_backward(pos.detach(), nrm.detach(), gate4=1.0)
    buf = buf.cpu().numpy()
    np.testing.assert_allclose(buf[:6], [gl["pos_rec"], gl["lap"], gl["norm_rec"], gl["bnf1"], gl["pos_norm"],
                                         gl["total"]], rtol=2e-5)
    assert relerr(dpos, gl["total_dpos"]) < 1e-4 and relerr(dnorm, gl["total_dnorm"]) < 1e-4
    assert L.mad(nrm.detach(), m.fn) == pytest.approx(float(gl["mad"]), rel=1e-12)


def test_loss_ltype_conventions(dev, golden_dir):
    from dual_dmp_amd import loss as L
    m = _golden_mesh(golden_dir, "grid4")
    pos = torch.zeros(len(m.vs), 3, device=dev)
    with pytest.raises(NotImplementedError):
        L.pos_rec_loss(pos, m.vs, ltype="l1mae")
    with pytest.raises(SystemExit):                      # reference: print("[ERROR]: ltype error"); exit()
        L.pos_rec_loss(pos, m.vs, ltype="bogus")


# ------------------------------------------------------------------------------------ nets
def _case(dev, which="ico3"):
    from dual_dmp_amd import synth
    from dual_dmp_amd.datamaker import dataset_from_meshes
    v, f = synth.icosphere(3) if which == "ico3" else synth.open_grid(12, 9)
    v, f = synth.permute_vertices(v, f, 3)
    gt, noisy, smooth = synth.make_triplet(v, f)
    return gt, noisy, smooth, dataset_from_meshes(noisy, smooth)


@pytest.mark.parametrize("cin,cout", [(16, 32), (7, 32), (64, 32), (512, 256), (32, 3), (5, 6)])
def test_gcnconv_dropin_matches_oracle(dev, oracle, cin, cout):
    from dual_dmp_amd.nn_ops import GCNConv
    _, noisy, _, data = _case(dev, "grid")
    torch.manual_seed(cin * cout)
    ref = oracle.GCNConvRef(cin, cout)
    with torch.no_grad():
        ref.bias.normal_()
    ours = GCNConv(cin, cout).to(dev)
    assert [n for n, _ in ours.named_parameters()] == [n for n, _ in ref.named_parameters()]
    a = (6.0 / (cin + cout)) ** 0.5
    assert float(ours.lin.weight.abs().max()) <= a and float(ours.bias.abs().max()) == 0.0
    ours.load_state_dict(ref.state_dict())
    x = torch.randn(len(noisy.vs), cin)
    dy = torch.randn(len(noisy.vs), cout)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr, data.edge_index)
    yr.backward(dy)
    xo = x.to(dev).requires_grad_(True)
    yo = ours(xo, data.edge_index.to(dev))
    yo.backward(dy.to(dev))
    assert relerr(yo, yr) < 1e-5
    assert relerr(xo.grad, xr.grad) < 1e-5
    assert relerr(ours.lin.weight.grad, ref.lin.weight.grad) < 1e-5
    assert relerr(ours.bias.grad, ref.bias.grad) < 1e-5


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("which", ["ico3", "grid"])
def test_nets_forward_backward_match_oracle(dev, oracle, fused, which):
    from dual_dmp_amd.networks import PosNet, NormalNet
    gt, noisy, smooth, data = _case(dev, which)
    odata = oracle.OracleDataset(noisy, smooth)
    assert torch.equal(odata.z1, data.z1) and torch.equal(odata.edge_index, data.edge_index)
    assert torch.equal(odata.z2, data.z2) and torch.equal(odata.face_index, data.face_index)
    for Ref, Ours, n_out, count in ((oracle.PosNetRef, PosNet, len(noisy.vs), 749955),
                                    (oracle.NormalNetRef, NormalNet, len(noisy.faces), 749667)):
        torch.manual_seed(5)
        ref = Ref()
        with torch.no_grad():                      # make the (otherwise zero) conv biases non-trivial
            for i in range(1, 13):
                getattr(ref, "conv%d" % i).bias.normal_(std=0.1)
                getattr(ref, "bn%d" % i).weight.uniform_(0.5, 1.5)
                getattr(ref, "bn%d" % i).bias.normal_(std=0.1)
        net = Ours(dev, fused=fused)
        net.load_state_dict(ref.state_dict())
        if fused:
            assert net.num_parameters() == count
        else:
            assert sum(p.numel() for p in net.parameters()) == count
        dout = torch.randn(n_out, 3)
        ref.train()
        o_ref = ref(odata)
        o_ref.backward(dout)
        net.train()
        o = net(data)
        o.backward(dout.to(dev))
        assert relerr(o, o_ref) < 1e-4, (Ours.__name__, relerr(o, o_ref))
        ref_grads = {n: p.grad for n, p in ref.named_parameters()}
        got = net.named_views(grads=True) if fused else {n: p.grad for n, p in net.named_parameters()}
        worst = 0.0
        for n, g in ref_grads.items():
            if n.startswith("conv") and n.endswith(".bias"):
                continue                           # analytically zero after BatchNorm (rounding noise only)
            worst = max(worst, relerr(got[n], g))
            assert relerr(got[n], g) < 1e-3, (n, relerr(got[n], g))
        # running statistics follow nn.BatchNorm1d
        sd = net.state_dict()
        assert relerr(sd["bn12.running_mean"], ref.bn12.running_mean) < 1e-4
        assert relerr(sd["bn12.running_var"], ref.bn12.running_var) < 1e-4
        assert int(sd["bn3.num_batches_tracked"]) == 1


def _oracle_run(oracle, noisy, smooth, sd_pos, sd_norm, steps, **kw):
    odata = oracle.OracleDataset(noisy, smooth)
    posnet, normnet = oracle.PosNetRef(), oracle.NormalNetRef()
    posnet.load_state_dict(sd_pos)
    normnet.load_state_dict(sd_norm)
    args = oracle.StepArgs(**kw)
    op = torch.optim.Adam(posnet.parameters(), lr=args.pos_lr)
    on = torch.optim.Adam(normnet.parameters(), lr=args.norm_lr)
    hist = []
    for ep in range(1, steps + 1):
        hist.append(oracle.train_step(posnet, normnet, op, on, odata, noisy, args, ep + kw.get("_ep0", 0)))
    return hist


@pytest.mark.parametrize("bnfloop,ep0", [(1, 0), (5, 100)])
def test_training_steps_match_oracle(dev, oracle, bnfloop, ep0):
    """10 iterations of main.py:88-110 from identical weights: fused trainer vs reference-shaped autograd
    loop (our nets + our losses + torch Adam / clip) vs the CPU oracle."""
    from dual_dmp_amd.networks import PosNet, NormalNet
    from dual_dmp_amd.trainer import FusedTrainer
    from dual_dmp_amd import loss as L
    gt, noisy, smooth, data = _case(dev, "ico3")
    torch.manual_seed(11)
    sd_pos, sd_norm = oracle.PosNetRef().state_dict(), oracle.NormalNetRef().state_dict()
    steps = 10
    k = (3.0, 4.0, 4.0, 4.0, 1.0)
    hist = _oracle_run(oracle, noisy, smooth, sd_pos, sd_norm, steps, bnfloop=bnfloop, _ep0=ep0)

    # (a) fused trainer
    posnet, normnet = PosNet(dev), NormalNet(dev)
    posnet.load_state_dict(sd_pos)
    normnet.load_state_dict(sd_norm)
    tr = FusedTrainer(posnet, normnet, data, noisy, k=k, bnfloop=bnfloop)
    tr.epoch = ep0
    for s in range(steps):
        loss = tr.step().item()
        ref_loss, ref_pos, ref_norm, parts = hist[s]
        assert abs(loss - ref_loss) <= 2e-4 * abs(ref_loss), (s, loss, ref_loss)
        assert float((tr.pos.cpu() - ref_pos).abs().max()) < 1e-3, s
        assert float((tr.norm.cpu() - ref_norm).abs().max()) < 2e-3, s
    fn_o, _ = oracle.face_normals_np(hist[-1][1].numpy().astype(np.float64), noisy.faces)
    fn_h, _ = oracle.face_normals_np(tr.pos.cpu().numpy().astype(np.float64), noisy.faces)
    assert abs(oracle.mad_np(fn_o, gt.fn) - oracle.mad_np(fn_h, gt.fn)) < 1e-3

    # (b) the reference's loop shape on our modules (autograd + torch optimisers)
    posnet, normnet = PosNet(dev), NormalNet(dev)
    posnet.load_state_dict(sd_pos)
    normnet.load_state_dict(sd_norm)
    op = torch.optim.Adam(posnet.parameters(), lr=0.01)
    on = torch.optim.Adam(normnet.parameters(), lr=0.01)
    for s in range(4):
        epoch = s + 1 + ep0
        posnet.train(); normnet.train()
        op.zero_grad(); on.zero_grad()
        pos = posnet(data)
        l1 = L.pos_rec_loss(pos, noisy.vs)
        l2 = L.mesh_laplacian_loss(pos, noisy)
        norm = normnet(data)
        l3 = L.norm_rec_loss(norm, noisy.fn)
        l4, _ = L.fn_bnf_loss(pos, norm, noisy, loop=bnfloop)
        if epoch <= 100:
            l4 = l4 * 0.0
        l5 = L.pos_norm_loss(pos, norm, noisy)
        loss = k[0] * l1 + k[1] * l2 + k[2] * l3 + k[3] * l4 + k[4] * l5
        loss.backward()
        torch.nn.utils.clip_grad_norm_(normnet.parameters(), 0.8)
        op.step(); on.step()
        assert abs(loss.item() - hist[s][0]) <= 2e-4 * abs(hist[s][0]), (s, loss.item(), hist[s][0])
        assert float((pos.detach().cpu() - hist[s][1]).abs().max()) < 1e-3
