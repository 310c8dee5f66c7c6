"""Pin the oracle against golden vectors captured from the reference itself
(tests/golden/make_golden.py: util/loss.py, util/mesh.py, util/models.py run here)."""
import os
import types

import numpy as np
import pytest
import torch

NAMES = ["ico2", "grid4", "cube3", "grid7x5"]


def _load(golden_dir, kind, name):
    return np.load(os.path.join(golden_dir, "%s_%s.npz" % (kind, name)))


def _mesh_ns(gm):
    v2v = torch.sparse_coo_tensor(torch.from_numpy(gm["v2v_indices"]), torch.from_numpy(gm["v2v_values"]),
                                  size=(len(gm["vs"]), len(gm["vs"])))
    return types.SimpleNamespace(vs=gm["vs"], faces=gm["faces"], f2f=gm["f2f"], fn=gm["fn"],
                                 v2v_mat=v2v, v_dims=torch.from_numpy(gm["v_dims"]))


def _grad(fn, *ts):
    leaves = [torch.from_numpy(t).clone().requires_grad_(True) for t in ts]
    out = fn(*leaves)
    extra = None
    if isinstance(out, tuple):
        out, extra = out
    g = torch.autograd.grad(out, leaves, allow_unused=True)
    g = [torch.zeros_like(l) if x is None else x for x, l in zip(g, leaves)]
    return out.detach(), extra, g


@pytest.mark.parametrize("name", NAMES)
def test_mesh_tables_match_reference(oracle, golden_dir, name):
    gm = _load(golden_dir, "mesh", name)
    t = oracle.mesh_tables_loops(gm["vs"], gm["faces"])
    assert t["edges"].dtype == np.int32 and np.array_equal(t["edges"], gm["edges"])
    assert np.array_equal(t["v_dims"], gm["v_dims"])
    # neighbour order inside a row follows CPython set iteration: compare rows as sets
    assert np.array_equal(np.sort(t["f2f"], 1), np.sort(gm["f2f"], 1))
    a = set(map(tuple, t["f_edges"].T.tolist()))
    b = set(map(tuple, gm["f_edges"].T.tolist()))
    assert a == b and t["f_edges"].shape == gm["f_edges"].shape
    fn, fa = oracle.face_normals_np(gm["vs"], gm["faces"])
    assert np.array_equal(fn, gm["fn"]) and np.array_equal(fa, gm["fa"])


@pytest.mark.parametrize("name", NAMES)
def test_losses_match_reference(oracle, golden_dir, name):
    gl = _load(golden_dir, "loss", name)
    m = _mesh_ns(_load(golden_dir, "mesh", name))
    pos, nrm = gl["pos"], gl["norm"]
    tol = dict(rtol=1e-6, atol=1e-7)

    l, _, g = _grad(lambda p: oracle.pos_rec_loss(p, m.vs), pos)
    assert l.dtype == torch.float64            # float64 promotion quirk (util/loss.py:20-31)
    np.testing.assert_allclose(l.numpy(), gl["pos_rec"], rtol=1e-12)
    np.testing.assert_allclose(g[0].numpy(), gl["pos_rec_dpos"], **tol)

    l, _, g = _grad(lambda p: oracle.mesh_laplacian_loss(p, m.v2v_mat, m.v_dims), pos)
    assert l.dtype == torch.float32
    np.testing.assert_allclose(l.numpy(), gl["lap"], **tol)
    np.testing.assert_allclose(g[0].numpy(), gl["lap_dpos"], **tol)

    l, _, g = _grad(lambda n: oracle.norm_rec_loss(n, m.fn), nrm)
    assert l.dtype == torch.float64
    np.testing.assert_allclose(l.numpy(), gl["norm_rec"], rtol=1e-12)
    np.testing.assert_allclose(g[0].numpy(), gl["norm_rec_dnorm"], **tol)

    for loop in (1, 5):
        l, new_fn, g = _grad(lambda p, n: oracle.fn_bnf_loss(p, n, m.faces, m.f2f, loop=loop), pos, nrm)
        np.testing.assert_allclose(l.numpy(), gl["bnf%d" % loop], **tol)
        np.testing.assert_allclose(new_fn.detach().numpy(), gl["bnf%d_newfn" % loop], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(g[1].numpy(), gl["bnf%d_dnorm" % loop], rtol=1e-4, atol=1e-6)
        assert np.all(g[0].numpy() == 0) and np.all(gl["bnf%d_dpos" % loop] == 0)   # pos detached

    l, _, g = _grad(lambda p, n: oracle.pos_norm_loss(p, n, m.faces, len(m.vs)), pos, nrm)
    np.testing.assert_allclose(l.numpy(), gl["pos_norm"], **tol)
    np.testing.assert_allclose(g[0].numpy(), gl["pos_norm_dpos"], **tol)
    np.testing.assert_allclose(g[1].numpy(), gl["pos_norm_dnorm"], **tol)

    args = oracle.StepArgs()
    l, _, g = _grad(lambda p, n: oracle.losses(p, n, m, args, epoch=101)[0], pos, nrm)
    assert l.dtype == torch.float64
    np.testing.assert_allclose(l.numpy(), gl["total"], rtol=1e-6)
    np.testing.assert_allclose(g[0].numpy(), gl["total_dpos"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(g[1].numpy(), gl["total_dnorm"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("name", NAMES)
def test_metric_matches_reference(oracle, golden_dir, name):
    gl = _load(golden_dir, "loss", name)
    gm = _load(golden_dir, "mesh", name)
    assert oracle.mad_np(torch.from_numpy(gl["norm"]), gm["fn"]) == pytest.approx(float(gl["mad"]), rel=1e-12)
    fn, fa = oracle.face_normals_np(gl["pos"].astype(np.float64), gm["faces"])
    np.testing.assert_allclose(fn, gl["cfn_fn"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(fa, gl["cfn_fa"], rtol=1e-13)
    assert oracle.mad_np(fn, gm["fn"]) == pytest.approx(float(gl["mad_pos"]), rel=1e-12)
    # util/models.py:5-10 (float32 torch) agrees with the float64 normals to f32 rounding
    np.testing.assert_allclose(gl["models_compute_fn"], fn, atol=2e-6)


@pytest.mark.parametrize("name", NAMES)
def test_bnf_matches_reference(oracle, golden_dir, name):
    """util/loss.py:195-259 (classical bilateral normal filter + vertex update; round 5)."""
    gb = _load(golden_dir, "bnf", name)
    gl = _load(golden_dir, "loss", name)
    gm = _load(golden_dir, "mesh", name)
    for it in (1, 3):
        nf, vs, fc, fa = oracle.bnf_np(gl["norm"].astype(np.float64), gm["vs"], gm["faces"], gm["f2f"], gm["fc"], gm["fa"], iters=it)
        np.testing.assert_allclose(nf, gb["newfn_%d" % it], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(vs, gb["vs_%d" % it], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(fc, gb["fc_%d" % it], rtol=1e-10, atol=1e-12)     # iter 1: the input mesh's, untouched
        np.testing.assert_allclose(fa, gb["fa_%d" % it], rtol=1e-10, atol=1e-12)
    nf, vs, _, _ = oracle.bnf_np(gl["norm"], gm["vs"], gm["faces"], gm["f2f"], gm["fc"], gm["fa"], 0.5, 0.3, 2)
    np.testing.assert_allclose(nf, gb["newfn_f32in"], rtol=0, atol=2e-6)             # float32 input: first-sweep |dn| in float32
    np.testing.assert_allclose(vs, gb["vs_f32in"], rtol=0, atol=2e-6)


def test_gcnconv_pyg_shape_vs_dense(oracle, golden_dir):
    """The PyG-shaped restatement against the independent float64 dense form
    (parity with real PyG is unpinned: the package is absent, see oracle header)."""
    gm = _load(golden_dir, "mesh", "ico2")
    e = torch.from_numpy(gm["edges"].T.astype(np.int64))
    ei = torch.cat([e, e[[1, 0]]], 1)
    torch.manual_seed(0)
    conv = oracle.GCNConvRef(16, 32)
    with torch.no_grad():
        conv.bias.normal_()
    x = torch.randn(len(gm["vs"]), 16)
    y = conv(x, ei)
    yd = oracle.gcn_conv_dense(x, conv.lin.weight.detach(), conv.bias.detach(), ei)
    assert torch.allclose(y.double(), yd, rtol=1e-5, atol=1e-5)
    a = (6.0 / 48.0) ** 0.5
    assert conv.lin.weight.abs().max() <= a and conv.lin.weight.shape == (32, 16)
    # multi-edges count with multiplicity
    ei2 = torch.cat([ei, ei[:, :7]], 1)
    assert torch.allclose(conv(x, ei2).double(),
                          oracle.gcn_conv_dense(x, conv.lin.weight.detach(), conv.bias.detach(), ei2),
                          rtol=1e-5, atol=1e-5)


def test_param_counts(oracle):
    """PosNet 749,955 / NormalNet 749,667 parameters."""
    assert sum(p.numel() for p in oracle.PosNetRef().parameters()) == 749955
    assert sum(p.numel() for p in oracle.NormalNetRef().parameters()) == 749667
