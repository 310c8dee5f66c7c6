"""bf16-feature mode (BASELINE.json configs[1]): every bf16 kernel, called through the C ABI, against float64
restatements evaluated on the SAME bf16-rounded inputs, then the two nets and the training iteration against the oracle.

Tolerances.  A bf16 kernel unpacks to float32, accumulates in float32 and rounds ONCE on the store, so against a float64
reference on identical (already rounded) inputs an output element is off by at most half a bf16 ulp (2^-9 relative)
plus float32 accumulation noise: asserted as |got - ref| <= 2^-8 |ref| + atol with atol scaled to the accumulation.
Float32 outputs of bf16 inputs (weight gradients, column sums, head outputs) are float32-exact: rel-L2 <= 1e-5.
End to end (12 layers of rounded features) the error is a property of the format, not of a kernel: the measured
figures are asserted with ~2x head-room and written next to each assert.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BF = torch.bfloat16
EPS = 2.0 ** -8


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def close_bf16(got, ref, atol):
    """elementwise: one bf16 rounding of the exact result + atol of accumulation noise"""
    got, ref = got.double().cpu(), ref.double().cpu()
    bad = (got - ref).abs() > EPS * ref.abs() + atol
    return int(bad.sum()), float(((got - ref).abs() - EPS * ref.abs()).max())


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def graphs(dev):
    from dual_dmp_amd import synth
    from dual_dmp_amd.mesh import Mesh
    out = {}
    for name, (v, f) in {"ico3": synth.icosphere(3), "grid": synth.open_grid(9, 7)}.items():
        v, f = synth.permute_vertices(v, f, 1)
        m = Mesh(vs=v, faces=f)
        e = torch.tensor(m.edges.T, dtype=torch.long)
        out[name + "_v"] = (torch.cat([e, e[[1, 0]]], 1), len(v))
        out[name + "_f"] = (torch.from_numpy(m.f_edges), len(f))
    return out


def dense_ahat(ei, n):
    A = torch.zeros(n, n, dtype=torch.float64)
    A.index_put_((ei[1], ei[0]), torch.ones(ei.shape[1], dtype=torch.float64), accumulate=True)
    A += torch.eye(n, dtype=torch.float64)
    d = A.sum(1).pow(-0.5)
    return d[:, None] * A * d[None, :]


def f_ref(x, a, b, slope=0.01):
    z = x * a + b
    return torch.where(z > 0, z, slope * z)


def bn_bwd_ref(dz, y, bn4, c10, slope=0.01):
    a, b = bn4[0].double(), bn4[1].double()
    g = dz * torch.where(y * a + b > 0, 1.0, slope)
    return a * g + c10[0].double() * y + c10[1].double()


def rb(x):
    """bf16 tensor + its exact float64 value"""
    xb = x.to(BF)
    return xb, xb.double()


def test_conversion_round_to_nearest_even(dev):
    from dual_dmp_amd import ops
    x = torch.randn(4097, device=dev) * torch.logspace(-20, 20, 4097, device=dev)
    x[:4] = torch.tensor([1.00390625, 1.01171875, -1.00390625, 0.0], device=dev)     # ties: to even
    assert torch.equal(ops.to_bf16(x), x.to(BF))


@pytest.mark.parametrize("C", [8, 16, 32, 64, 128, 256, 512, 192])
@pytest.mark.parametrize("gname", ["ico3_v", "grid_f"])
def test_spmm_bf16_matches_dense(dev, graphs, C, gname):
    from dual_dmp_amd import ops
    ei, n = graphs[gname]
    torch.manual_seed(C)
    xb, x = rb(torch.randn(n, C))
    bias = torch.randn(C)
    a, b = torch.rand(C) + 0.5, torch.randn(C)
    A = dense_ahat(ei, n)
    g = ops.graph_for(ei.to(dev), n)
    y = ops.spmm(g, xb.to(dev))
    assert y.dtype == BF
    nbad, worst = close_bf16(y, A @ x, 1e-5)
    assert nbad == 0, worst
    y = ops.spmm(g, xb.to(dev), bias=bias.to(dev), pro=(a.to(dev), b.to(dev)))
    ref = A @ f_ref(x, a.double(), b.double()) + bias.double()
    nbad, worst = close_bf16(y, ref, 1e-5)
    assert nbad == 0, worst


@pytest.mark.parametrize("C", [32, 256, 64])
def test_spmm_bf16_fused_backward_forms(dev, graphs, C):
    """bnred: SpMM + the BatchNorm-backward column reductions of its (rounded) output; bnbwd: A_hat . dY with dY rebuilt
    on the gather from (dZ, Y)."""
    from dual_dmp_amd import ops
    for gname in ("ico3_f", "grid_v"):
        ei, n = graphs[gname]
        torch.manual_seed(C + n)
        A = dense_ahat(ei, n)
        g = ops.graph_for(ei.to(dev), n)
        xb, x = rb(torch.randn(n, C))
        ypb, yp = rb(torch.randn(n, C) * 2 + 0.3)
        bn4 = torch.stack([torch.rand(C) + 0.5, torch.randn(C), torch.randn(C), torch.rand(C) + 0.5])
        out = torch.empty(n, C, dtype=BF, device=dev)
        sums = torch.zeros(2 * C, dtype=torch.float64, device=dev)
        ops.spmm_bnred(g, xb.to(dev), out, ypb.to(dev), bn4.to(dev), sums)
        assert torch.equal(out, ops.spmm(g, xb.to(dev)))                # the same kernel arithmetic
        o = out.double().cpu()
        gg = o * torch.where(yp * bn4[0].double() + bn4[1].double() > 0, 1.0, 0.01)
        ref = torch.cat([gg.sum(0), (gg * (yp - bn4[2].double()) * bn4[3].double()).sum(0)])
        assert relerr(sums, ref) < 2e-5, relerr(sums, ref)              # float32 partials per 64 rows
        # gather form
        c10 = torch.stack([torch.randn(C) * 0.1, torch.randn(C) * 0.1])
        dzb, dz = rb(torch.randn(n, C))
        res = torch.empty(n, C, dtype=BF, device=dev)
        ops.spmm_bnbwd(g, dzb.to(dev), ypb.to(dev), bn4.to(dev), c10.to(dev), res)
        ref = A @ bn_bwd_ref(dz, yp, bn4, c10)
        nbad, worst = close_bf16(res, ref, 2e-5)
        assert nbad == 0, worst


GEMM_SHAPES = [(1000, 32, 64), (777, 64, 32), (3000, 512, 512), (2049, 256, 512), (2500, 512, 256), (1500, 128, 256),
               (1300, 256, 128), (900, 128, 64), (5000, 16, 32), (5000, 8, 32), (130, 256, 256)]


@pytest.mark.parametrize("n,K,M", GEMM_SHAPES)
def test_gemm_nt_bf16(dev, n, K, M):
    """Y = f(A) W^T + b: bf16 A, float32 W (rounded to bf16 by the kernel), bf16 Y."""
    from dual_dmp_amd import ops
    torch.manual_seed(n + K + M)
    ab, a = rb(torch.randn(n, K))
    w = torch.randn(M, K) / K ** 0.5
    wq = w.to(BF).double() if K >= 32 else w.double()          # K = 8 | 16 (first layer): plain float32 weights on the VALU
    bias = torch.randn(M)
    y = ops.gemm_nt(ab.to(dev), w.to(dev), bias=bias.to(dev))
    assert y.dtype == BF and y.shape == (n, M)
    atol = 3e-6 * float((a.abs() @ wq.abs().t()).max())
    nbad, worst = close_bf16(y, a @ wq.t() + bias.double(), atol)
    assert nbad == 0, worst
    if K >= 32:
        sc, sh = torch.rand(K) + 0.5, torch.randn(K) * 0.3
        y = ops.gemm_nt(ab.to(dev), w.to(dev), pro=(sc.to(dev), sh.to(dev)))
        z = f_ref(a, sc.double(), sh.double()).float().to(BF).double()       # the prologue result is rounded to bf16
        # (the kernel rounds lrelu(fma(...)) computed in float32: identical up to float32 rounding before the bf16 one,
        #  which can move an element by one bf16 ulp -- hence the rel-L2 form)
        assert relerr(y, z @ wq.t()) < 4e-3


@pytest.mark.parametrize("n,K,M", [(70000, 64, 16), (70000, 128, 40), (70000, 256, 48), (70001, 512, 200), (33000, 64, 24)])
def test_gemm_nt_bf16_partial_width_panels(dev, n, K, M):
    """Output widths below the padded panel width (M < 32 NJ WC): whole waves issue no epilogue stores, so the counted
    vmcnt wait behind an epilogue must not assume them (round-2 advisor finding; a miscount leaves global->LDS copies of
    the next stage in flight at the barrier -- timing dependent, hence the repeats on a many-tile input)."""
    from dual_dmp_amd import ops
    torch.manual_seed(n + K + M)
    ab, a = rb(torch.randn(n, K))
    w = torch.randn(M, K) / K ** 0.5
    wq = w.to(BF).double()
    ref = a @ wq.t()
    atol = 3e-6 * float((a.abs() @ wq.abs().t()).max())
    ad, wd = ab.to(dev), w.to(dev)
    first = None
    for rep in range(6):
        y = ops.gemm_nt(ad, wd)
        nbad, worst = close_bf16(y, ref, atol)
        assert nbad == 0, (rep, worst)
        first = y if first is None else first
        assert torch.equal(y, first), rep


RR_SHAPES = [(33000, 256, 512), (70001, 512, 512), (40000, 64, 256), (25000, 128, 200), (21000, 512, 136), (20001, 32, 256),
             (66000, 256, 256)]


@pytest.mark.parametrize("n,K,M", RR_SHAPES)
def test_gemm_rr_bf16_forward_statistics_and_prologue(dev, n, K, M):
    """The row-register bf16 kernel (csrc/gemm_rr_b16.inc; >= 20k rows, 64 < M <= 512): plain, with the BatchNorm+LeakyReLU
    prologue, with bias, and with the BatchNorm statistics of the STORED (bf16) output from the epilogue -- the latter must
    be the float64 column sums of exactly what was stored (== ddmp_bn_stats_bf16 of the output)."""
    from dual_dmp_amd import ops
    if os.environ.get("DDMP_GEMM_RR") == "0":
        pytest.skip("row-register kernels disabled")
    from dual_dmp_amd import _lib
    assert _lib.lib().ddmp_gemm_fused_bf16_supported(M, K, n) & 1
    torch.manual_seed(n + K + M)
    ab, a = rb(torch.randn(n, K) + 0.2)
    w = torch.randn(M, K) / K ** 0.5
    wq = w.to(BF).double()
    bias = torch.randn(M)
    ad, wd = ab.to(dev), w.to(dev)
    ref = a @ wq.t() + bias.double()
    atol = 3e-6 * float((a.abs() @ wq.abs().t()).max())
    sums = torch.zeros(2 * M, dtype=torch.float64, device=dev)
    y = ops.gemm_nt_stats(ad, wd, sums, bias=bias.to(dev))
    assert y.dtype == BF and y.shape == (n, M)
    nbad, worst = close_bf16(y, ref, atol)
    assert nbad == 0, worst
    yd = y.double().cpu()
    assert relerr(sums[:M], yd.sum(0)) < 1e-12 and relerr(sums[M:], (yd * yd).sum(0)) < 1e-12
    nbad, worst = close_bf16(ops.gemm_nt(ad, wd, bias=bias.to(dev)), ref, atol)    # (the plain form: either kernel)
    assert nbad == 0, worst
    sc, sh = torch.rand(K) + 0.5, torch.randn(K) * 0.3
    y2 = ops.gemm_nt_stats(ad, wd, sums, pro=(sc.to(dev), sh.to(dev)))
    z = f_ref(a, sc.double(), sh.double()).float().to(BF).double()
    assert relerr(y2, z @ wq.t()) < 4e-3
    yd = y2.double().cpu()
    assert relerr(sums[:M], yd.sum(0)) < 1e-12 and relerr(sums[M:], (yd * yd).sum(0)) < 1e-12
    for _ in range(3):                                                              # stable from launch to launch
        assert torch.equal(ops.gemm_nt_stats(ad, wd, sums, pro=(sc.to(dev), sh.to(dev))), y2)


@pytest.mark.parametrize("n,cin,cout", [(66000, 256, 512), (70001, 512, 512), (33000, 128, 256), (40000, 256, 256)])
def test_gemm_bnbwd_bf16_fused_matches_composition(dev, n, cin, cout, monkeypatch):
    """bf16 dgrad / wgrad with the BatchNorm+LeakyReLU backward rebuilt on the operand load == ddmp_bn_bwd_apply_bf16 followed
    by the plain GEMMs: the rebuilt dY is rounded to bf16 exactly as the pass would have stored it, so the products agree to
    float32 accumulation order; and == the float64 formula on the rounded operands."""
    from dual_dmp_amd import ops
    if os.environ.get("DDMP_GEMM_RR") == "0":
        pytest.skip("fused bf16 GEMMs not available")
    # (512 <- 512 is not DISPATCHED by the engine -- it measured slower than the separate pass -- but the kernels take it)
    torch.manual_seed(n + cin)
    dzb, dz = rb(torch.randn(n, cout))
    ybb, yb = rb(torch.randn(n, cout) * 2 + 0.5)
    pb, p = rb(torch.randn(n, cin))
    w = torch.randn(cout, cin) / cout ** 0.5
    bn4 = torch.stack([torch.rand(cout) + 0.5, torch.randn(cout), torch.randn(cout), torch.rand(cout) + 0.5])
    c10 = torch.stack([torch.randn(cout) * 0.1, torch.randn(cout) * 0.1])
    dzg, ybg, wg, pg, bn4g, c10g = (t.to(dev) for t in (dzb, ybb, w, pb, bn4, c10))
    dy = torch.empty_like(dzg)
    sums = torch.empty(2 * cout, dtype=torch.float64, device=dev)
    ops.bn_bwd_apply(dzg, ybg, bn4g, c10g, dy, sums)
    dyd = dy.double().cpu()
    nbad, worst = close_bf16(dy, bn_bwd_ref(dz, yb, bn4, c10), 2e-6)
    assert nbad == 0, worst
    dx = ops.gemm_nn_bnbwd(dzg, ybg, wg, bn4g, c10g)
    assert dx.dtype == BF and dx.shape == (n, cin)
    wq = w.to(BF).double()
    atol = 3e-6 * float((dyd.abs() @ wq.abs()).max())
    nbad, worst = close_bf16(dx, dyd @ wq, atol)
    assert nbad == 0, worst
    dw = ops.gemm_tn_bnbwd(dzg, ybg, pg, bn4g, c10g)
    assert dw.dtype == torch.float32 and relerr(dw, dyd.t() @ p) < 1e-5
    assert relerr(dw, ops.gemm_tn(dy, pg)) < 2e-6                                   # same operands, same arithmetic


@pytest.mark.parametrize("n,M,K", [(1000, 64, 32), (3000, 512, 512), (2049, 512, 256), (2500, 256, 512), (900, 32, 64),
                                    (1300, 128, 256), (130, 256, 256), (66000, 512, 256), (33001, 256, 512), (20500, 128, 128)])
def test_gemm_nn_bf16(dev, n, M, K):
    from dual_dmp_amd import ops
    torch.manual_seed(n + K + M)
    ab, a = rb(torch.randn(n, M))
    w = torch.randn(M, K) / M ** 0.5
    wq = w.to(BF).double()
    y = ops.gemm_nn(ab.to(dev), w.to(dev))
    assert y.dtype == BF and y.shape == (n, K)
    atol = 3e-6 * float((a.abs() @ wq.abs()).max())
    nbad, worst = close_bf16(y, a @ wq, atol)
    assert nbad == 0, worst


@pytest.mark.parametrize("n,M,K", [(1000, 64, 32), (40000, 512, 512), (20000, 256, 512), (3000, 512, 256), (777, 32, 16),
                                    (5000, 32, 8), (9000, 128, 64), (2500, 256, 256), (33, 64, 64)])
def test_gemm_tn_bf16(dev, n, M, K):
    """dW = G^T f(Z): bf16 operands, float32 result; exercises the ds_read_b64_tr_b16 fragment path with asymmetric data."""
    from dual_dmp_amd import ops
    torch.manual_seed(n + K + M)
    gb, g = rb(torch.randn(n, M) * torch.linspace(0.5, 2.0, M))
    zb, z = rb(torch.randn(n, K) + torch.linspace(-1.0, 1.0, K))
    dw = ops.gemm_tn(gb.to(dev), zb.to(dev))
    assert dw.dtype == torch.float32 and dw.shape == (M, K)
    assert relerr(dw, g.t() @ z) < 1e-5, relerr(dw, g.t() @ z)
    sc, sh = torch.rand(K) + 0.5, torch.randn(K) * 0.3
    dw = ops.gemm_tn(gb.to(dev), zb.to(dev), pro=(sc.to(dev), sh.to(dev)))
    zz = f_ref(z, sc.double(), sh.double()).float().to(BF).double()
    assert relerr(dw, g.t() @ zz) < 2e-3


@pytest.mark.parametrize("C", [16, 32, 128, 512])
def test_batchnorm_passes_bf16(dev, C):
    from dual_dmp_amd import ops
    n = 3001
    torch.manual_seed(C)
    yb, y = rb(torch.randn(n, C) * 1.7 + 0.4)
    dzb, dz = rb(torch.randn(n, C))
    sums = ops.bn_stats(yb.to(dev))
    assert relerr(sums, torch.cat([y.sum(0), (y * y).sum(0)])) < 1e-12
    bn4 = torch.stack([torch.rand(C) + 0.5, torch.randn(C), y.mean(0).float(), torch.rand(C) + 0.5])
    s2 = ops.bn_bwd_reduce(dzb.to(dev), yb.to(dev), bn4.to(dev))
    gg = dz * torch.where(y * bn4[0].double() + bn4[1].double() > 0, 1.0, 0.01)
    ref = torch.cat([gg.sum(0), (gg * ((y.float() - bn4[2]) * bn4[3]).double()).sum(0)])
    assert relerr(s2, ref) < 1e-6
    c10 = torch.stack([torch.randn(C) * 0.1, torch.randn(C) * 0.1])
    dy = torch.empty(n, C, dtype=BF, device=dev)
    db = torch.empty(2 * C, dtype=torch.float64, device=dev)
    ops.bn_bwd_apply(dzb.to(dev), yb.to(dev), bn4.to(dev), c10.to(dev), dy, db)
    nbad, worst = close_bf16(dy, bn_bwd_ref(dz, y, bn4, c10), 1e-6)
    assert nbad == 0, worst
    assert relerr(db[:C], dy.double().sum(0)) < 1e-12                   # column sums of the values as stored
    z = ops.bn_lrelu_apply(yb.to(dev), bn4[0].to(dev), bn4[1].to(dev))
    nbad, worst = close_bf16(z, f_ref(y, bn4[0].double(), bn4[1].double()), 1e-6)
    assert nbad == 0, worst


@pytest.mark.parametrize("kind", [0, 1])
def test_heads_bf16_equal_the_float32_heads_on_the_same_values(dev, kind):
    from dual_dmp_amd import ops
    n = 1500
    torch.manual_seed(kind)
    yb = (torch.randn(n, 32) * 1.5).to(BF).to(dev)
    bn4 = torch.stack([torch.rand(32) + 0.5, torch.randn(32) * 0.2, torch.zeros(32), torch.ones(32)]).to(dev)
    W1, b1 = (torch.randn(16, 32) * 0.2).to(dev), (torch.randn(16) * 0.1).to(dev)
    W2, b2 = (torch.randn(3, 16) * 0.3).to(dev), (torch.randn(3) * 0.1).to(dev)
    xp = torch.randn(n, 3, device=dev)
    dout = torch.randn(n, 3, device=dev)
    res = {}
    for name, y in (("bf16", yb), ("f32", yb.float())):
        out = torch.empty(n, 3, device=dev)
        ops.head_fwd(y, bn4, W1, b1, W2, b2, kind, xp, out)
        dz = torch.empty(n, 32, dtype=y.dtype, device=dev)
        gr = [torch.empty_like(t) for t in (W1, b1, W2, b2)]
        ops.head_bwd(y, bn4, W1, b1, W2, b2, kind, dout, dz, *gr)
        res[name] = (out, dz, gr)
    assert torch.equal(res["bf16"][0], res["f32"][0])
    assert torch.equal(res["bf16"][1], res["f32"][1].to(BF))
    for a, b in zip(res["bf16"][2], res["f32"][2]):
        assert torch.equal(a, b)


def _nets(dev, oracle, dtype, mesh="ico3"):
    from dual_dmp_amd import synth
    from dual_dmp_amd.datamaker import dataset_from_meshes
    from dual_dmp_amd.networks import PosNet, NormalNet
    v, f = synth.icosphere(3) if mesh == "ico3" else synth.open_grid(24, 17)
    gt, noisy, smooth = synth.make_triplet(v, f)
    data = dataset_from_meshes(noisy, smooth)
    torch.manual_seed(0)
    ref_pos, ref_norm = oracle.PosNetRef(), oracle.NormalNetRef()
    posnet, normnet = PosNet(dev, dtype=dtype), NormalNet(dev, dtype=dtype)
    posnet.load_state_dict(ref_pos.state_dict())
    normnet.load_state_dict(ref_norm.state_dict())
    return gt, noisy, smooth, data, ref_pos, ref_norm, posnet, normnet


@pytest.mark.parametrize("mesh", ["ico3", "grid"])
def test_nets_bf16_forward_backward_vs_oracle(dev, oracle, mesh):
    """Whole-net forward and parameter gradients of the bf16-feature mode against the float32 oracle from identical
    (random-init) weights.  Measured on MI355X (this test prints them; ico3 / grid): forward max|dpos| 1.1e-2 on a
    unit-edge mesh, max|dnorm| 0.15 (rms ~2e-2: unit normals of tiny tanh arguments amplify the relative error),
    weight-gradient rel-L2 0.14 / 0.20 -- not kernel error (every kernel is exact to one rounding, tests above, and the
    engine reproduces a rounding-by-rounding restatement layer by layer, test below) but what 12 layers of 8-bit significands do:
    ~0.4 % of the LeakyReLU units sit within a bf16 ulp of zero and take the other slope.  Asserted with ~2x head-room."""
    gt, noisy, smooth, data, ref_pos, ref_norm, posnet, normnet = _nets(dev, oracle, BF, mesh)
    odata = oracle.OracleDataset(noisy, smooth)
    pos, norm = posnet(data), normnet(data)
    rp, rn = ref_pos(odata), ref_norm(odata)
    e_p = float((pos.detach().cpu() - rp.detach()).abs().max())
    e_n = float((norm.detach().cpu() - rn.detach()).abs().max())
    r_p = float((pos.detach().cpu() - rp.detach()).pow(2).mean().sqrt())
    r_n = float((norm.detach().cpu() - rn.detach()).pow(2).mean().sqrt())
    # a fixed linear functional of the outputs as the loss: gradients of every layer
    torch.manual_seed(1)
    gp, gn = torch.randn_like(rp), torch.randn_like(rn)
    (pos * gp.to(dev)).sum().backward()
    (norm * gn.to(dev)).sum().backward()
    (rp * gp).sum().backward()
    (rn * gn).sum().backward()
    errs = {}
    for net, ref, tag in ((posnet, ref_pos, "pos"), (normnet, ref_norm, "norm")):
        gv = net.named_views(grads=True)
        num = den = 0.0
        for name, p in ref.named_parameters():
            if name.startswith("conv") and name.endswith(".bias"):
                continue                                    # analytically zero after BatchNorm (rounding noise only)
            num += float((gv[name].cpu().double() - p.grad.double()).pow(2).sum())
            den += float(p.grad.double().pow(2).sum())
        errs[tag] = (num / den) ** 0.5
    print("bf16 nets on %s: max|dpos| %.2e (rms %.2e)  max|dnorm| %.2e (rms %.2e)  grad rel-L2 pos %.2e norm %.2e"
          % (mesh, e_p, r_p, e_n, r_n, errs["pos"], errs["norm"]))
    assert e_p < 2.5e-2 and e_n < 0.3 and r_p < 6e-3 and r_n < 6e-2, (e_p, e_n, r_p, r_n)
    assert errs["pos"] < 0.4 and errs["norm"] < 0.4, errs


def test_bf16_engine_layer_by_layer_against_rounding_emulation(dev, oracle):
    """Wiring of the bf16 engine, layer by layer and teacher-forced: layer l is restated in float64 from the engine's
    OWN stored input (Y_{l-1} as bf16, the BatchNorm coefficients the kernels computed), rounding where the kernels
    round, and compared with the engine's stored Y_l.  What may differ is float32-vs-float64 accumulation moving an
    element across a rounding boundary: never by more than one bf16 ulp, and only for a small fraction of the elements."""
    gt, noisy, smooth, data, ref_pos, ref_norm, posnet, normnet = _nets(dev, oracle, BF, "ico3")
    V, F = len(noisy.vs), len(noisy.faces)
    pos, norm = posnet(data), normnet(data)

    def r(t):
        return t.float().to(BF).double()

    def f(x, a, b):
        z = torch.addcmul(b, x.float(), a)
        return torch.where(z > 0, z, 0.01 * z).double()

    worst_frac = 0.0
    for net, ei, n in ((posnet, data.edge_index, V), (normnet, data.face_index, F)):
        eng = net._engine
        perm = eng.perm.cpu()
        A = dense_ahat(ei, n)[perm][:, perm]                     # the engine's (Morton) numbering
        L = eng.layout
        arena = net.arena.detach().cpu()
        X, pro = eng.x0.cpu().double(), None
        for l in range(12):
            W = L.view(arena, "conv%d.lin.weight" % (l + 1), true_shape=False).double()
            b = L.view(arena, "conv%d.bias" % (l + 1)).double()
            Z = X if pro is None else f(X, *pro)
            if eng.agg_first[l]:
                P = r(A @ Z)
                assert torch.equal(eng.P[l].cpu().double(), P) or float((eng.P[l].cpu().double() - P).abs().max()) <= \
                    2.0 ** -7 * float(P.abs().max())
                P = eng.P[l].cpu().double()                      # teacher-forced into the GEMM as well
                Y = r(P @ (W if l == 0 else r(W)).t() + b)
            else:
                H = r(r(Z) @ r(W).t())
                Y = r(A @ H + b)
            got = eng.Y[l].cpu().double()
            ulp = 2.0 ** (torch.floor(torch.log2(torch.maximum(Y.abs(), got.abs()).clamp_min(1e-30))) - 7)    # bf16: 8 significant bits
            d = (got - Y).abs()
            atol = 3e-6 * float(Y.abs().max())                   # float32 accumulation noise (elements that nearly cancel)
            off = d > 1.001 * ulp + atol
            if eng.agg_first[l]:
                assert not bool(off.any()), (l, float(((d - atol) / ulp).max()))
            else:
                # transform-first: the GEMM operand bf16(f(Y_prev)) is produced inside the kernel (fmaf) and here (mul, add):
                # a last-bit difference before the bf16 rounding moves an operand element by one bf16 ulp now and then,
                # i.e. an output by ~|w| ulp(z) ~ 3e-4 |z| -- rare and small, but many ulps of an output that nearly cancels
                assert float(off.double().mean()) < 5e-3 and float(d.max()) < 5e-3 * float(Y.abs().max()), \
                    (l, float(off.double().mean()), float(d.max()))
            frac = float((d > 0).double().mean())
            worst_frac = max(worst_frac, frac)
            # BatchNorm coefficients from the stored values
            mu = got.sum(0) / n
            var = ((got * got).sum(0) / n - mu * mu).clamp_min(0)
            a_ref = L.view(arena, "bn%d.weight" % (l + 1)).double() / (var + 1e-5).sqrt()
            assert relerr(eng.bn4[l][0], a_ref) < 1e-6
            assert float((eng.bn4[l][2].cpu().double() - mu).abs().max()) <= 1e-6 * float(mu.abs().max() + 1)
            X, pro = got, (eng.bn4[l][0].cpu(), eng.bn4[l][1].cpu())
    print("bf16 engine vs per-layer rounding emulation: at most %.3f %% of a layer's elements differ (by one bf16 ulp)" % (100 * worst_frac))
    assert worst_frac < 0.02


def test_training_bf16_tracks_float32(dev, oracle):
    """30 free-running iterations from identical weights: the bf16-feature run against the float32 HIP run.  Measured on
    MI355X: loss within a few % of the float32 run, MAD within 0.3 degrees (the spread of float32 runs across seeds is of
    the same size); asserted: loss within 10 %, MAD within 1 degree, both improving on the noisy input."""
    from dual_dmp_amd import synth
    from dual_dmp_amd.datamaker import dataset_from_meshes
    from dual_dmp_amd.loss import mad
    from dual_dmp_amd.mesh import Mesh
    from dual_dmp_amd.networks import PosNet, NormalNet
    from dual_dmp_amd.trainer import FusedTrainer
    v, f = synth.icosphere(4)
    gt, noisy, smooth = synth.make_triplet(v, f)
    data = dataset_from_meshes(noisy, smooth)
    out = {}
    for dtype in (torch.float32, BF):
        torch.manual_seed(0)
        posnet, normnet = PosNet(dev, dtype=dtype), NormalNet(dev, dtype=dtype)
        tr = FusedTrainer(posnet, normnet, data, noisy)
        losses = [float(tr.step().item()) for _ in range(30)]
        o = Mesh.__new__(Mesh)
        o.vs, o.faces = tr.pos.cpu().numpy().astype(np.float64), noisy.faces
        Mesh.compute_face_normals(o)
        out[dtype] = (losses, float(mad(o.fn, gt.fn)))
    m0 = float(mad(noisy.fn, gt.fn))
    (l32, m32), (l16, m16) = out[torch.float32], out[BF]
    print("30 iterations: loss f32 %.4f bf16 %.4f | MAD noisy %.3f f32 %.3f bf16 %.3f" % (l32[-1], l16[-1], m0, m32, m16))
    assert abs(l16[0] - l32[0]) <= 2e-3 * abs(l32[0])        # first iteration: same weights, forward error only
    assert l16[-1] < l16[0] and abs(l16[-1] - l32[-1]) <= 0.1 * abs(l32[-1])
    assert m16 < m0 and abs(m16 - m32) <= 1.0


def test_bf16_iteration_on_an_irregular_mesh_at_144k(dev):
    """BASELINE.json configs[1]'s shape -- a non-CAD (irregular-valence) mesh with bf16 features: the 144,400-face torus after
    ten rounds of random edge flips + a valence-24 hub (vertex-graph rows of 4 ... 25 entries; both graphs above every
    fused-route threshold, the LDS-patch gather with register entries + LDS tails in its bf16 form).  Five iterations from
    identical weights, hipGraph + two streams as in the bench, against the float32 HIP run: first loss to forward accuracy,
    the fifth within 5 %, outputs finite."""
    from dual_dmp_amd import synth
    from dual_dmp_amd.datamaker import dataset_from_meshes
    from dual_dmp_amd.networks import PosNet, NormalNet
    from dual_dmp_amd.trainer import FusedTrainer
    v, f = synth.torus(380, 190)
    f = synth.flip_edges(v, f, rounds=10, seed=1)
    f = synth.add_hub(v, f, 1000, 24)
    v, f = synth.permute_vertices(v, f, 3)
    gt, noisy, smooth = synth.make_triplet(v, f)
    data = dataset_from_meshes(noisy, smooth)
    out = {}
    for dtype in (torch.float32, BF):
        torch.manual_seed(0)
        tr = FusedTrainer(PosNet(dev, dtype=dtype), NormalNet(dev, dtype=dtype), data, noisy, use_graph=True, overlap=True)
        losses = [float(tr.step().item()) for _ in range(5)]
        assert bool(torch.isfinite(tr.pos).all()) and bool(torch.isfinite(tr.norm).all())
        assert tr.peng.g.max_row_nnz == 25 and tr.neng.g.max_row_nnz == 4
        out[dtype] = losses
        del tr
    l32, l16 = out[torch.float32], out[BF]
    assert abs(l16[0] - l32[0]) <= 2e-3 * abs(l32[0]), (l16[0], l32[0])
    assert abs(l16[-1] - l32[-1]) <= 0.05 * abs(l32[-1]), (l16, l32)


def test_c_abi_dtype_argument_is_checked(dev):
    import ctypes
    from dual_dmp_amd import _lib
    L = _lib.lib()
    y = torch.zeros(64, 32, device=dev)
    s = torch.zeros(64, dtype=torch.float64, device=dev)
    ws = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
    st = L.ddmp_bn_stats(ctypes.c_void_p(y.data_ptr()), 32, 64, 32, 7, ctypes.c_void_p(s.data_ptr()),
                         ctypes.c_void_p(ws.data_ptr()), ws.numel(), None)
    assert st == -1
