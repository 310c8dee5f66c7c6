"""Parity of every HIP kernel (called through the C ABI) against float64 PyTorch-CPU restatements.

Tolerances (stated per test): the kernels compute in exact f32 (f32 FMA / f32-input MFMA), statistics in
f64; against a float64 reference the expected error is f32 round-off: rel-L2 <= 1e-5 per op.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def graphs(dev):
    from dual_dmp_amd import synth, ops
    from dual_dmp_amd.mesh import Mesh
    out = {}
    for name, (v, f) in {"ico3": synth.icosphere(3), "grid": synth.open_grid(9, 7)}.items():
        v, f = synth.permute_vertices(v, f, 1)
        m = Mesh(vs=v, faces=f)
        e = torch.tensor(m.edges.T, dtype=torch.long)
        ei = torch.cat([e, e[[1, 0]]], 1)
        fi = torch.from_numpy(m.f_edges)
        out[name + "_v"] = (ei, len(v))
        out[name + "_f"] = (fi, len(f))
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


def test_graph_tables_match_gcn_norm(dev, graphs, oracle):
    from dual_dmp_amd import ops
    ei, n = graphs["grid_v"]
    rowptr, col, dinv = ops.csr_build_host(ei.numpy(), n)
    row, colo, w = oracle.gcn_norm(ei, n, torch.float64)
    deg = torch.zeros(n, dtype=torch.float64).scatter_add_(0, colo, torch.ones_like(w))
    np.testing.assert_allclose(dinv, deg.pow(-0.5).numpy(), rtol=1e-7)
    assert rowptr[-1] == ei.shape[1] + n
    g = ops.graph_for(ei.to(dev), n)
    assert (g.n_rows, g.n_cols, g.nnz) == (n, n, ei.shape[1] + n)


@pytest.mark.parametrize("C", [8, 16, 32, 64, 128, 256, 512, 3, 20])
@pytest.mark.parametrize("gname", ["ico3_v", "grid_f"])
def test_spmm_matches_dense(dev, graphs, C, gname):
    from dual_dmp_amd import ops
    ei, n = graphs[gname]
    torch.manual_seed(C)
    x = torch.randn(n, C)
    bias = torch.randn(C)
    a, b = torch.rand(C) + 0.5, torch.randn(C)
    A = dense_ahat(ei, n)
    g = ops.graph_for(ei.to(dev), n)
    y = ops.spmm(g, x.to(dev))
    assert relerr(y, A @ x.double()) < 1e-6
    y = ops.spmm(g, x.to(dev), bias=bias.to(dev), pro=(a.to(dev), b.to(dev)))
    ref = A @ f_ref(x.double(), a.double(), b.double()) + bias.double()
    assert relerr(y, ref) < 1e-6


def test_spmm_lds_patch_route_on_a_large_face_graph(dev):
    """Round 4: the LDS-patch gather (csrc/spmm_patch.hip: distinct rows of a 64-row chunk copied to LDS once by DMA, gathers
    from LDS, the entries' patch offsets and weights in registers) is part of the default library and selected for float32
    features on graphs with <= 8 entries per row from 64k rows at C >= 256 (plain) / C = 256 (prologue).  144,400-face torus,
    FACE graph (4 entries per row: NE = 4) and VERTEX graph (7 entries: NE = 8): the route is asserted, results against a
    float64 sparse product and -- same sums in the same order -- bit for bit against the lean gather's; fused reductions and
    the narrow widths stay on the lean gather.  Round 5: the prologue at C = 512, the fused reduction at C = 512 and the
    statistics form (C >= 256) take this kernel too."""
    from dual_dmp_amd import ops, synth, _lib
    from dual_dmp_amd.mesh import Mesh
    if os.environ.get("DDMP_SPMM_PATCH") == "0":
        pytest.skip("LDS-patch gather switched off")
    v, f = synth.rcb_relabel(*synth.torus(380, 190))             # (the engines relabel by coordinate bisection: compact patches)
    m = Mesh(vs=v, faces=f)
    fi = torch.from_numpy(m.f_edges)
    e = torch.tensor(m.edges.T, dtype=torch.long)
    vi = torch.cat([e, e[[1, 0]]], 1)
    L = _lib.lib()
    for ei, n in ((fi, len(f)), (vi, len(v))):
        g = ops.graph_for(ei.to(dev), n)
        sel = L.ddmp_spmm_patch_selected(g._h, 256, 0, 0, 0) == 1       # (144,400 faces / 72,200 vertices: both from 64k rows)
        if os.environ.get("DDMP_SPMM_PATCH") is None:
            # round 6: the chunks of a regular mesh whose patch exceeds the LDS buffers are walked as two 32-row halves inside the same
            # launch (this mesh has some, so the comparisons below cover that path): no heavy list, no second launch per aggregation
            import ctypes
            kd, nh, ns = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            assert L.ddmp_graph_patch_info(g._h, ctypes.byref(kd), ctypes.byref(nh), ctypes.byref(ns)) == 0
            assert kd.value in (3, 4) and nh.value == 0 and ns.value > 0, (kd.value, nh.value, ns.value)
        assert sel and L.ddmp_spmm_patch_selected(g._h, 256, 0, 1, 0) == 1 and L.ddmp_spmm_patch_selected(g._h, 512, 0, 0, 0) == 1
        if os.environ.get("DDMP_SPMM_PATCH") is None:
            # round 5: also the prologue at C = 512, the fused reduction at C = 512 and the statistics form from C = 256;
            # round 6 (one-round-trip chunk set-up): from C = 128, nothing at C = 64; bfloat16: not its fused reduction, as before
            assert L.ddmp_spmm_patch_selected(g._h, 512, 0, 1, 0) == 1 and L.ddmp_spmm_patch_selected(g._h, 128, 0, 0, 0) == 1
            assert L.ddmp_spmm_patch_selected(g._h, 128, 0, 0, 1) == 1 and L.ddmp_spmm_patch_selected(g._h, 128, 0, 1, 2) == 1
            assert L.ddmp_spmm_patch_selected(g._h, 64, 0, 0, 0) == 0 and L.ddmp_spmm_patch_selected(g._h, 64, 0, 0, 1) == 0
            assert L.ddmp_spmm_patch_selected(g._h, 128, 1, 0, 0) == 1 and L.ddmp_spmm_patch_selected(g._h, 256, 1, 0, 1) == 0     # bf16
            assert L.ddmp_spmm_patch_selected(g._h, 256, 1, 0, 0) == 1 and L.ddmp_spmm_patch_selected(g._h, 256, 1, 1, 2) == 1    # bf16
            assert L.ddmp_spmm_patch_selected(g._h, 256, 0, 0, 1) == 1 and L.ddmp_spmm_patch_selected(g._h, 256, 0, 0, 2) == 1
        # float64 reference: D^-1/2 (A + I) D^-1/2 as a sparse matrix
        src = torch.cat([ei[0], torch.arange(n)])
        dst = torch.cat([ei[1], torch.arange(n)])
        deg = torch.zeros(n, dtype=torch.float64).index_add_(0, dst, torch.ones(len(dst), dtype=torch.float64))
        w = deg[src].pow(-0.5) * deg[dst].pow(-0.5)
        A = torch.sparse_coo_tensor(torch.stack([dst, src]), w, (n, n)).coalesce()
        for C in (128, 256, 512):
            torch.manual_seed(C)
            x = torch.randn(n, C)
            a, b, bias = torch.rand(C) + 0.5, torch.randn(C), torch.randn(C)
            y = ops.spmm(g, x.to(dev))
            yp = ops.spmm(g, x.to(dev), bias=bias.to(dev), pro=(a.to(dev), b.to(dev)))
            assert relerr(y, torch.sparse.mm(A, x.double())) < 1e-6
            assert relerr(yp, torch.sparse.mm(A, f_ref(x.double(), a.double(), b.double())) + bias.double()) < 1e-6
            # the fused-reduction form runs the lean gather: its output is the same sums in the same order
            if sel:
                out = torch.empty_like(y)
                bn4 = torch.stack([torch.rand(C) + 0.5, torch.randn(C), torch.randn(C), torch.rand(C) + 0.5]).to(dev)
                ops.spmm_bnred(g, x.to(dev), out, torch.randn(n, C, device=dev), bn4, torch.zeros(2 * C, dtype=torch.float64, device=dev))
                assert torch.equal(out, y)
                # round 6: the BatchNorm backward on the gather takes the LDS-patch kernel too (two tensors per buffer): the
                # same sums in the same order as bn_bwd_apply followed by the plain gather
                if os.environ.get("DDMP_SPMM_PATCH") is None:
                    assert L.ddmp_spmm_patch_selected(g._h, C, 0, 1, 3) == 1 and L.ddmp_spmm_patch_selected(g._h, 64, 0, 1, 3) == 0
                dz, yb = torch.randn(n, C, device=dev), torch.randn(n, C, device=dev) * 2 + 0.5
                c10 = torch.stack([torch.randn(C) * 0.1, torch.randn(C) * 0.1]).to(dev)
                ops.spmm_bnbwd(g, dz, yb, bn4, c10, out)
                dy = torch.empty_like(dz)
                ops.bn_bwd_apply(dz, yb, bn4, c10, dy, torch.empty(2 * C, dtype=torch.float64, device=dev))
                assert torch.equal(out, ops.spmm(g, dy))
                # ... and with bfloat16 features (same form of the kernel; against float64 to the rounding of the stored output)
                zb, yb16 = dz.to(torch.bfloat16), yb.to(torch.bfloat16)
                o16 = torch.empty(n, C, dtype=torch.bfloat16, device=dev)
                ops.spmm_bnbwd(g, zb, yb16, bn4, c10, o16)
                a_, b_, k1, k0 = (t.double().cpu() for t in (bn4[0], bn4[1], c10[0], c10[1]))
                yd, zd = yb16.double().cpu(), zb.double().cpu()
                dyr = a_ * zd * torch.where(yd * a_ + b_ > 0, 1.0, 0.01) + k1 * yd + k0
                assert relerr(o16, torch.sparse.mm(A, dyr)) < 3e-3
                if os.environ.get("DDMP_SPMM_PATCH") is None:
                    assert L.ddmp_spmm_patch_selected(g._h, C, 1, 1, 3) == 1


@pytest.mark.parametrize("C", [32, 256, 8])
def test_spmm_with_bn_backward_reduce(dev, graphs, C):
    """SpMM + the BatchNorm-backward column reductions of its output from one kernel == the two separate calls."""
    from dual_dmp_amd import ops
    for gname in graphs:
        ei, n = graphs[gname]
        torch.manual_seed(C + n)
        x, yp = torch.randn(n, C), torch.randn(n, C) * 2 + 0.3
        bn4 = torch.stack([torch.rand(C) + 0.5, torch.randn(C), torch.randn(C), torch.rand(C) + 0.5]).to(dev)
        g = ops.graph_for(ei.to(dev), n)
        ref_y = ops.spmm(g, x.to(dev))
        ref_s = ops.bn_bwd_reduce(ref_y, yp.to(dev), bn4)
        out = torch.empty_like(ref_y)
        sums = torch.zeros(2 * C, dtype=torch.float64, device=dev)
        ops.spmm_bnred(g, x.to(dev), out, yp.to(dev), bn4, sums)
        assert torch.equal(out, ref_y)
        assert relerr(sums, ref_s) < 1e-5, (gname, relerr(sums, ref_s))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C", [32, 256, 64, 16])
def test_spmm_with_forward_statistics(dev, graphs, C, dtype):
    """SpMM + the BatchNorm statistics of its output from one kernel (transform-first layers, forward): the output is the
    plain kernel's bit for bit; the statistics are summed around a per-column reference in float32 over 16 rows, float64
    from there on -- with a reference near the mean (what the engine passes: the previous iteration's batch mean) the
    variance keeps float32-class accuracy even when |mean| >> std; C = 16 takes the unfused route (exact)."""
    from dual_dmp_amd import ops
    for gname in graphs:
        ei, n = graphs[gname]
        torch.manual_seed(C + n)
        g = ops.graph_for(ei.to(dev), n)
        bias = (torch.randn(C) * 50).to(dev)                       # column means far from 0: |mean| / std up to ~100
        x = torch.randn(n, C, device=dev).to(dtype)
        ref_y = ops.spmm(g, x, bias=bias)
        s_ref = ops.bn_stats(ref_y)
        mean = (s_ref[:C] / n).float()
        var_ref = (s_ref[C:] / n - (s_ref[:C] / n) ** 2)
        for ref, tol_var in ((mean * 1.01, 2e-5), (torch.zeros(C, device=dev), None)):
            out = torch.empty_like(ref_y)
            sums = torch.zeros(2 * C, dtype=torch.float64, device=dev)
            ops.spmm_stats(g, x, out, ref.contiguous(), sums, bias=bias)
            assert torch.equal(out, ref_y)
            assert relerr(sums, s_ref) < 2e-6, (gname, relerr(sums, s_ref))
            var = sums[C:] / n - (sums[:C] / n) ** 2
            if tol_var is not None:
                assert float(((var - var_ref).abs() / var_ref).max()) < tol_var, (gname, float(((var - var_ref).abs() / var_ref).max()))
        # with the tail-fused coefficients on top
        bn4, run = torch.zeros(4, C, device=dev), torch.stack([torch.zeros(C), torch.ones(C)]).to(dev)
        gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        ops.spmm_stats(g, x, out, (mean * 1.01).contiguous(), sums, bias=bias,
                       bn=ops.BnFwd(n, gamma, beta, bn4, running=(run[0], run[1])))
        want = torch.zeros(4, C, device=dev)
        ops.bn_prepare(sums, n, gamma, beta, want)
        assert torch.equal(bn4, want)


@pytest.mark.parametrize("C", [32, 256, 64])
def test_spmm_with_bn_backward_on_the_gather(dev, graphs, C):
    """A_hat . dY with dY = BatchNorm+LeakyReLU backward of (dZ, Y) rebuilt on the gather == bn_bwd_apply followed by
    the plain SpMM, bit for bit (same per-element arithmetic, same accumulation order), and == the float64 formula."""
    from dual_dmp_amd import ops
    ei, n = graphs["grid_f"]
    g = ops.graph_for(ei.to(dev), n)
    torch.manual_seed(C)
    dz, yb = torch.randn(n, C, device=dev), torch.randn(n, C, device=dev) * 2 + 0.5
    bn4 = torch.stack([torch.rand(C) + 0.5, torch.randn(C), torch.randn(C), torch.rand(C) + 0.5]).to(dev)
    c10 = torch.stack([torch.randn(C) * 0.1, torch.randn(C) * 0.1]).to(dev)
    assert ops.spmm_bnbwd_supported(C)
    out = torch.empty(n, C, device=dev)
    ops.spmm_bnbwd(g, dz, yb, bn4, c10, out)
    dy = torch.empty_like(dz)
    sums = torch.empty(2 * C, dtype=torch.float64, device=dev)
    ops.bn_bwd_apply(dz, yb, bn4, c10, dy, sums)
    assert torch.equal(out, ops.spmm(g, dy))
    a, b, k1, k0 = (t.double().cpu() for t in (bn4[0], bn4[1], c10[0], c10[1]))
    z = yb.double().cpu() * a + b
    dy_ref = a * dz.double().cpu() * torch.where(z > 0, 1.0, 0.01) + k1 * yb.double().cpu() + k0
    assert relerr(out, dense_ahat(ei, n) @ dy_ref) < 1e-6


def test_spmm_multi_edges_and_self_loops(dev):
    from dual_dmp_amd import ops
    ei = torch.tensor([[0, 1, 1, 2, 2, 0, 1, 3, 3], [1, 0, 2, 1, 0, 2, 0, 3, 3]])
    n = 5                                               # node 4 isolated, node 3 only explicit self loops
    x = torch.randn(n, 32)
    y = ops.spmm(ops.graph_for(ei.to(dev), n), x.to(dev))
    keep = ei[:, ei[0] != ei[1]]
    assert relerr(y, dense_ahat(keep, n) @ x.double()) < 1e-6


@pytest.fixture(params=[6, 3, 0, 13], ids=["bf16x6", "bf16x3", "f32mfma", "f16x3"])
def gemm_mode(request):
    """Every GEMM arithmetic of the library: the bf16x6 split MFMA (f32-class accuracy), bf16x3 (three products,
    ~2^-16 per product), the f32-input MFMA kernels, and f16x3 (scaled operands, two f16 terms, three products:
    f32-class accuracy) in the row-panel kernels with bf16x6 elsewhere."""
    from dual_dmp_amd import ops
    old = ops.get_gemm_mode()
    ops.set_gemm_mode(request.param)
    yield request.param
    ops.set_gemm_mode(old)


GEMM_TOL = {6: 2e-6, 3: 3e-5, 0: 2e-6, 13: 2e-6}


@pytest.mark.parametrize("n,K,M", [(1000, 32, 64), (777, 512, 512), (130, 8, 32), (513, 256, 128),
                                   (300, 16, 32), (2000, 64, 32), (129, 128, 256), (50, 32, 3),
                                   # row-panel kernels: padded column panels (M < 128*WC), short K, ragged row tiles,
                                   # and more row tiles than CUs (the persistent loop crosses tile boundaries)
                                   (21000, 96, 384), (20300, 64, 260), (66500, 320, 384), (70001, 64, 256),
                                   # narrow-output row panels (512-row blocks): M <= 32 | 64 | 128, K = 32 .. 256
                                   (20500, 32, 64), (33000, 64, 32), (21001, 256, 128), (25000, 128, 20), (20100, 64, 100),
                                   (40000, 256, 512),
                                   # wgrad panels of 256 x 128 and 128 x 256 (round 4: the 128 <-> 256 layers), ragged row counts
                                   (40001, 128, 256), (66500, 256, 128), (40003, 128, 320), (40005, 384, 128)])     # (+ partial panels)
def test_gemm_nt_nn_tn(dev, gemm_mode, n, K, M):
    from dual_dmp_amd import ops
    tol = GEMM_TOL[gemm_mode]
    torch.manual_seed(n + K + M)
    a, w, bias = torch.randn(n, K), torch.randn(M, K) / K ** 0.5, torch.randn(M)
    sc, sh = torch.rand(K) + 0.5, torch.randn(K)
    y = ops.gemm_nt(a.to(dev), w.to(dev), bias=bias.to(dev))
    assert relerr(y, a.double() @ w.double().t() + bias.double()) < tol
    y = ops.gemm_nt(a.to(dev), w.to(dev), pro=(sc.to(dev), sh.to(dev)))
    assert relerr(y, f_ref(a.double(), sc.double(), sh.double()) @ w.double().t()) < tol
    if M % 4 == 0:
        g = torch.randn(n, M)
        dx = ops.gemm_nn(g.to(dev), w.to(dev))
        assert relerr(dx, g.double() @ w.double()) < tol
        dw = ops.gemm_tn(g.to(dev), a.to(dev))
        assert relerr(dw, g.double().t() @ a.double()) < tol
        dw = ops.gemm_tn(g.to(dev), a.to(dev), pro=(sc.to(dev), sh.to(dev)))
        assert relerr(dw, g.double().t() @ f_ref(a.double(), sc.double(), sh.double())) < tol


@pytest.mark.parametrize("n,cin,cout", [(65537, 256, 512), (66000, 512, 512), (70001, 256, 256),
                                        (65999, 128, 256), (70003, 64, 128), (66001, 32, 64)])     # + narrow aggregate-first layers
def test_gemm_bnbwd_fused_matches_composition(dev, n, cin, cout):
    """dgrad / wgrad with the BatchNorm+LeakyReLU backward folded into the operand load == bn_bwd_apply followed by
    the plain GEMMs (same arithmetic per element, so only the GEMM rounding differs) and == the float64 formula."""
    from dual_dmp_amd import ops
    if not ops.gemm_bnbwd_supported(cout, cin, n):
        pytest.skip("fused kernels not available in this GEMM mode")
    torch.manual_seed(n + cin)
    dz, yb = torch.randn(n, cout), torch.randn(n, cout) * 2 + 0.5
    w, p = torch.randn(cout, cin) / cout ** 0.5, torch.randn(n, cin)
    bn4 = torch.stack([torch.rand(cout) + 0.5, torch.randn(cout), torch.randn(cout), torch.rand(cout) + 0.5])
    c10 = torch.stack([torch.randn(cout) * 0.1, torch.randn(cout) * 0.1])
    a, b, k1, k0 = bn4[0].double(), bn4[1].double(), c10[0].double(), c10[1].double()
    z = yb.double() * a + b
    dy_ref = a * dz.double() * torch.where(z > 0, 1.0, 0.01) + k1 * yb.double() + k0
    dzg, ybg, wg, pg, bn4g, c10g = (t.to(dev) for t in (dz, yb, w, p, bn4, c10))
    dx = ops.gemm_nn_bnbwd(dzg, ybg, wg, bn4g, c10g)
    assert relerr(dx, dy_ref @ w.double()) < 3e-6
    dw = ops.gemm_tn_bnbwd(dzg, ybg, pg, bn4g, c10g)
    assert relerr(dw, dy_ref.t() @ p.double()) < 3e-6
    sc, sh = (torch.rand(cin) + 0.5).to(dev), torch.randn(cin).to(dev)
    dw = ops.gemm_tn_bnbwd(dzg, ybg, pg, bn4g, c10g, pro=(sc, sh))
    assert relerr(dw, dy_ref.t() @ f_ref(p.double(), sc.cpu().double(), sh.cpu().double())) < 3e-6
    # against the unfused composition on the device
    dy = torch.empty_like(dzg)
    sums = torch.empty(2 * cout, dtype=torch.float64, device=dev)
    ops.bn_bwd_apply(dzg, ybg, bn4g, c10g, dy, sums)
    assert relerr(dy, dy_ref) < 1e-6
    assert relerr(dx, ops.gemm_nn(dy, wg).double().cpu()) < 3e-6


@pytest.mark.parametrize("n,M,K", [(65999, 128, 256), (40003, 256, 128), (33001, 128, 512)])
def test_wgrad_narrow_panels_with_bn_backward_on_the_load(dev, n, M, K):
    """Round 4: the f16x3 wgrad kernel's 256 x 128 / 128 x 256 panels in their three-stream form -- dW = dY^T . f(Z) with dY the
    BatchNorm+LeakyReLU backward of (dZ, Yb) rebuilt on the load -- against float64, with and without the prologue on Z, at row
    counts that leave a ragged last stage and a ragged last split."""
    from dual_dmp_amd import ops
    if not ops.gemm_tn_bnbwd_supported(M, K, n):
        pytest.skip("fused wgrad not available in this GEMM mode")
    torch.manual_seed(n + M)
    dz, yb, z = torch.randn(n, M), torch.randn(n, M) * 2 + 0.5, torch.randn(n, K)
    bn4 = torch.stack([torch.rand(M) + 0.5, torch.randn(M), torch.randn(M), torch.rand(M) + 0.5])
    c10 = torch.stack([torch.randn(M) * 0.1, torch.randn(M) * 0.1])
    a, b, k1, k0 = bn4[0].double(), bn4[1].double(), c10[0].double(), c10[1].double()
    dy_ref = a * dz.double() * torch.where(yb.double() * a + b > 0, 1.0, 0.01) + k1 * yb.double() + k0
    dzg, ybg, zg, bn4g, c10g = (t.to(dev) for t in (dz, yb, z, bn4, c10))
    dw = ops.gemm_tn_bnbwd(dzg, ybg, zg, bn4g, c10g)
    assert relerr(dw, dy_ref.t() @ z.double()) < 3e-6
    sc, sh = (torch.rand(K) + 0.5).to(dev), torch.randn(K).to(dev)
    dw = ops.gemm_tn_bnbwd(dzg, ybg, zg, bn4g, c10g, pro=(sc, sh))
    assert relerr(dw, dy_ref.t() @ f_ref(z.double(), sc.cpu().double(), sh.cpu().double())) < 3e-6


@pytest.mark.parametrize("cin", [16, 8])
def test_first_layer_wgrad_with_bn_backward_on_the_load(dev, cin):
    """Layer 0 has no dgrad: its dY (BatchNorm backward of (dZ, Y), 32 columns) feeds the wgrad only and is rebuilt there."""
    from dual_dmp_amd import ops
    n, cout = 70001, 32
    if not ops.gemm_tn_bnbwd_supported(cout, cin, n):
        pytest.skip("fused wgrad not available in this GEMM mode")
    torch.manual_seed(cin)
    dz, yb, p = torch.randn(n, cout), torch.randn(n, cout) * 2 + 0.5, torch.randn(n, cin)
    bn4 = torch.stack([torch.rand(cout) + 0.5, torch.randn(cout), torch.randn(cout), torch.rand(cout) + 0.5])
    c10 = torch.stack([torch.randn(cout) * 0.1, torch.randn(cout) * 0.1])
    a, b, k1, k0 = bn4[0].double(), bn4[1].double(), c10[0].double(), c10[1].double()
    z = yb.double() * a + b
    dy_ref = a * dz.double() * torch.where(z > 0, 1.0, 0.01) + k1 * yb.double() + k0
    dw = ops.gemm_tn_bnbwd(dz.to(dev), yb.to(dev), p.to(dev), bn4.to(dev), c10.to(dev))
    assert relerr(dw, dy_ref.t() @ p.double()) < 3e-6


def _rr_shapes(seed, count):
    """Random shapes inside the row-register kernel's domain: >= 20k rows (ragged last tiles, fewer tiles than workgroup slots
    and many more), contraction 64..512 in steps of 32, 129..512 output columns in steps of 4 (padded 256 / 512 panels)."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        n = int(rng.choice([20000 + int(rng.integers(0, 4000)), 33000 + int(rng.integers(0, 300)), 70000 + int(rng.integers(0, 9000))]))
        out.append((n, int(rng.integers(2, 17)) * 32, int(rng.integers(33, 129)) * 4))
    return out


@pytest.mark.parametrize("n,K,M", _rr_shapes(3, 10))
def test_row_register_gemm_random_shapes(dev, f16x3, n, K, M):
    """Forward (bias; prologue; statistics), dgrad and the BatchNorm-backward dgrad of the row-register f16x3 kernel on random
    shapes against float64, each launched twice (bit-stable)."""
    from dual_dmp_amd import ops
    torch.manual_seed(n + K + M)
    a, w, bias = torch.randn(n, K), torch.randn(M, K) / K ** 0.5, torch.randn(M)
    sc, sh = torch.rand(K) + 0.5, torch.randn(K)
    ad, wd = a.to(dev), w.to(dev)
    ref = a.double() @ w.double().t()
    y = ops.gemm_nt(ad, wd, bias=bias.to(dev))
    assert relerr(y, ref + bias.double()) < 2e-6
    assert torch.equal(ops.gemm_nt(ad, wd, bias=bias.to(dev)), y)
    sums = torch.zeros(2 * M, dtype=torch.float64, device=dev)
    y2 = ops.gemm_nt_stats(ad, wd, sums, pro=(sc.to(dev), sh.to(dev)))
    assert relerr(y2, f_ref(a.double(), sc.double(), sh.double()) @ w.double().t()) < 2e-6
    yd = y2.double().cpu()
    assert relerr(sums[:M], yd.sum(0)) < 1e-6 and relerr(sums[M:], (yd * yd).sum(0)) < 1e-6
    if M % 32 == 0 and K > 128:                                      # as a dgrad: contraction M, output K
        g = torch.randn(n, M)
        dx = ops.gemm_nn(g.to(dev), w.to(dev))
        assert relerr(dx, g.double() @ w.double()) < 2e-6
        if ops.gemm_bnbwd_supported(M, K, n):
            yb = torch.randn(n, M) * 2 + 0.5
            bn4 = torch.stack([torch.rand(M) + 0.5, torch.randn(M), torch.randn(M), torch.rand(M) + 0.5])
            c10 = torch.stack([torch.randn(M) * 0.1, torch.randn(M) * 0.1])
            a4, b4, k1, k0 = bn4[0].double(), bn4[1].double(), c10[0].double(), c10[1].double()
            z = yb.double() * a4 + b4
            dy = a4 * g.double() * torch.where(z > 0, 1.0, 0.01) + k1 * yb.double() + k0
            dx = ops.gemm_nn_bnbwd(g.to(dev), yb.to(dev), w.to(dev), bn4.to(dev), c10.to(dev))
            assert relerr(dx, dy @ w.double()) < 3e-6


@pytest.mark.parametrize("n,M,K", [(66001, 256, 512), (40000, 128, 256), (33000, 512, 512), (20300, 64, 256),
                                   (70001, 64, 128), (66003, 32, 64), (25000, 256, 128)])      # + narrow outputs (row-panel kernel)
def test_gemm_nn_with_bn_backward_reductions(dev, n, M, K):
    """dgrad of a transform-first layer with the next BatchNorm-backward column reductions from its epilogue (row-register
    kernel) == gemm_nn followed by bn_bwd_reduce of its output, and == the float64 formulas."""
    from dual_dmp_amd import ops
    if not ops.gemm_nn_bnred_supported(M, K, n):
        pytest.skip("row-register f16x3 kernels not active in this configuration")
    torch.manual_seed(n + M + K)
    dh, w = torch.randn(n, M) * 1e-2, torch.randn(M, K) / M ** 0.5
    yp = torch.randn(n, K) * 2 + 0.3
    bn4 = torch.stack([torch.rand(K) + 0.5, torch.randn(K), torch.randn(K) * 0.1 + 0.3, torch.rand(K) + 0.5])
    dhg, wg, ypg, bn4g = dh.to(dev), w.to(dev), yp.to(dev), bn4.to(dev)
    sums = torch.zeros(2 * K, dtype=torch.float64, device=dev)
    out = ops.gemm_nn_bnred(dhg, wg, ypg, bn4g, sums)
    ref = dh.double() @ w.double()
    assert relerr(out, ref) < 2e-6
    assert torch.equal(out, ops.gemm_nn(dhg, wg))                                   # the same kernel without the epilogue
    sums_pass = torch.zeros(2 * K, dtype=torch.float64, device=dev)
    ops.bn_bwd_reduce(out, ypg, bn4g, sums2=sums_pass)
    assert relerr(sums, sums_pass) < 2e-6, relerr(sums, sums_pass)                  # float32 over 16 rows, float64 from there on
    a, b, mu, rs = (bn4[i].double() for i in range(4))
    g = out.double().cpu() * torch.where(yp.double() * a + b > 0, 1.0, 0.01)
    ref_s = torch.cat([g.sum(0), (g * (yp.double() - mu) * rs).sum(0)])
    assert relerr(sums, ref_s) < 1e-5


@pytest.mark.parametrize("n,K,M", [(20777, 512, 512), (70001, 64, 256), (21000, 96, 384), (513, 256, 128), (130, 8, 32),
                                   (30001, 64, 128), (20480, 32, 64), (22222, 128, 32)])
def test_gemm_nt_stats_matches_bn_stats(dev, gemm_mode, n, K, M):
    """Statistics from the GEMM epilogue (panel shapes) or from the fallback pass == bn_stats of the stored output."""
    from dual_dmp_amd import ops
    if M & (M - 1) and (gemm_mode == 0 or os.environ.get("DDMP_GEMM_PANEL") == "0"):
        pytest.skip("the fallback pass (ddmp_bn_stats_f32) takes power-of-two widths only")
    torch.manual_seed(n + M)
    a, w, bias = torch.randn(n, K) + 0.3, torch.randn(M, K) / K ** 0.5, torch.randn(M)
    sums = torch.zeros(2 * M, dtype=torch.float64, device=dev)
    y = ops.gemm_nt_stats(a.to(dev), w.to(dev), sums, bias=bias.to(dev))
    ref = a.double() @ w.double().t() + bias.double()
    assert relerr(y, ref) < GEMM_TOL[gemm_mode]
    yd = y.double().cpu()
    assert relerr(sums[:M], yd.sum(0)) < 1e-6
    assert relerr(sums[M:], (yd * yd).sum(0)) < 1e-6
    # a column mean 1e3 times the spread (mean^2 / var = 1e6: float32 partial sums would be off by ~10 %): the variance is a difference of the two sums, which must therefore be (nearly)
    # exact sums of the stored float32 values -- float64 partials in the epilogue, as in the separate pass
    y = ops.gemm_nt_stats(a.to(dev), (w * 0.1).to(dev), sums, bias=(bias * 0 + 100.0).to(dev))
    yd = y.double().cpu()
    var_ref = yd.var(0, unbiased=False)
    var = sums[M:].cpu() / n - (sums[:M].cpu() / n) ** 2
    assert float(((var - var_ref).abs() / var_ref).max()) < 1e-6


@pytest.fixture
def f16x3():
    from dual_dmp_amd import ops
    if os.environ.get("DDMP_GEMM_PANEL") == "0":
        pytest.skip("f16x3 lives in the row-panel kernels, which are disabled")
    old = ops.get_gemm_mode()
    ops.set_gemm_mode(13)
    yield
    ops.set_gemm_mode(old)


@pytest.mark.parametrize("scale", [1.0, 1e-20, 1e20, 1e-30])
def test_f16x3_operand_scaling(dev, f16x3, scale):
    """f16 has 5 exponent bits: the f16x3 kernels scale every operand by a power of two derived from its absolute
    maximum (measured in a pre-pass when the caller names no scale slots).  Operands of any float32 magnitude, with
    columns and rows spread over many orders of magnitude, come out at float32-class accuracy (norm-wise)."""
    from dual_dmp_amd import ops
    n, K, M = 40000, 256, 512
    torch.manual_seed(5)
    a = torch.randn(n, K) * scale
    a[:, : K // 4] *= 1e-3                                          # quiet columns
    a[: n // 2] *= 0.02                                             # quiet rows
    w = torch.randn(M, K) / K ** 0.5
    g0 = torch.randn(n, M) * 1e-4 * torch.rand(n, 1) ** 4           # gradient-like: few loud rows
    g, gt = g0 * scale, g0 / scale                                  # (the wgrad's product stays in float32 range)
    ag, wg, gg, gtg = a.to(dev), w.to(dev), g.to(dev), gt.to(dev)
    assert relerr(ops.gemm_nt(ag, wg), a.double() @ w.double().t()) < 2e-6
    assert relerr(ops.gemm_nn(gg, wg), g.double() @ w.double()) < 2e-6
    assert relerr(ops.gemm_tn(gtg, ag), gt.double().t() @ a.double()) < 3e-6
    z = torch.zeros(n, K, device=dev)                               # all-zero operand: scale 1, exact zeros
    assert float(ops.gemm_nt(z, wg).abs().max()) == 0.0


def _f16x3_elementwise_bound(a, w, target):
    """Per-element error bound of the f16x3 product sum_k a_k w_k (float64 tensors [n,K], [M,K]) from the scheme itself
    (csrc/gemm_f16s.inc): an operand x is scaled by the power of two s that puts max|x| in [2^(target-1), 2^target) and kept
    as h1 + h2 (two f16 terms: 22 bits, but never finer than the f16 subnormal step 2^-24 -- rounding error <= 2^-25 / s
    in the operand's own units), the product h2.h2' (<= 2^-22 |a||w|) is dropped, accumulation is float32:
        |err| <= sum_k ( |a_k| ew_k + |w_k| ea_k + ea_k ew_k + 2^-22 |a_k||w_k| ) + 2^-23 sqrt(K) sum_k |a_k||w_k|
        e x_k  = max(2^-23 |x_k|, 2^-25 / s_x)
    (the last term: float32 accumulation, random-walk estimate; the caller applies a factor 2 of safety)."""
    def eps(x):
        amax = float(x.abs().max())
        e = np.floor(np.log2(amax)) + 1                    # amax in [2^(e-1), 2^e)
        s = 2.0 ** (target - e)
        return torch.maximum(x.abs() * 2.0 ** -23, torch.full_like(x, 2.0 ** -25 / s))
    ea, ew = eps(a), eps(w)
    aa, ww = a.abs(), w.abs()
    K = a.shape[1]
    return aa @ ew.t() + ea @ ww.t() + ea @ ew.t() + (2.0 ** -22 + 2.0 ** -23 * K ** 0.5) * (aa @ ww.t())


def test_f16x3_per_row_and_per_column_error(dev, f16x3):
    """The f16x3 split is fixed-point relative to the operand's GLOBAL maximum, so rel-L2 over the whole result says
    nothing about quiet rows / columns (round-2 verdict).  Here the rows of A span 1e-6 .. 1 of its maximum and the
    rows of W (= output columns) span 1e-4 .. 1: every output ELEMENT must sit inside the bound that follows from the
    scheme (two f16 terms of the scaled operand, dropped h2.h2, float32 accumulation), with the exact pre-pass scale
    (target 2^15) and with a one-iteration-old scale slot (target 2^10); and every ROW and every COLUMN must be
    float32-class in its own norm down to 2^-13 of the maximum (1e-5), degrading no faster than the bound says below."""
    from dual_dmp_amd import ops
    n, K, M = 40000, 256, 512
    torch.manual_seed(9)
    rs = 10.0 ** (-6.0 * torch.rand(n, 1, dtype=torch.float64))      # row scales, log-uniform in [1e-6, 1]
    rs[0] = 1.0
    cs = 10.0 ** (-4.0 * torch.rand(M, 1, dtype=torch.float64))
    cs[0] = 1.0
    a = (torch.randn(n, K, dtype=torch.float64) * rs).float().double()
    w = (torch.randn(M, K, dtype=torch.float64) / K ** 0.5 * cs).float().double()
    ref = a @ w.t()
    ag, wg = a.float().to(dev), w.float().to(dev)
    slots = torch.zeros(1, 4, device=dev)
    for target, stale in ((15, False), (10, True)):
        if stale:
            ops.gemm_nt(ag, wg, scales=(slots[0], None, True))
            ops.gemm_scales_roll(slots)
        y = ops.gemm_nt(ag, wg, **({"scales": (slots[0], None, False)} if stale else {})).double().cpu()
        err = (y - ref).abs()
        bound = 2.0 * _f16x3_elementwise_bound(a, w, target) + 1e-300
        worst = float((err / bound).max())
        assert worst <= 1.0, (target, worst)
        row_rel = (err.norm(dim=1) / ref.norm(dim=1)).numpy()
        col_rel = (err.norm(dim=0) / ref.norm(dim=0)).numpy()
        loud_rows = (rs[:, 0] >= (1e-5 if not stale else 1e-3)).numpy()
        assert row_rel[loud_rows].max() < 4e-6, (target, row_rel[loud_rows].max())
        assert col_rel.max() < 4e-6, (target, col_rel.max())          # columns mix loud and quiet rows of A: never quiet
        # the quietest rows (1e-6 of the maximum): what the fixed-point floor leaves -- still ~5 digits with the exact scale
        assert row_rel.max() < (1e-3 if not stale else 4e-2), (target, row_rel.max())
    # the wgrad form (both operands activations, reduction over the rows): per output element against the same model
    g = (torch.randn(n, 256, dtype=torch.float64) * rs * 1e-3).float().double()
    dw = ops.gemm_tn(g.float().to(dev), ag).double().cpu()
    refw = g.t() @ a
    errw = (dw - refw).abs()
    boundw = 2.0 * _f16x3_elementwise_bound(g.t().contiguous(), a.t().contiguous(), 15) + 1e-300
    assert float((errw / boundw).max()) <= 1.0, float((errw / boundw).max())
    assert float(errw.norm() / refw.norm()) < 3e-6


def test_f16x3_scale_slots(dev, f16x3):
    """The training-loop protocol: persistent slots, primed once, then each call uses the maximum recorded by the
    previous iteration's kernels (ddmp_gemm_scales_roll).  Growth within the head-room (x 64) is exact business as
    usual; growth beyond it raises the slot's flag AND is healed on the spot: the same call re-launches the kernel,
    which redoes the product with the maximum it has just measured (gemm_f16s.inc) -- the result the caller sees is the
    float32-class one, the roll counts the event."""
    from dual_dmp_amd import ops
    n, K, M = 40000, 256, 256
    torch.manual_seed(6)
    a, w, g = torch.randn(n, K), torch.randn(M, K) / K ** 0.5, torch.randn(n, M) * 1e-3
    ag, wg, gg = a.to(dev), w.to(dev), g.to(dev)
    slots = torch.zeros(2, 4, device=dev)
    ref = a.double() @ w.double().t()
    assert relerr(ops.gemm_nt(ag, wg, scales=(slots[0], None, True)), ref) < 2e-6
    amax = float(a.abs().max())
    assert float(slots[0, 0]) == amax and float(slots[0, 1]) == amax          # measured, and seen by the kernel
    ops.gemm_scales_roll(slots)
    assert float(slots[0, 0]) == amax and float(slots[0, 1]) == 0.0
    for grow in (1.0, 30.0, 0.01):                                  # stale scale, data moved: still float32-class
        assert relerr(ops.gemm_nt(ag * grow, wg, scales=(slots[0], None, False)), ref * grow) < 2e-6
        assert int(slots[0, 2].view(torch.int32)) == 0
        assert float(slots[0, 1]) == float((a * grow).abs().max())
        slots[0, 1] = 0.0
    # both operands of the wgrad form
    assert relerr(ops.gemm_tn(gg, ag, scales=(slots[1], slots[0], True)), g.double().t() @ a.double()) < 3e-6
    ops.gemm_scales_roll(slots)
    assert relerr(ops.gemm_tn(gg * 5, ag, scales=(slots[1], slots[0], False)), 5 * (g.double().t() @ a.double())) < 3e-6
    assert int(slots[:, 2].view(torch.int32).abs().sum()) == 0
    # beyond the head-room (x 1000 between two iterations): flagged, healed, counted
    ops.gemm_scales_roll(slots)
    y = ops.gemm_nt(ag * 1000.0, wg, scales=(slots[0], None, False))
    assert int(slots[0, 2].view(torch.int32)) == 1
    assert relerr(y, ref * 1000.0) < 2e-6                          # NOT the clamped product
    dw = ops.gemm_tn(gg, ag * 1000.0, scales=(slots[1], slots[0], False))      # the wgrad on the same (outgrown) operand slot
    assert relerr(dw, 1000.0 * (g.double().t() @ a.double())) < 3e-6
    ops.gemm_scales_roll(slots)
    assert int(slots[0, 2].view(torch.int32)) == 0 and float(slots[0, 3]) == 1.0 and float(slots[1, 3]) == 0.0
    assert float(slots[0, 0]) == float((a * 1000.0).abs().max())  # and the next iteration starts from the right scale
    # the dgrad-with-BatchNorm-backward and statistics forms heal the same way
    yb = torch.randn(n, M, device=dev)
    bn4 = torch.stack([torch.rand(M) + 0.5, torch.randn(M), torch.randn(M), torch.rand(M) + 0.5]).to(dev)
    c10 = torch.stack([torch.randn(M) * 0.1, torch.randn(M) * 0.1]).to(dev)
    if ops.gemm_bnbwd_supported(M, K, n):
        slot = torch.zeros(1, 4, device=dev)
        ops.gemm_nn_bnbwd(gg, yb, wg, bn4, c10, scales=(slot[0], None, True))
        ops.gemm_scales_roll(slot)
        big = ops.gemm_nn_bnbwd(gg * 1e6, yb * 1e3, wg, bn4, c10, scales=(slot[0], None, False))
        dy = torch.empty(n, M, device=dev)
        ops.bn_bwd_apply(gg * 1e6, yb * 1e3, bn4, c10, dy, torch.empty(2 * M, dtype=torch.float64, device=dev))
        assert int(slot[0, 2].view(torch.int32)) == 1
        assert relerr(big, dy.double().cpu() @ w.double()) < 3e-6
    # a non-finite operand cannot be healed: the roll marks the slot -1
    slot = torch.zeros(1, 4, device=dev)
    ops.gemm_nt(ag, wg, scales=(slot[0], None, True))
    ops.gemm_scales_roll(slot)
    bad = ag.clone()
    bad[5, 7] = float("inf")
    ops.gemm_nt(bad, wg, scales=(slot[0], None, False))
    ops.gemm_scales_roll(slot)
    assert float(slot[0, 3]) == -1.0


def test_gemm_transpose_detecting(dev, gemm_mode):
    """A = I with an asymmetric W: a swapped C-write would show."""
    from dual_dmp_amd import ops
    n = 128
    a = torch.eye(n)
    w = torch.arange(n * n, dtype=torch.float32).reshape(n, n) / n
    y = ops.gemm_nt(a.to(dev), w.to(dev))
    assert torch.equal(y.cpu(), w.t())
    assert torch.equal(ops.gemm_nn(a.to(dev), w.to(dev)).cpu(), w)
    assert torch.equal(ops.gemm_tn(a.to(dev), w.to(dev)).cpu(), w)


@pytest.mark.parametrize("n,C", [(642, 32), (1280, 512), (100, 8), (3000, 256)])
def test_batchnorm_lrelu_forward_backward(dev, n, C):
    from dual_dmp_amd import ops
    torch.manual_seed(C)
    y = (torch.randn(n, C) * 2 + 3).requires_grad_(True)
    gamma, beta = (torch.rand(C) + 0.5).requires_grad_(True), torch.randn(C).requires_grad_(True)
    dz = torch.randn(n, C)
    bn = torch.nn.BatchNorm1d(C).double()
    with torch.no_grad():
        bn.weight.copy_(gamma)
        bn.bias.copy_(beta)
    yd = y.detach().double().requires_grad_(True)
    z = torch.nn.functional.leaky_relu(bn(yd), 0.01)
    z.backward(dz.double())

    yg = y.detach().to(dev)
    sums = ops.bn_stats(yg)
    bn4 = torch.empty(4, C, device=dev)
    run = torch.stack([torch.zeros(C), torch.ones(C)]).to(dev)
    ops.bn_prepare(sums, n, gamma.detach().to(dev), beta.detach().to(dev), bn4, running=(run[0], run[1]))
    zz = ops.bn_lrelu_apply(yg, bn4[0], bn4[1])
    assert relerr(zz, z) < 2e-6
    assert relerr(run[0], bn.running_mean) < 1e-6 and relerr(run[1], bn.running_var) < 1e-6
    sums2 = ops.bn_bwd_reduce(dz.to(dev), yg, bn4)
    dgamma, dbeta, c10 = torch.empty(C, device=dev), torch.empty(C, device=dev), torch.empty(2, C, device=dev)
    ops.bn_bwd_prepare(sums2, n, bn4, dgamma, dbeta, c10)
    dy = torch.empty(n, C, device=dev)
    dbs = torch.empty(2 * C, dtype=torch.float64, device=dev)
    ops.bn_bwd_apply(dz.to(dev), yg, bn4, c10, dy, dbs)
    assert relerr(dy, yd.grad) < 1e-5
    assert relerr(dgamma, bn.weight.grad) < 1e-5 and relerr(dbeta, bn.bias.grad) < 1e-5
    # conv-bias gradient = column sums of dY: analytically zero after BN
    assert float(dbs[:C].abs().max()) < 1e-3 * float(yd.grad.abs().sum(0).max())
    assert relerr(ops.colsum(dz.to(dev)), dz.double().sum(0)) < 1e-7


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,C", [(1500, 32), (13068, 512), (70001, 256)])
def test_tail_fused_coefficients_are_bitwise_those_of_the_prepare_kernels(dev, graphs, n, C, dtype):
    """The per-call option ``bn=`` (DDMP_OPT_BN_FWD / DDMP_OPT_BN_BWD of the ``_o`` entry points): the second stage of the call's
    reduction writes the BatchNorm coefficients itself (finalize.h) -- same sums, same coefficients, same running statistics, on
    every reducing entry; a call whose route has no second stage launches the stand-alone kernel by itself."""
    from dual_dmp_amd import ops
    torch.manual_seed(n + C)
    y = (torch.randn(n, C, device=dev) * 2 + 1).to(dtype)
    dz = torch.randn(n, C, device=dev).to(dtype)
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)

    def coeffs(armed, stats):
        bn4 = torch.zeros(4, C, device=dev)
        run = torch.stack([torch.full((C,), 0.25), torch.full((C,), 2.0)]).to(dev)
        sums = stats({"bn": ops.BnFwd(n, gamma, beta, bn4, running=(run[0], run[1]))} if armed else {})
        if not armed:
            ops.bn_prepare(sums, n, gamma, beta, bn4, running=(run[0], run[1]))
        return sums.clone(), bn4, run

    ref = coeffs(False, lambda kw: ops.bn_stats(y, **kw))
    got = coeffs(True, lambda kw: ops.bn_stats(y, **kw))
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    bn4 = ref[1]
    if C >= 64:                                               # statistics from the GEMM epilogue (or GEMM + pass): both routes
        K = 64
        a_ = torch.randn(n, K, device=dev).to(dtype)
        w_ = torch.randn(C, K, device=dev) / 8
        out = torch.empty(n, C, device=dev, dtype=dtype)
        sums = torch.empty(2 * C, dtype=torch.float64, device=dev)
        r2 = coeffs(False, lambda kw: (ops.gemm_nt_stats(a_, w_, sums, out=out, **kw), sums)[1])
        g2 = coeffs(True, lambda kw: (ops.gemm_nt_stats(a_, w_, sums, out=out, **kw), sums)[1])
        for a, b in zip(r2, g2):
            assert torch.equal(a, b)

    def bwd(armed, red):
        dgamma, dbeta, c10 = torch.zeros(C, device=dev), torch.zeros(C, device=dev), torch.zeros(2, C, device=dev)
        sums2 = red({"bn": ops.BnBwd(n, bn4, dgamma, dbeta, c10)} if armed else {})
        if not armed:
            ops.bn_bwd_prepare(sums2, n, bn4, dgamma, dbeta, c10)
        return sums2.clone(), dgamma, dbeta, c10

    for a, b in zip(bwd(False, lambda kw: ops.bn_bwd_reduce(dz, y, bn4, **kw)), bwd(True, lambda kw: ops.bn_bwd_reduce(dz, y, bn4, **kw))):
        assert torch.equal(a, b)
    if n == 1500:                                             # the reductions from the gather's epilogue
        ei, ng = graphs["ico3_f"]
        g = ops.graph_for(ei.to(dev), ng)
        x, yp, o = dz[:ng].contiguous(), y[:ng].contiguous(), torch.empty(ng, C, device=dev, dtype=dtype)
        s2 = torch.empty(2 * C, dtype=torch.float64, device=dev)
        for a, b in zip(bwd(False, lambda kw: (ops.spmm_bnred(g, x, o, yp, bn4, s2, **kw), s2)[1]),
                        bwd(True, lambda kw: (ops.spmm_bnred(g, x, o, yp, bn4, s2, **kw), s2)[1])):
            assert torch.equal(a, b)
    c10 = bwd(True, lambda kw: ops.bn_bwd_reduce(dz, y, bn4, **kw))[3]
    assert ops.next_pending() == 0                            # nothing outlives the call it was passed to
    # dY without its (analytically zero) column sums
    dy0, dy1 = torch.empty_like(y), torch.empty_like(y)
    ops.bn_bwd_apply(dz, y, bn4, c10, dy0, torch.empty(2 * C, dtype=torch.float64, device=dev))
    ops.bn_bwd_apply(dz, y, bn4, c10, dy1, None)
    assert torch.equal(dy0, dy1)


@pytest.mark.parametrize("n", [3000, 26000])
def test_prepared_weight_planes_give_the_same_products(dev, n):
    """ddmp_gemm_prepare_weights: all weight matrices split in two launches into caller-owned plane buffers; the GEMM calls
    that are handed those buffers skip their own split.  Same values bit for bit on every route (f16x3 / bf16x6 row panels
    from 20k rows, wave-specialised and plain kernels below), and a buffer prepared for ANOTHER matrix is not trusted."""
    from dual_dmp_amd import ops
    torch.manual_seed(n)
    shapes = [(512, 512), (256, 512), (512, 256), (128, 256), (64, 128), (32, 64), (32, 16), (256, 128)]     # (M = cout, K = cin)
    ws = [torch.randn(M, K, device=dev) / K ** 0.5 for M, K in shapes]
    items, bufs = [], {}
    for i, w in enumerate(ws):
        for form in (0, 1):
            buf = torch.empty(ops.gemm_rows_workspace_bytes(w.shape[1], w.shape[0]), dtype=torch.uint8, device=dev)
            bufs[(i, form)] = buf
            items.append((w, form, False, buf))
    ops.gemm_prepare_weights(items, n, torch.empty(8 * len(items), device=dev))
    for i, w in enumerate(ws):
        M, K = w.shape
        a = torch.randn(n, K, device=dev)
        g = torch.randn(n, M, device=dev)
        assert torch.equal(ops.gemm_nt(a, w, wplanes=bufs[(i, 0)]), ops.gemm_nt(a, w)), (M, K)
        assert torch.equal(ops.gemm_nn(g, w, wplanes=bufs[(i, 1)]), ops.gemm_nn(g, w)), (M, K)
        if M >= 64:
            s0, s1 = (torch.empty(2 * M, dtype=torch.float64, device=dev) for _ in range(2))
            y0 = ops.gemm_nt_stats(a, w, s0, wplanes=bufs[(i, 0)])
            y1 = ops.gemm_nt_stats(a, w, s1)
            assert torch.equal(y0, y1) and torch.equal(s0, s1)
    # the weights change (an optimizer step) and nobody re-prepares: announcing the stale buffer with ANOTHER matrix of the
    # same shape is detected (the library re-splits); re-preparing makes it current again
    w2 = ws[0] + 1.0
    a = torch.randn(n, 512, device=dev)
    assert torch.equal(ops.gemm_nt(a, w2, wplanes=bufs[(0, 0)]), ops.gemm_nt(a, w2))
    ws[0].add_(0.5)
    ops.gemm_prepare_weights([(ws[0], 0, False, bufs[(0, 0)])], n, torch.empty(8, device=dev))
    assert torch.equal(ops.gemm_nt(a, ws[0], wplanes=bufs[(0, 0)]), ops.gemm_nt(a, ws[0]))


@pytest.mark.parametrize("kind", [0, 1])
def test_heads_forward_backward(dev, kind):
    from dual_dmp_amd import ops
    torch.manual_seed(kind)
    n = 1500
    y = torch.randn(n, 32)
    a, b = torch.rand(32) + 0.5, torch.randn(32)
    W1, b1 = (torch.randn(16, 32) / 5).requires_grad_(True), torch.randn(16).requires_grad_(True)
    W2, b2 = (torch.randn(3, 16) / 3).requires_grad_(True), torch.randn(3).requires_grad_(True)
    x_pos = torch.randn(n, 3)
    dout = torch.randn(n, 3)
    z = f_ref(y.double(), a.double(), b.double()).requires_grad_(True)
    t = torch.nn.functional.leaky_relu(z @ W1.double().t() + b1.double(), 0.01)
    u = t @ W2.double().t() + b2.double()
    if kind == 0:
        out = x_pos.double() + u
    else:
        v = torch.tanh(u)
        out = v * torch.reciprocal(torch.norm(v, dim=1, keepdim=True) + 1e-12)
    gz, gW1, gb1, gW2, gb2 = torch.autograd.grad(out, [z, W1, b1, W2, b2], dout.double())

    bn4 = torch.stack([a, b, torch.zeros(32), torch.ones(32)]).to(dev)
    P = [p.detach().to(dev).contiguous() for p in (W1, b1, W2, b2)]
    o = torch.empty(n, 3, device=dev)
    ops.head_fwd(y.to(dev), bn4, *P, kind, x_pos.to(dev), o)
    assert relerr(o, out) < 2e-6
    dz = torch.empty(n, 32, device=dev)
    G = [torch.empty_like(p) for p in P]
    ops.head_bwd(y.to(dev), bn4, *P, kind, dout.to(dev), dz, *G)
    assert relerr(dz, gz) < 1e-5
    for got, ref in zip(G, (gW1, gb1, gW2, gb2)):
        assert relerr(got, ref) < 1e-5


def test_clip_and_adam_match_torch(dev):
    from dual_dmp_amd import ops
    torch.manual_seed(0)
    n = 100003
    p0 = torch.randn(n)
    pt = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([pt], lr=0.01)
    p = p0.to(dev)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for step in range(1, 6):
        g = torch.randn(n) * (3.0 if step % 2 else 1e-3)
        pt.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([pt], 0.8)
        opt.step()
        gd = g.to(dev)
        ss = ops.grad_sumsq(gd)
        ops.adam_step_(p, gd, m, v, 0.01, step, clip_sumsq=ss, max_norm=0.8)
        assert relerr(p, pt.detach()) < 1e-6
        g2 = gd.clone()
        ops.grad_clip_(g2, ss, 0.8)
        assert relerr(g2, pt.grad) < 1e-6
    # without clip
    ops.adam_step_(p, gd, m, v, 0.01, 6)
    pt.grad = g.clone()
    opt.step()
    assert relerr(p, pt.detach()) < 1e-6


def test_no_cpu_fallback():
    """The product path refuses CPU tensors instead of silently computing elsewhere."""
    from dual_dmp_amd import ops, _lib
    with pytest.raises(_lib.DdmpError):
        ops.gemm_nt(torch.randn(8, 8), torch.randn(8, 8))


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_spmm_forms_are_bit_stable_beside_panel_gemm(dtype):
    """Two-stream iterations run the gather of one net beside the other net's row-panel GEMMs on the same CUs.  A version of
    the lean SpMM kernel that kept (quad count, dinv[row]) in ONE 64-bit LDS word returned zeros in the low lane of the packed
    FMA that broadcast its high half -- only beside the f16x3 panel GEMM, 30-60 % of launches.  Every form
    of the gather has to reproduce its own result bit for bit while that GEMM runs on a second stream."""
    from dual_dmp_amd import ops, synth
    from dual_dmp_amd.mesh import Mesh
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    dt = torch.bfloat16 if dtype == "bf16" else torch.float32
    v, f = synth.open_grid(20, 15)
    m = Mesh(vs=v, faces=f)
    e = torch.tensor(m.edges.T, dtype=torch.long)
    graphs = ((torch.cat([e, e[[1, 0]]], 1).to(dev), len(v)), (torch.from_numpy(m.f_edges).to(dev), len(f)))
    side = torch.cuda.Stream()
    A = torch.randn(20000, 256, device=dev)
    W = torch.randn(256, 256, device=dev) / 16.0
    H = torch.empty(20000, 256, device=dev)
    C = 256
    for idx, n in graphs:
        g = ops.graph_for(idx, n)
        X = torch.randn(n, C, device=dev).to(dt)
        Yp = torch.randn(n, C, device=dev).to(dt)
        sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        bn4 = torch.rand(4, C, device=dev) + 0.5
        c10 = torch.rand(2, C, device=dev) * 0.1
        bias = torch.randn(C, device=dev)
        forms = {"plain": lambda Y, s: ops.spmm(g, X, out=Y, bias=bias), "prologue": lambda Y, s: ops.spmm(g, X, out=Y, pro=(sc, sh)),
                 "bnred": lambda Y, s: ops.spmm_bnred(g, X, Y, Yp, bn4, s), "bnbwd": lambda Y, s: ops.spmm_bnbwd(g, X, Yp, bn4, c10, Y)}
        for name, fn in forms.items():
            Y0 = torch.empty(n, C, device=dev, dtype=dt)
            s0 = torch.zeros(2 * C, dtype=torch.float64, device=dev)
            fn(Y0, s0)
            torch.cuda.synchronize()
            for it in range(120):
                with torch.cuda.stream(side):
                    ops.gemm_nt(A, W, out=H)
                Y1 = torch.empty(n, C, device=dev, dtype=dt)
                s1 = torch.zeros(2 * C, dtype=torch.float64, device=dev)
                fn(Y1, s1)
                torch.cuda.synchronize()
                assert torch.equal(Y0, Y1) and torch.equal(s0, s1), (name, n, it)
