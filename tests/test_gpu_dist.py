"""Partitioned HIP path on ONE GPU: P logical ranks (threads, ThreadComm) with the real kernels on
halo-extended local CSR graphs (n_rows < n_cols) must reproduce the unpartitioned FusedTrainer."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P,kind", [(2, "ico3"), (4, "ico3"), (3, "grid")])
def test_partitioned_step_matches_single_device(P, kind):
    from dual_dmp_amd import synth, dist as D
    from dual_dmp_amd.datamaker import dataset_from_meshes
    from dual_dmp_amd.networks import PosNet, NormalNet
    from dual_dmp_amd.trainer import FusedTrainer
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    v, f = synth.icosphere(3) if kind == "ico3" else synth.open_grid(20, 15)
    v, f = synth.permute_vertices(v, f, 4)
    gt, noisy, smooth = synth.make_triplet(v, f)
    data = dataset_from_meshes(noisy, smooth)
    steps = 2
    torch.manual_seed(0)
    posnet, normnet = PosNet(dev), NormalNet(dev)
    ref = FusedTrainer(posnet, normnet, data, noisy, bnfloop=5)
    ref.epoch = 100                                   # BNF term active
    ref_hist = []
    for _ in range(steps):
        ref_hist.append((ref.step().item(), ref.pos.clone(), ref.norm.clone()))
    ref_grads = (posnet._grad_arena.clone(), normnet._grad_arena.clone())

    nets = []
    for _ in range(P):
        torch.manual_seed(0)
        nets.append((PosNet(dev), NormalNet(dev)))
    comms = D.ThreadComm.make(P)
    results, errs = {}, []

    def work(r):
        try:
            torch.cuda.set_device(0)
            tr = D.make_distributed_trainer(noisy, smooth, data, dev, r, P, bnfloop=5, backend=comms[r], nets=nets[r])
            tr.epoch = 100
            hist = []
            for _ in range(steps):
                hist.append((tr.step().item(), tr.gather_pos().clone(), tr.gather_norm().clone()))
            results[r] = (hist, tr)
        except BaseException as e:       # noqa: BLE001
            errs.append(e)
            comms[r].s.barrier.abort()

    ths = [threading.Thread(target=work, args=(r,)) for r in range(P)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errs, errs
    for r in range(P):
        hist, tr = results[r]
        assert tr.peng.n_rows < tr.peng.n_cols and tr.neng.n_rows < tr.neng.n_cols      # halos exist
        # iteration 1 from identical state: strict; iteration 2 carries the Adam sign-flip noise
        (l0, p0, n0), (l1, p1, n1) = ref_hist[0], hist[0]
        assert abs(l0 - l1) <= 1e-6 * abs(l0), (r, l0, l1)
        assert float((p0 - p1).abs().max()) < 2e-5 and float((n0 - n1).abs().max()) < 2e-5
        assert abs(ref_hist[1][0] - hist[1][0]) <= 1e-2 * abs(ref_hist[1][0])
        assert torch.equal(tr.posnet.arena.data, results[0][1].posnet.arena.data)          # replicas in sync
        assert torch.equal(tr.normnet.arena.data, results[0][1].normnet.arena.data)
    assert sum(results[r][1].peng.n_rows for r in range(P)) == len(noisy.vs)
    assert sum(results[r][1].neng.n_rows for r in range(P)) == len(noisy.faces)


@pytest.mark.parametrize("P,kind,loop,partition", [(2, "ico3", 1, "morton"), (4, "ico3", 3, "morton"), (3, "grid", 2, "morton"),
                                                   (4, "grid", 1, "random"), (3, "ico3", 2, "random")])
def test_sharded_losses_match_whole_mesh_losses(P, kind, loop, partition):
    """Every rank evaluates the loss terms of its own rows on its ghost closure (dist.LossShard): the twelve loss scalars
    and the gradient rows each rank keeps must be those of the whole-mesh LossEngine on the same pos / norm -- closed
    mesh and open mesh (f2f == -1 -> the mesh's LAST face), Morton chunks and a scattered (worst-case) ownership."""
    from dual_dmp_amd import synth, dist as D
    from dual_dmp_amd.datamaker import dataset_from_meshes
    from dual_dmp_amd.networks import PosNet, NormalNet
    from dual_dmp_amd.loss import LossEngine
    dev = torch.device("cuda:0")
    v, f = synth.icosphere(3) if kind == "ico3" else synth.open_grid(20, 15)
    v, f = synth.permute_vertices(v, f, 4)
    gt, noisy, smooth = synth.make_triplet(v, f)
    data = dataset_from_meshes(noisy, smooth)
    V, F = len(noisy.vs), len(noisy.faces)
    g = torch.Generator().manual_seed(7)
    pos = (torch.from_numpy(np.asarray(smooth.vs, dtype=np.float32)) + 0.02 * torch.randn(V, 3, generator=g)).to(dev)
    norm = torch.nn.functional.normalize(torch.from_numpy(np.asarray(noisy.fn, dtype=np.float32)) + 0.3 * torch.randn(F, 3, generator=g), dim=1).to(dev)
    whole = LossEngine(noisy, dev, bnfloop=loop)
    lb, dp, dn = whole.forward_backward(pos, norm, 1.0)
    lb, dp, dn = lb.clone(), dp.clone(), dn.clone()
    owner = None
    if partition == "random":
        owner = np.random.RandomState(3).randint(0, P, size=F).astype(np.int32)
    nets = []
    for _ in range(P):
        torch.manual_seed(0)
        nets.append((PosNet(dev), NormalNet(dev)))
    comms = D.ThreadComm.make(P)
    res, errs = {}, []

    def work(r):
        try:
            torch.cuda.set_device(0)
            sd = D.ShardedData(data, noisy, r, P, face_owner=owner)
            tr = D.DistributedTrainer(nets[r][0], nets[r][1], sd, noisy, comms[r], dev, bnfloop=loop, losses="sharded")
            ov, of = torch.from_numpy(sd.vplan.owned).to(dev), torch.from_numpy(sd.fplan.owned).to(dev)
            with D.ops.on_device(dev):
                l, a, b = tr._sharded_losses(pos[ov].contiguous(), norm[of].contiguous(), 1.0)
            res[r] = (l.clone(), a.clone(), b.clone(), ov, of, tr.lshard)
        except BaseException as e:       # noqa: BLE001
            errs.append(e)
            comms[r].s.barrier.abort()

    ths = [threading.Thread(target=work, args=(r,)) for r in range(P)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errs, errs
    seen_v, seen_f = torch.zeros(V, dtype=torch.bool, device=dev), torch.zeros(F, dtype=torch.bool, device=dev)
    for r in range(P):
        l, a, b, ov, of, ls = res[r]
        # the sums are float64 over a different association of the same float32 terms
        assert torch.allclose(l, lb, rtol=1e-9, atol=1e-12), (r, l, lb)
        sa, sb = float(dp.abs().max()), float(dn.abs().max())
        assert float((a - dp[ov]).abs().max()) <= 2e-6 * sa, (r, float((a - dp[ov]).abs().max()), sa)
        assert float((b - dn[of]).abs().max()) <= 2e-6 * sb, (r, float((b - dn[of]).abs().max()), sb)
        seen_v[ov] = True
        seen_f[of] = True
        if partition == "morton" and loop == 1:
            assert ls.vplan.n_cols < V and ls.fplan.n_cols < F                     # a closure, not the whole mesh
    assert bool(seen_v.all()) and bool(seen_f.all())


@pytest.mark.parametrize("P", [2, 4, 8])
def test_interior_boundary_split_is_bit_identical_with_and_without_overlap(P, monkeypatch):
    """Round 6: with more than one rank every aggregation runs as two launches -- the rows that
    reference no halo row (the leading chunks of the interior-first local order) while the layer's halo exchange travels, the
    boundary rows behind it (GcnEngine ``split``; row-slice graphs of ddmp_graph_create_csr_rows_host).  48,400 faces over P
    threaded ranks, gate open, bnfloop 2, two iterations:
      * split + overlap (default) vs the same two launches with every exchange waited for first (DDMP_DIST_SPLIT=noverlap): bit-identical
        losses, outputs and parameters;
      * vs the unsplit partitioned path (DDMP_DIST_SPLIT=0: one launch per aggregation): equal to float32 rounding -- the fused
        column sums of the two halves are added in float64, another association of the same terms;
      * the split really happened on every rank and both graphs (asserted)."""
    from dual_dmp_amd import synth, dist as D
    from dual_dmp_amd.datamaker import dataset_from_meshes
    from dual_dmp_amd.networks import PosNet, NormalNet
    dev = torch.device("cuda:0")
    v, f = synth.torus(220, 110)
    v, f = synth.permute_vertices(v, f, 4)
    gt, noisy, smooth = synth.make_triplet(v, f)
    data = dataset_from_meshes(noisy, smooth)

    def run(env):
        monkeypatch.delenv("DDMP_DIST_SPLIT", raising=False)
        for k, val in env.items():
            monkeypatch.setenv(k, val)
        nets = []
        for _ in range(P):
            torch.manual_seed(0)
            nets.append((PosNet(dev), NormalNet(dev)))
        comms = D.ThreadComm.make(P)
        results, errs = {}, []

        def work(r):
            try:
                torch.cuda.set_device(0)
                tr = D.make_distributed_trainer(noisy, smooth, data, dev, r, P, bnfloop=2, backend=comms[r], nets=nets[r])
                tr.epoch = 100
                hist = [(tr.step().item(), tr.gather_pos().clone(), tr.gather_norm().clone()) for _ in range(2)]
                results[r] = (hist, tr.posnet.arena.data.clone(), tr.normnet.arena.data.clone(),
                              (tr.peng.split is not None, tr.neng.split is not None), (tr.peng.n_int if tr.peng.split else 0, tr.peng.n_rows))
            except BaseException as e:       # noqa: BLE001
                errs.append(e)
                comms[r].s.barrier.abort()
        ths = [threading.Thread(target=work, args=(r,)) for r in range(P)]
        [t.start() for t in ths]
        [t.join() for t in ths]
        assert not errs, errs
        return results

    a, b, c = run({}), run({"DDMP_DIST_SPLIT": "noverlap"}), run({"DDMP_DIST_SPLIT": "0"})
    for r in range(P):
        assert a[r][3] == (True, True) and b[r][3] == (True, True) and c[r][3] == (False, False), (r, a[r][3], c[r][3])
        assert 0 < a[r][4][0] < a[r][4][1]
        for (la, pa, na), (lb, pb, nb), (lc, pc, nc) in zip(a[r][0], b[r][0], c[r][0]):
            assert la == lb and torch.equal(pa, pb) and torch.equal(na, nb)
            assert abs(la - lc) <= 2e-6 * abs(lc), (la, lc)
        assert torch.equal(a[r][1], b[r][1]) and torch.equal(a[r][2], b[r][2])
    (l1, p1, n1), (lc1, pc1, nc1) = a[0][0][0], c[0][0][0]
    assert float((p1 - pc1).abs().max()) < 2e-5 and float((n1 - nc1).abs().max()) < 2e-5
