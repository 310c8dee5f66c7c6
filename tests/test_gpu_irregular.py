"""Irregular-valence graphs through EVERY form of the gather.

The other GPU tests run on regular synthetic meshes (at most 7 CSR entries per row).  Meshes the reference is run on
(README.md:57-67; util/mesh.py:189-197 accepts any valence) are not regular, and until round 5 the gather's fast paths were
selected by the graph's LONGEST row.  Here:

  * small graphs against the dense float64 A_hat: a flipped torus (valence 3 ... 12+), the same with a valence-24 hub (a row
    of 25 entries: beyond the 16 the per-row LDS slots used to hold), a latitude / longitude sphere (two poles of valence 40),
    and a hand-made CSR with EMPTY rows, an empty first chunk and one row of 1500 entries (its chunk overflows the kernel's
    LDS slots: the in-kernel global-memory path);
  * a 72,200-vertex flipped torus in RCB order with hubs of valence 24, 300 and 1100 (>= 64k rows: the LDS-patch kernel with
    register entries + LDS tails; the hubs' chunks are "heavy" and go to the lean gather's chunk list) against a float64
    sparse product, on the vertex AND the face graph;
  * every form: plain, prologue + bias, forward statistics, BatchNorm-backward reductions, BatchNorm backward on the gather;
    float32 and bfloat16 features;
  * the same file again in its own process under each kernel-selection switch (LDS-patch forced / off, entries from LDS,
    round-2 slab kernel).

Tolerances: float32 rel-L2 <= 1e-6 against float64 (f32 FMA chains of <= 1500 terms: 2e-6 for the star graph); bfloat16
features: the float64 reference is evaluated on the SAME bf16-rounded inputs, rel-L2 <= 4e-3 (one rounding of the output).
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


def _vertex_edges(v, f):
    from dual_dmp_amd.mesh import Mesh
    m = Mesh(vs=v, faces=f)
    e = torch.tensor(m.edges.T, dtype=torch.long)
    return torch.cat([e, e[[1, 0]]], 1), torch.from_numpy(m.f_edges)


def _csr_of(ei, n):
    """(rowptr, col, dinv) as the library builds them from an edge_index (host code: ddmp_csr_build_host)."""
    from dual_dmp_amd import ops
    return ops.csr_build_host(ei.numpy(), n)


class G:
    """A graph under test: the device handle + its float64 sparse operator."""

    def __init__(self, dev, rowptr, col, dinv, n_cols, handle=None):
        from dual_dmp_amd import ops
        self.n, self.n_cols = len(rowptr) - 1, n_cols
        self.max_nnz = int(np.diff(rowptr).max())
        self.g = handle if handle is not None else ops.Graph.from_csr_host(rowptr, col, dinv, n_cols)
        rows = np.repeat(np.arange(self.n), np.diff(rowptr))
        w = dinv.astype(np.float64)[rows] * dinv.astype(np.float64)[col]
        self.A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, col.astype(np.int64)])), torch.from_numpy(w),
                                         (self.n, n_cols)).coalesce()

    def mm(self, x):
        return torch.sparse.mm(self.A, x.double().cpu())


@pytest.fixture(scope="module")
def small(dev):
    from dual_dmp_amd import synth, ops
    out = {}
    v, f = synth.torus(24, 12)
    f = synth.flip_edges(v, f, rounds=12, seed=2)
    hist = synth.valence_histogram(f, len(v))
    assert hist[:3].sum() == 0 and len(hist) - 1 >= 10, hist      # valence 3 ... >= 10
    fh = synth.add_hub(v, f, 17, 24)
    assert len(synth.valence_histogram(fh, len(v))) - 1 == 24
    vp, fp = synth.uv_sphere(40, 8)
    for name, (vv, ff) in {"flip": (v, f), "hub24": (v, fh), "pole40": (vp, fp)}.items():
        vv, ff = synth.permute_vertices(vv, ff, 5)
        ei, _ = _vertex_edges(vv, ff)
        rowptr, col, dinv = _csr_of(ei, len(vv))
        out[name] = G(dev, rowptr, col, dinv, len(vv), handle=ops.graph_for(ei.to(dev), len(vv)))
        out[name]._keep = ei
    # hand-made CSR: 200 rows over 260 columns; rows 0..69 empty (an empty first chunk), row 100 has 1500 entries (repeats
    # allowed: multi-edges), rows 150..155 empty, the rest 1..9 entries; arbitrary positive "dinv"
    rng = np.random.default_rng(7)
    n, nc = 200, 260
    lens = rng.integers(1, 10, size=n)
    lens[:70] = 0
    lens[150:156] = 0
    lens[100] = 1500
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    col = rng.integers(0, nc, size=int(rowptr[-1])).astype(np.int32)
    dinv = (rng.random(nc) * 0.9 + 0.1).astype(np.float32)
    out["star"] = G(dev, rowptr, col, dinv, nc)
    return out


@pytest.fixture(scope="module")
def big(dev):
    """72,200 vertices / 144,400 faces, flipped, three hubs, RCB order: vertex graph (rows up to 1101 entries) and face graph."""
    from dual_dmp_amd import synth, ops, _lib
    v, f = synth.torus(380, 190)
    f = synth.flip_edges(v, f, rounds=10, seed=1)
    for vert, val in ((1000, 24), (30000, 300), (60000, 1100)):
        f = synth.add_hub(v, f, vert, val)
    hist = synth.valence_histogram(f, len(v))
    assert len(hist) - 1 == 1100 and hist[300] == 1 and hist[24] >= 1
    order = ops.rcb_order_host(v, 64).astype(np.int64)
    inv = np.empty_like(order)
    inv[order] = np.arange(len(order))
    v, f = v[order], inv[f]
    f = f[ops.rcb_order_host(v[f].mean(1), 64)]
    ei, fi = _vertex_edges(v, f)
    out = {}
    for name, (idx, n) in {"vert": (ei, len(v)), "face": (fi, len(f))}.items():
        rowptr, col, dinv = _csr_of(idx, n)
        out[name] = G(dev, rowptr, col, dinv, n, handle=ops.graph_for(idx.to(dev), n))
        out[name]._keep = idx
    L = _lib.lib()
    if os.environ.get("DDMP_SPMM_PATCH") in (None, "1"):
        # the LDS-patch kernel takes both graphs whatever their longest row is (round 5)
        assert out["vert"].max_nnz == 1101 and out["face"].max_nnz == 4
        for g in out.values():
            assert L.ddmp_spmm_patch_selected(g.g._h, 256, 0, 0, 0) == 1 and L.ddmp_spmm_patch_selected(g.g._h, 256, 0, 1, 0) == 1
    return out


def f_ref(x, a, b, slope=0.01):
    z = x * a + b
    return torch.where(z > 0, z, slope * z)


def _forms(dev, gr, C, dtype, tol):
    """Every form of the gather on graph `gr` at width C against float64."""
    from dual_dmp_amd import ops
    g, n, nc = gr.g, gr.n, gr.n_cols
    torch.manual_seed(1000 * C + n)
    x = torch.randn(nc, C).to(dtype)
    xd = x.double()
    bias, a, b = torch.randn(C), torch.rand(C) + 0.5, torch.randn(C)
    xg = x.to(dev)
    # plain / prologue + bias
    y = ops.spmm(g, xg)
    assert y.shape == (n, C) and relerr(y, gr.mm(xd)) < tol, ("plain", relerr(y, gr.mm(xd)))
    yp = ops.spmm(g, xg, bias=bias.to(dev), pro=(a.to(dev), b.to(dev)))
    ref_p = gr.mm(f_ref(xd, a.double(), b.double())) + bias.double()
    assert relerr(yp, ref_p) < tol, ("prologue", relerr(yp, ref_p))
    # forward statistics of the output (fused where C allows, else the composition: the same call)
    if ops.spmm_stats_supported(C, dtype) or C % 8 == 0:
        out = torch.empty_like(y)
        sums = torch.zeros(2 * C, dtype=torch.float64, device=dev)
        ref_mean = (ref_p.mean(0) * 1.01).float().to(dev).contiguous()
        ops.spmm_stats(g, xg, out, ref_mean, sums, bias=bias.to(dev), pro=(a.to(dev), b.to(dev)))
        assert torch.equal(out, yp), "statistics form: output differs from the plain kernel's"
        od = out.double().cpu()
        want = torch.cat([od.sum(0), (od * od).sum(0)])
        assert relerr(sums, want) < 2e-6, ("stats", relerr(sums, want))
    # BatchNorm-backward reductions of the output
    ypre = (torch.randn(n, C) * 2 + 0.3).to(dtype)
    bn4 = torch.stack([torch.rand(C) + 0.5, torch.randn(C), torch.randn(C), torch.rand(C) + 0.5])
    out = torch.empty_like(y)
    sums = torch.zeros(2 * C, dtype=torch.float64, device=dev)
    ops.spmm_bnred(g, xg, out, ypre.to(dev), bn4.to(dev), sums)
    assert torch.equal(out, y), "reduction form: output differs from the plain kernel's"
    od, yd = out.double().cpu(), ypre.double()
    sc, sh, mu, rs = (t.double() for t in bn4)
    gg = od * torch.where(yd * sc + sh > 0, 1.0, 0.01)
    want = torch.cat([gg.sum(0), (gg * (yd - mu) * rs).sum(0)])
    # float32 partial sums over 16 rows, float64 above; the knife edge a*y+b == 0 does not occur with random data
    assert relerr(sums, want) < 2e-5, ("bnred", relerr(sums, want))
    # BatchNorm backward rebuilt on the gather (square graphs: dZ and Yb have n_cols rows)
    if ops.spmm_bnbwd_supported(C) and n == nc:
        dz = torch.randn(nc, C).to(dtype)
        yb = (torch.randn(nc, C) * 2 + 0.5).to(dtype)
        c10 = torch.stack([torch.randn(C) * 0.1, torch.randn(C) * 0.1])
        out = torch.empty_like(y)
        ops.spmm_bnbwd(g, dz.to(dev), yb.to(dev), bn4.to(dev), c10.to(dev), out)
        z = yb.double() * sc + sh
        dy = sc * dz.double() * torch.where(z > 0, 1.0, 0.01) + c10[0].double() * yb.double() + c10[1].double()
        assert relerr(out, gr.mm(dy)) < tol, ("bnbwd", relerr(out, gr.mm(dy)))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("C", [32, 64, 256, 16])
@pytest.mark.parametrize("gname", ["flip", "hub24", "pole40", "star"])
def test_gather_forms_on_small_irregular_graphs(dev, small, gname, C, dtype):
    tol = 4e-3 if dtype == torch.bfloat16 else (2e-6 if gname == "star" else 1e-6)
    _forms(dev, small[gname], C, dtype, tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("C", [256, 512, 128, 64])
@pytest.mark.parametrize("gname", ["vert", "face"])
def test_gather_forms_on_a_72k_vertex_flipped_mesh_with_hubs(dev, big, gname, C, dtype):
    _forms(dev, big[gname], C, dtype, 4e-3 if dtype == torch.bfloat16 else 2e-6)


def test_csr_host_graph_with_empty_rows_at_patch_size(dev):
    """from 64k rows a csr_host graph gets the LDS-patch tables; empty rows, an empty FIRST chunk and an
    empty LAST row used to read pl_col[-1] / uninitialised LDS there.  A ring graph over 70,000 rows with rows 0..63, every
    97th row and the last row empty."""
    from dual_dmp_amd import ops, _lib
    n = 70000
    lens = np.full(n, 5, dtype=np.int64)
    lens[:64] = 0
    lens[::97] = 0
    lens[-1] = 0
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    rows = np.repeat(np.arange(n), lens)
    k = np.arange(int(rowptr[-1])) - rowptr[rows]
    col = ((rows + k - 2) % n).astype(np.int32)
    dinv = (np.random.default_rng(3).random(n) * 0.5 + 0.5).astype(np.float32)
    gr = G(dev, rowptr, col, dinv, n)
    if os.environ.get("DDMP_SPMM_PATCH") in (None, "1"):
        assert _lib.lib().ddmp_spmm_patch_selected(gr.g._h, 256, 0, 0, 0) == 1
    for dtype, tol in ((torch.float32, 1e-6), (torch.bfloat16, 4e-3)):
        _forms(dev, gr, 256, dtype, tol)
    x = torch.randn(n, 256, device=dev)
    bias = torch.randn(256, device=dev)
    y = ops.spmm(gr.g, x, bias=bias)
    empty = torch.from_numpy(np.flatnonzero(lens == 0)).to(dev)
    assert torch.equal(y[empty], bias.expand(len(empty), -1))     # an empty row aggregates nothing: bias exactly


def _random_csr(seed, n, nc, band):
    """Random CSR: heavy-tailed row lengths (many 0 ... 12, a few 30 ... 90, now and then several hundred), columns anywhere
    (band = 0) or within +-band of the row (a graph with locality: what the LDS-patch tables are built for); repeated columns
    allowed."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, 13, size=n)
    k = max(1, n // 50)
    lens[rng.integers(0, n, size=k)] = rng.integers(30, 90, size=k)
    lens[rng.integers(0, n, size=3)] = rng.integers(200, 700, size=3)
    if seed % 2:
        lens[: min(n, 70)] = 0                                    # an empty first chunk
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    rows = np.repeat(np.arange(n), lens)
    if band:
        col = (rows * (nc / n)).astype(np.int64) + rng.integers(-band, band + 1, size=len(rows))
        col = np.clip(col, 0, nc - 1).astype(np.int32)
    else:
        col = rng.integers(0, nc, size=len(rows)).astype(np.int32)
    dinv = (rng.random(max(n, nc)) * 0.9 + 0.1).astype(np.float32)
    return rowptr, col, dinv


@pytest.mark.parametrize("seed,n,nc,band,C", [
    (1, 63, 63, 0, 32), (2, 64, 90, 0, 64), (3, 65, 65, 0, 256), (4, 129, 129, 0, 128), (5, 4097, 4097, 0, 512),
    (6, 1000, 5000, 0, 64), (7, 70001, 70001, 40, 256), (8, 66000, 66000, 25, 512), (9, 70001, 70001, 0, 256),
    (10, 131072, 131072, 60, 256), (11, 5000, 5000, 30, 16), (12, 65536, 70000, 50, 128)])
def test_gather_forms_on_random_csr_graphs(dev, seed, n, nc, band, C):
    """Fuzz: every form of the gather on random CSR graphs -- sizes around the 64-row chunk and the 64k-row threshold of the
    LDS-patch tables, rows of 0 ... several hundred entries, with and without locality, square and rectangular -- against the
    float64 sparse product, float32 and bfloat16."""
    rowptr, col, dinv = _random_csr(seed, n, nc, band)
    gr = G(dev, rowptr, col, dinv[:nc], nc)
    for dtype, tol in ((torch.float32, 3e-6), (torch.bfloat16, 4e-3)):
        if dtype == torch.bfloat16 and C % 8:
            continue
        _forms(dev, gr, C, dtype, tol)


@pytest.mark.parametrize("C", [256, 128])
def test_oversized_chunks_are_split_into_halves_or_quarters(dev, C):
    """Round 6: a chunk whose patch (distinct referenced rows) exceeds the LDS buffers of the LDS-patch gather is walked as 2 halves or
    4 quarters inside the same launch when every part fits; only what does not fit even then goes to the lean gather's heavy list.
    A designed graph: 70,400 rows with 4 local entries each (patches of ~70 rows: patch_kd = 3, 96-row buffers), and three chunks
    whose rows reference far-apart columns -- 2 distinct per row (whole patch 128: halves of 64 fit), 5 per row (halves 160,
    quarters 80 fit), 8 per row (quarters 128: stays heavy).  Every form against float64, float32 and bfloat16."""
    import ctypes
    from dual_dmp_amd import _lib
    if os.environ.get("DDMP_SPMM_PATCH") == "0":
        pytest.skip("LDS-patch gather switched off")
    n = 70400
    rng = np.random.default_rng(5)
    rows_cols = []
    special = {100: 2, 500: 5, 900: 8}                            # chunk -> distinct far columns per row
    for c in range(n // 64):
        r = np.arange(64 * c, 64 * c + 64)
        k = special.get(c)
        if k is None:
            cols = np.stack([r, np.clip(r + 1, 0, n - 1), np.clip(r - 1, 0, n - 1), np.clip(r + 3, 0, n - 1)], 1)
        else:                                                     # row i of the chunk: k columns nobody else in the chunk references
            cols = (np.arange(64 * k).reshape(64, k) * 97 + 1000 * c) % n
        rows_cols.append(cols)
    lens = np.array([len(x) for cc in rows_cols for x in cc])
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    col = np.concatenate([np.asarray(x) for cc in rows_cols for x in cc]).astype(np.int32)
    dinv = (rng.random(n) * 0.9 + 0.1).astype(np.float32)
    gr = G(dev, rowptr, col, dinv, n)
    if os.environ.get("DDMP_SPMM_PATCH") is None:
        kd, nh, ns = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        assert _lib.lib().ddmp_graph_patch_info(gr.g._h, ctypes.byref(kd), ctypes.byref(nh), ctypes.byref(ns)) == 0
        assert (kd.value, nh.value, ns.value) == (3, 1, 1 + 3), (kd.value, nh.value, ns.value)
    for dtype, tol in ((torch.float32, 3e-6), (torch.bfloat16, 4e-3)):
        _forms(dev, gr, C, dtype, tol)


class GSlice(G):
    """Rows [r0, r1) of a graph as a graph of their own (ops.Graph.from_csr_host(rows=...), ddmp_graph_create_csr_rows_host): the
    interior / boundary halves of a partitioned graph (dist.py, round 6).  Output row i is node r0 + i; the columns keep the
    numbering of the whole graph."""

    def __init__(self, dev, rowptr, col, dinv, n_cols, r0, r1):
        from dual_dmp_amd import ops
        self.n, self.n_cols = r1 - r0, n_cols
        rp = (rowptr[r0:r1 + 1] - rowptr[r0]).astype(np.int64)
        cc = col[rowptr[r0]:rowptr[r1]]
        self.max_nnz = int(np.diff(rp).max()) if self.n else 0
        self.g = ops.Graph.from_csr_host(rowptr, col, dinv, n_cols, rows=(r0, r1))
        rows = np.repeat(np.arange(self.n), np.diff(rp))
        w = dinv.astype(np.float64)[rows + r0] * dinv.astype(np.float64)[cc]
        self.A = torch.sparse_coo_tensor(torch.from_numpy(np.stack([rows, cc.astype(np.int64)])), torch.from_numpy(w),
                                         (self.n, n_cols)).coalesce()


@pytest.fixture(scope="module")
def sliced():
    """CSR tables of the flipped 144,400-face mesh's two graphs in RCB order (host arrays, built once per module)."""
    from dual_dmp_amd import ops, synth
    v, f = synth.torus(380, 190)
    f = synth.flip_edges(v, f, rounds=10, seed=1)
    f = synth.add_hub(v, f, 1000, 24)
    order = ops.rcb_order_host(v, 64).astype(np.int64)
    inv = np.empty_like(order)
    inv[order] = np.arange(len(order))
    v, f = v[order], inv[f]
    f = f[ops.rcb_order_host(v[f].mean(1), 64)]
    ei, fi = _vertex_edges(v, f)
    return {"vert": _csr_of(ei, len(v)) + (len(v),), "face": _csr_of(fi, len(f)) + (len(f),)}


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("C", [512, 256, 64])
@pytest.mark.parametrize("which,cut", [("face", 131072), ("face", 70016), ("vert", 66048), ("vert", 640)])
def test_gather_forms_on_row_slices_of_a_graph(dev, sliced, which, cut, C, dtype):
    """Round 6: every form of the gather on the two row slices [0, cut) and [cut, n) of the flipped 144k-face mesh's graphs (cuts
    at chunk multiples, as the interior-first order makes them; slices above and below the 64k-row threshold of the LDS-patch
    tables, i.e. both kernels): against the float64 operator of the slice, and -- plain and prologue forms -- BIT for bit the
    rows of the whole graph's output (a row's entries are summed in CSR order whichever chunk the row sits in)."""
    from dual_dmp_amd import ops
    if C != 256 and any(k.startswith("DDMP_SPMM") for k in os.environ):
        pytest.skip("switched re-runs of this file take the C = 256 cases only (suite time)")
    rowptr, col, dinv, n = sliced[which]
    whole = ops.Graph.from_csr_host(rowptr, col, dinv, n)
    tol = 4e-3 if dtype == torch.bfloat16 else 1e-6
    torch.manual_seed(C)
    x = torch.randn(n, C).to(dtype).to(dev)
    a, b, bias = (torch.rand(C) + 0.5).to(dev), torch.randn(C).to(dev), torch.randn(C).to(dev)
    y_all = ops.spmm(whole, x)
    yp_all = ops.spmm(whole, x, bias=bias, pro=(a, b))
    for r0, r1 in ((0, cut), (cut, n)):
        gs = GSlice(dev, rowptr, col, dinv, n, r0, r1)
        _forms(dev, gs, C, dtype, tol)
        assert torch.equal(ops.spmm(gs.g, x), y_all[r0:r1])
        assert torch.equal(ops.spmm(gs.g, x, bias=bias, pro=(a, b)), yp_all[r0:r1])


SWITCHES = [{"DDMP_SPMM_PATCH": "1"}, {"DDMP_SPMM_PATCH": "0"}, {"DDMP_SPMM_PATCH_NE": "0"}, {"DDMP_SPMM_LEAN": "0"},
            {"DDMP_SPMM_PATCH": "1", "DDMP_SPMM_PATCH_NE": "0"}]


_SWITCH_RUNS = {}


def _switch_runs():
    """All five switched re-runs are started TOGETHER, by whichever of the parametrised tests below runs first (each is a pytest
    process of ~60 s, most of it start-up and host-side mesh building: one after the other they were 300 of the suite's 810 s)."""
    if not _SWITCH_RUNS:
        for env in SWITCHES:
            e = dict(os.environ)
            e.update(env)
            key = tuple(sorted(env.items()))
            _SWITCH_RUNS[key] = subprocess.Popen(
                [sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-k", "not under_switch",
                 "-p", "no:cacheprovider"], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=os.path.dirname(HERE))
    return _SWITCH_RUNS


@pytest.mark.parametrize("env", SWITCHES, ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_irregular_graphs_under_switch(env):
    """The tests above once more, in their own process, with the gather's kernel selection switched (read once per process)."""
    if any(k.startswith("DDMP_SPMM") for k in os.environ):
        pytest.skip("already inside a switched run")
    proc = _switch_runs()[tuple(sorted(env.items()))]
    try:
        out, err = proc.communicate(timeout=1500)
    except subprocess.TimeoutExpired:
        proc.kill()
        raise
    assert proc.returncode == 0, (env, out[-3000:], err[-2000:])
