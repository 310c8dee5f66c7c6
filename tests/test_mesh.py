"""Vectorised Mesh builder == reference Mesh (golden) == loop oracle (random meshes)."""
import os

import numpy as np
import pytest

from dual_dmp_amd import synth
from dual_dmp_amd.mesh import Mesh

NAMES = ["ico2", "grid4", "cube3", "grid7x5"]


def _check(m, ref):
    assert m.vs.dtype == np.float64 and m.faces.dtype == np.int64
    assert m.edges.dtype == np.int32 and np.array_equal(m.edges, ref["edges"])
    assert m.f2f.dtype == np.int64 and m.f2f.shape == (len(m.faces), 3)
    assert np.array_equal(np.sort(m.f2f, 1), np.sort(ref["f2f"], 1))
    # padding sits at the end of the row, as in the reference
    pad = m.f2f < 0
    assert np.all(pad[:, :-1] <= pad[:, 1:])
    assert m.f_edges.dtype == np.int64 and m.f_edges.shape == ref["f_edges"].shape
    assert set(map(tuple, m.f_edges.T.tolist())) == set(map(tuple, ref["f_edges"].T.tolist()))
    assert np.array_equal(m.f_edges[0], np.sort(m.f_edges[0]))          # grouped by face
    assert np.array_equal(m.v_dims.numpy(), ref["v_dims"]) and m.v_dims.numpy().dtype == np.float32


@pytest.mark.parametrize("name", NAMES)
def test_mesh_matches_reference_golden(golden_dir, name, tmp_path):
    g = np.load(os.path.join(golden_dir, "mesh_%s.npz" % name))
    p = tmp_path / (name + ".obj")
    p.write_bytes(g["obj_text"].tobytes())
    m = Mesh(str(p))
    assert np.array_equal(m.vs, g["vs"]) and np.array_equal(m.faces, g["faces"])
    _check(m, g)
    for k in ("fn", "fa", "fc"):
        assert np.array_equal(getattr(m, k), g[k]), k
    np.testing.assert_allclose(m.vn, g["vn"], rtol=1e-13, atol=1e-15)
    assert np.array_equal(m.v2v_mat._indices().numpy(), g["v2v_indices"])
    assert np.array_equal(m.v2v_mat._values().numpy(), g["v2v_values"])
    assert np.array_equal(m.vf_ptr, g["vf_ptr"]) and np.array_equal(m.vf_idx, g["vf_idx"])
    assert [sorted(s) for s in m.vf] == [g["vf_idx"][g["vf_ptr"][i]:g["vf_ptr"][i + 1]].tolist()
                                         for i in range(len(m.vs))]
    # column order inside a vertex row follows CPython set iteration in the reference
    a, b = m.v2f_mat._indices().numpy(), g["v2f_indices"]
    assert np.array_equal(a[0], b[0])
    assert np.array_equal(a[:, np.lexsort(a[::-1])], b[:, np.lexsort(b[::-1])])
    # Mesh.save text format (util/mesh.py:267-285)
    out = tmp_path / "saved.obj"
    m.save(str(out))
    assert out.read_bytes() == g["save_text"].tobytes()


@pytest.mark.parametrize("seed", range(4))
def test_mesh_vs_loop_oracle_random_relabel(oracle, seed):
    vs, faces = synth.icosphere(1) if seed % 2 == 0 else synth.open_grid(5, 4)
    vs, faces = synth.permute_vertices(vs, faces, seed)
    faces = synth.permute_faces(faces, seed)
    rng = np.random.default_rng(seed)
    roll = rng.integers(0, 3, len(faces))
    faces = np.stack([np.roll(f, r) for f, r in zip(faces, roll)])
    m = Mesh(vs=vs, faces=faces)
    _check(m, oracle.mesh_tables_loops(vs, faces))


def test_obj_parser_variants(tmp_path):
    p = tmp_path / "t.obj"
    p.write_text("# c\nv 0 0 0\nv 1 0 0\n\nv 0 1 0\nv 0 0 1\nvn 0 0 1\nf 1/1/1 2/2/2 3/3/3\nf -4 -2 -1\n")
    m = Mesh(str(p))
    assert m.faces.tolist() == [[0, 1, 2], [0, 2, 3]]
    assert m.f2f.tolist() == [[1, -1, -1], [0, -1, -1]]
    q = tmp_path / "q.obj"
    q.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nf 1 2 3 4\n")
    with pytest.raises(AssertionError):
        Mesh(str(q))


def test_non_manifold_rejected():
    vs = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1.0]])
    faces = np.array([[0, 1, 2], [1, 0, 3], [0, 1, 4]])
    with pytest.raises(ValueError):
        Mesh(vs=vs, faces=faces)


def test_generators_and_triplet():
    v, f = synth.torus(20, 10)
    assert len(v) == 200 and len(f) == 400
    gt, noisy, smooth = synth.make_triplet(v, f)
    assert abs(synth.mean_edge_length(gt.vs, gt.edges) - 1.0) < 1e-12
    assert (gt.f2f >= 0).all()
    d_n = np.linalg.norm(noisy.vs - gt.vs, axis=1).mean()
    d_s = np.linalg.norm(smooth.vs - gt.vs, axis=1).mean()
    assert 0.05 < d_n < 0.4 and d_s > 0
    v, f = synth.cube_cad(4)
    m = Mesh(vs=v, faces=f)
    assert len(f) == 12 * 16 and len(v) - m.edges_count + len(f) == 2


def test_laplacian_smooth_follows_the_published_vcglib_rule():
    """synth.laplacian_smooth restates MeshLab 2021.10's "Laplacian Smooth" (preprocess/noisemaker.py:25-26: pymeshlab
    ``laplacian_smooth``, cotangentweight=False; vcglib Smooth::VertexCoordLaplacian): interior vertex <- (p + 2 sum_nbr p_j) /
    (2 deg + 1); border vertex <- (2 p + p_a + p_b) / 4 with its two BORDER neighbours only; all vertices at once.  Checked by hand
    on a 3 x 3 grid of vertices (8 triangles, one interior vertex) -- no MeshLab output exists to compare with."""
    xs, ys = np.meshgrid(np.arange(3.0), np.arange(3.0), indexing="ij")
    rng = np.random.default_rng(0)
    v = np.stack([xs.ravel(), ys.ravel(), rng.random(9)], 1)            # vertex i * 3 + j at (i, j, random height)
    idx = lambda i, j: i * 3 + j
    f = []
    for i in range(2):
        for j in range(2):
            f.append([idx(i, j), idx(i + 1, j), idx(i + 1, j + 1)])
            f.append([idx(i, j), idx(i + 1, j + 1), idx(i, j + 1)])
    f = np.array(f)
    m = Mesh(vs=v, faces=f)
    be = synth.border_edges(f)
    assert len(be) == 8 and 4 not in be                                  # the outer square; the centre vertex is interior
    out = synth.laplacian_smooth(m.vs, m.vv_ptr, m.vv_idx, steps=1, faces=f)
    nbr4 = sorted(m.vv_idx[m.vv_ptr[4]:m.vv_ptr[5]].tolist())
    assert nbr4 == [0, 1, 3, 5, 7, 8]                                    # the 1-ring of the centre in this triangulation
    assert np.allclose(out[4], (v[4] + 2 * v[nbr4].sum(0)) / (2 * 6 + 1))
    assert np.allclose(out[1], (2 * v[1] + v[0] + v[2]) / 4)             # an edge-midpoint border vertex: interior neighbour 4 ignored
    assert np.allclose(out[0], (2 * v[0] + v[1] + v[3]) / 4)             # a corner
    # simultaneous update: two steps == the one-step map applied twice
    two = synth.laplacian_smooth(m.vs, m.vv_ptr, m.vv_idx, steps=2, faces=f)
    assert np.allclose(two, synth.laplacian_smooth(out, m.vv_ptr, m.vv_idx, steps=1, faces=f))
    # a closed mesh has no border: with and without faces alike
    vi, fi = synth.icosphere(1)
    mi = Mesh(vs=vi, faces=fi)
    assert len(synth.border_edges(fi)) == 0
    assert np.array_equal(synth.laplacian_smooth(mi.vs, mi.vv_ptr, mi.vv_idx, steps=3, faces=fi),
                          synth.laplacian_smooth(mi.vs, mi.vv_ptr, mi.vv_idx, steps=3))


@pytest.mark.parametrize("name", ["ico2", "grid7x5", "cube3"])
def test_preprocess_numpy_halves_match_reference_golden(golden_dir, name, tmp_path):
    """SURVEY 8 f3, the half that needs no MeshLab, pinned to what the reference's own functions write
    (tests/golden/make_golden.py::noise_golden runs preprocess/noisemaker.py:32-42,60-73 and the body of
    preprocess/preprocess.py:56-78): the OBJ files come out byte for byte."""
    from dual_dmp_amd import preprocess
    from dual_dmp_amd.loss import mad
    g = np.load(os.path.join(golden_dir, "noise_%s.npz" % name))
    g_file, n_file, s_file = (str(tmp_path / (name + "_%s.obj" % k)) for k in ("gt", "noise", "smooth"))
    # noisemaker.py: pre-saved ground truth -> rescaled ground truth, noisy mesh
    open(g_file, "wb").write(g["pre_text"].tobytes())
    gt, noisy = preprocess.rescale_and_noise(g_file, n_file, float(g["level"]))
    assert open(g_file, "rb").read() == g["gt_text"].tobytes()
    assert open(n_file, "rb").read() == g["noise_text"].tobytes()
    np.testing.assert_allclose(noisy.vs, g["noise_vs"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(gt.vs, g["gt_vs"], rtol=0, atol=1e-15)
    assert abs(mad(noisy.fn, gt.fn) - float(g["mad"])) < 1e-9                 # the figure noisemaker.py:79-80 prints
    # preprocess.py: three normalised layers -> divided by the noisy mesh's mean edge length
    for f, k in ((n_file, "p_noise_in"), (s_file, "p_smooth_in"), (g_file, "p_gt_in")):
        open(f, "wb").write(g[k].tobytes())
    preprocess.rescale_saved(n_file, s_file, g_file)
    for f, k in ((n_file, "p_noise_out"), (s_file, "p_smooth_out"), (g_file, "p_gt_out")):
        assert open(f, "rb").read() == g[k].tobytes(), k
    # ... and without a ground truth (preprocess.py:61-65)
    for f, k in ((n_file, "p_noise_in"), (s_file, "p_smooth_in")):
        open(f, "wb").write(g[k].tobytes())
    os.remove(g_file)
    out = preprocess.rescale_saved(n_file, s_file, g_file)
    assert out[0] is None and open(n_file, "rb").read() == g["p_noise_out"].tobytes()
    assert open(s_file, "rb").read() == g["p_smooth_out"].tobytes()
