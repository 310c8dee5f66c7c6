"""Synthetic inputs for the training step (the reference's ``datasets.zip`` is absent).

Generators return ``(vs [V,3] f64, faces [F,3] i64)``:

  icosphere(k)        closed genus-0, 20*4^k faces, radially modulated ("non-CAD")
  torus(nu, nv)       closed genus-1, exactly 2*nu*nv faces / nu*nv vertices
                      (torus(1000, 500) is the 1,000,000-face / 500,000-vertex bench mesh)
  cube_cad(n)         subdivided cube, 12*n^2 faces, sharp edges (fandisk stand-in; n=33 ->
                      13,068 faces vs fandisk's 12,946)
  open_grid(nx, ny)   planar patch with a boundary (exercises the -1 padding of f2f)

``make_triplet`` produces the (gt, noise, smooth) triple the reference reads from
``*_gt.obj / *_noise.obj / *_smooth.obj`` (``util/datamaker.py:26-40``):
  * rescale to unit mean edge length (``preprocess/noisemaker.py:32-36``),
  * Gaussian noise along vertex normals, ``np.random.seed(314)``, level 0.2
    (``preprocess/noisemaker.py:38-42``),
  * 30 steps of uniform Laplacian smoothing of the noisy mesh.  MeshLab's
    ``laplacian_smooth`` (``preprocess/preprocess.py:22-24``) is not in the reference
    tree; the definition used here is p <- (p + 2*sum_nbr p_j) / (2*deg + 1) for interior
    vertices (every interior edge is met from both of its faces) -- own definition,
    parity with MeshLab is unpinned.
"""
from __future__ import annotations

import os

import numpy as np

from .mesh import Mesh


# ---------------------------------------------------------------- generators
def _icosahedron():
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0],
                  [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11],
                  [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
                  [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9],
                  [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    return v, f


def _subdivide(v, f):
    nv = len(v)
    a = f[:, [0, 1, 2]].reshape(-1)
    b = f[:, [1, 2, 0]].reshape(-1)
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    key = lo * np.int64(nv) + hi
    uk, inv = np.unique(key, return_inverse=True)
    mid = 0.5 * (v[uk // nv] + v[uk % nv])
    m = (inv + nv).reshape(-1, 3)              # midpoint ids of edges (01, 12, 20)
    v2 = np.concatenate([v, mid])
    f2 = np.concatenate([
        np.stack([f[:, 0], m[:, 0], m[:, 2]], 1),
        np.stack([f[:, 1], m[:, 1], m[:, 0]], 1),
        np.stack([f[:, 2], m[:, 2], m[:, 1]], 1),
        np.stack([m[:, 0], m[:, 1], m[:, 2]], 1)])
    return v2, f2


def icosphere(k: int, modulate: float = 0.15):
    v, f = _icosahedron()
    for _ in range(k):
        v, f = _subdivide(v, f)
        v /= np.linalg.norm(v, axis=1, keepdims=True)
    if modulate:
        r = 1.0 + modulate * (np.sin(3.0 * v[:, 0]) * np.cos(2.0 * v[:, 1]) + 0.5 * np.sin(5.0 * v[:, 2]))
        v = v * r[:, None]
    return v, f


def torus(nu: int, nv: int, R: float = 1.0, r: float = 0.4, wobble: float = 0.1):
    """nu x nv quad grid on a torus, each quad split in two: V = nu*nv, F = 2*nu*nv."""
    u = np.arange(nu)[:, None] * (2 * np.pi / nu)
    w = np.arange(nv)[None, :] * (2 * np.pi / nv)
    rr = r * (1.0 + wobble * np.sin(3 * u) * np.cos(2 * w))
    x = (R + rr * np.cos(w)) * np.cos(u)
    y = (R + rr * np.cos(w)) * np.sin(u)
    z = rr * np.sin(w) + 0.0 * u
    vs = np.stack([x, y, z], -1).reshape(-1, 3)
    i = np.arange(nu)[:, None]
    j = np.arange(nv)[None, :]
    i1, j1 = (i + 1) % nu, (j + 1) % nv
    a = (i * nv + j).reshape(-1)
    b = (i1 * nv + j).reshape(-1)
    c = (i1 * nv + j1).reshape(-1)
    d = (i * nv + j1).reshape(-1)
    faces = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)]).astype(np.int64)
    return vs, faces


def open_grid(nx: int, ny: int):
    """(nx x ny) vertices, 2*(nx-1)*(ny-1) faces, z = gentle bump."""
    x, y = np.meshgrid(np.arange(nx, dtype=np.float64), np.arange(ny, dtype=np.float64), indexing="ij")
    z = 0.3 * np.sin(0.9 * x) * np.cos(0.7 * y)
    vs = np.stack([x, y, z], -1).reshape(-1, 3)
    i = np.arange(nx - 1)[:, None]
    j = np.arange(ny - 1)[None, :]
    a = (i * ny + j).reshape(-1)
    b = ((i + 1) * ny + j).reshape(-1)
    c = ((i + 1) * ny + j + 1).reshape(-1)
    d = (i * ny + j + 1).reshape(-1)
    faces = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)]).astype(np.int64)
    return vs, faces


def cube_cad(n: int):
    """Surface of a cube, n x n quads per side, welded along the 12 sharp edges."""
    lin = np.linspace(-1.0, 1.0, n + 1)
    g0, g1 = np.meshgrid(lin, lin, indexing="ij")
    g0, g1 = g0.reshape(-1), g1.reshape(-1)
    one = np.ones_like(g0)
    sides = [np.stack([one, g0, g1], 1), np.stack([-one, g1, g0], 1),
             np.stack([g1, one, g0], 1), np.stack([g0, -one, g1], 1),
             np.stack([g0, g1, one], 1), np.stack([g1, g0, -one], 1)]
    vs_all = np.concatenate(sides)
    i = np.arange(n)[:, None]
    j = np.arange(n)[None, :]
    a = (i * (n + 1) + j).reshape(-1)
    b = ((i + 1) * (n + 1) + j).reshape(-1)
    c = ((i + 1) * (n + 1) + j + 1).reshape(-1)
    d = (i * (n + 1) + j + 1).reshape(-1)
    quad = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)])
    faces_all = np.concatenate([quad + s * (n + 1) ** 2 for s in range(6)])
    # weld duplicated border vertices
    q = np.round(vs_all * n).astype(np.int64)
    _, first, inv = np.unique(q, axis=0, return_index=True, return_inverse=True)
    order = np.argsort(first)
    remap = np.empty_like(order)
    remap[order] = np.arange(len(order))
    vs = vs_all[first[order]]
    faces = remap[inv.reshape(-1)][faces_all]
    return vs, faces.astype(np.int64)


# ----------------------------------------------------------- relabelling
def permute_vertices(vs, faces, seed=0):
    """Random vertex relabelling (worst-case gather locality)."""
    rng = np.random.default_rng(seed)
    perm = rng.permutation(len(vs))           # new id -> old id
    inv = np.empty_like(perm)
    inv[perm] = np.arange(len(perm))
    return vs[perm], inv[faces]


def morton_relabel(vs, faces):
    """Relabel vertices by the Morton order of their positions and faces by the Morton order of their
    centroids: consecutive ids form compact 2-D patches of the surface (gather locality)."""
    from .dist import morton_order
    vo = morton_order(vs)                       # new id -> old id
    inv = np.empty_like(vo)
    inv[vo] = np.arange(len(vo))
    vs2, f2 = vs[vo], inv[faces]
    fo = morton_order(vs2[f2].mean(1))
    return vs2, f2[fo]


def rcb_relabel(vs, faces, leaf=64):
    """Relabel vertices and faces by recursive coordinate bisection (``dist.rcb_order``): every 64 consecutive ids are a
    compact patch -- the numbering the engines apply internally since round 5."""
    from .dist import rcb_order
    vo = rcb_order(vs, leaf)
    inv = np.empty_like(vo)
    inv[vo] = np.arange(len(vo))
    vs2, f2 = vs[vo], inv[faces]
    return vs2, f2[rcb_order(vs2[f2].mean(1), leaf)]


def permute_faces(faces, seed=0):
    rng = np.random.default_rng(seed)
    return faces[rng.permutation(len(faces))]


# ------------------------------------------------- noise / smoothing / scaling
def mean_edge_length(vs, edges):
    d = vs[edges[:, 0]] - vs[edges[:, 1]]
    return float(np.sum(np.linalg.norm(d, axis=1)) / edges.shape[0])


def gaussian_noise(vs, vn, level=0.2, seed=314):
    """``preprocess/noisemaker.py:38-42``."""
    np.random.seed(seed)
    noise = np.random.normal(loc=0, scale=level, size=(len(vs), 1))
    return vs + vn * noise


def border_edges(faces):
    """Edges with ONE incident face, as an [nb, 2] array (empty for a closed mesh)."""
    f = np.asarray(faces, dtype=np.int64)
    he = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    key = np.sort(he, axis=1)
    uniq, cnt = np.unique(key, axis=0, return_counts=True)
    return uniq[cnt == 1]


def laplacian_smooth(vs, vv_ptr, vv_idx, steps=30, faces=None):
    """``ms.apply_filter("laplacian_smooth", stepsmoothnum=steps, cotangentweight=False)`` of the reference
    (preprocess/noisemaker.py:25-26, preprocess/preprocess.py:22-24; pymeshlab==2021.10, requirements.txt:6) without MeshLab.

    MeshLab's source is not part of the reference tree and pymeshlab cannot be installed here: this restates the published
    algorithm behind that filter -- MeshLab 2021.10 "Laplacian Smooth" (filter_unsharp, FP_LAPLACIAN; its other parameters at their
    defaults: Boundary = true, selected = false) = ``vcg::tri::Smooth<CMeshO>::VertexCoordLaplacian`` with
    ``AccumulateLaplacianInfo`` (vcglib, vcg/complex/algorithms/smooth.h), per step and simultaneously for all vertices:

    * every NON-border edge is visited from both of its faces and adds the opposite end point with weight 1 each time, so an
      interior vertex moves to  (p + 2 sum_j p_j) / (2 deg + 1);
    * a vertex on the border is averaged with its border neighbours only ("1D boundary smoothing"): its accumulator is reset to
      (sum = p, count = 1) and every border edge adds the other end point once --  (p + (p + sum_b p_b)) / ((1 + n_b) + 1)  =
      (2 p + p_a + p_b) / 4  on a manifold border.

    ``faces`` (optional): needed to find the border; without it every edge counts as interior (closed meshes).  MeshLab keeps
    coordinates in float32, this runs in float64: agreement to float32 rounding is the best a MeshLab run could show -- no such run
    exists here (parity of this row is unpinned; the algorithm is restated, not guessed)."""
    n = len(vs)
    deg = np.diff(vv_ptr).astype(np.float64)[:, None]
    rows = np.repeat(np.arange(n), np.diff(vv_ptr))
    be = border_edges(faces) if faces is not None else np.zeros((0, 2), dtype=np.int64)
    on_border = np.zeros(n, dtype=bool)
    on_border[be.ravel()] = True
    nb = np.bincount(be.ravel(), minlength=n).astype(np.float64)[:, None]
    p = np.asarray(vs, dtype=np.float64).copy()
    for _ in range(steps):
        s = np.zeros_like(p)
        for c in range(3):
            s[:, c] = np.bincount(rows, weights=p[vv_idx, c], minlength=n)
        new = (p + 2.0 * s) / (2.0 * deg + 1.0)
        if len(be):
            sb = np.zeros_like(p)
            np.add.at(sb, be[:, 0], p[be[:, 1]])
            np.add.at(sb, be[:, 1], p[be[:, 0]])
            new[on_border] = ((2.0 * p + sb) / (nb + 2.0))[on_border]
        p = new
    return p


def make_triplet(vs, faces, level=0.2, steps=30):
    """-> (gt_mesh, noisy_mesh, smooth_mesh) as :class:`Mesh` objects."""
    gt = Mesh(vs=vs, faces=faces)
    scale = mean_edge_length(gt.vs, gt.edges)
    gt = Mesh(vs=gt.vs / scale, faces=faces)
    nvs = gaussian_noise(gt.vs, gt.vn, level=level)
    noisy = Mesh(vs=nvs, faces=faces)
    svs = laplacian_smooth(noisy.vs, noisy.vv_ptr, noisy.vv_idx, steps=steps, faces=faces)
    smooth = Mesh(vs=svs, faces=faces)
    return gt, noisy, smooth


def write_dataset_dir(root, name, gt, noisy, smooth):
    """Lay out ``<root>/<name>/<name>_{gt,noise,smooth}.obj`` as ``create_dataset`` expects."""
    d = os.path.join(root, name)
    os.makedirs(d, exist_ok=True)
    if gt is not None:
        gt.save(os.path.join(d, name + "_gt.obj"))
    noisy.save(os.path.join(d, name + "_noise.obj"))
    smooth.save(os.path.join(d, name + "_smooth.obj"))
    return d


# --------------------------------------------------------- irregular valence
# The generators above are regular: every vertex of the torus / grid has valence 6 (the icosphere keeps twelve of valence
# 5).  Meshes the reference is run on (scans, non-CAD models: README.md:57-67; util/mesh.py:189-197 accepts any valence)
# are not, and the gather kernels' fast paths depend on the rows' lengths.  Two ways to get there from any closed mesh:
# random manifold-preserving edge flips (a valence spread like a decimated scan's, 3 ... 12+) and a "hub" vertex grown
# by targeted flips (a pole / cone apex).  Both keep V, F and the surface's topology; vertex positions are untouched.
def _flip_candidates(faces, nv):
    """Interior edges of an oriented manifold mesh as (f0, s0, f1, s1, A, B, C, D): half-edge A->B is slot s0 of
    face f0 = (A, B, C), its twin B->A slot s1 of face f1 = (B, A, D)."""
    a = faces[:, [0, 1, 2]].reshape(-1)
    b = faces[:, [1, 2, 0]].reshape(-1)
    c = faces[:, [2, 0, 1]].reshape(-1)
    key = np.minimum(a, b) * np.int64(nv) + np.maximum(a, b)
    order = np.argsort(key, kind="stable")
    ks = key[order]
    start = np.flatnonzero(np.r_[True, ks[1:] != ks[:-1]])
    sizes = np.diff(np.r_[start, len(ks)])
    two = start[sizes == 2]
    h0, h1 = order[two], order[two + 1]
    ok = (a[h1] == b[h0]) & (b[h1] == a[h0]) & (c[h0] != c[h1])          # consistently oriented, not a folded pair
    h0, h1 = h0[ok], h1[ok]
    return h0 // 3, h0 % 3, h1 // 3, h1 % 3, a[h0], b[h0], c[h0], c[h1], ks[start]


def _flip_ok(vs, A, B, C, D, min_quality=0.15):
    """Geometric sanity of replacing (A,B,C), (B,A,D) by (A,D,C), (D,B,C): the quad is convex in its own plane and
    neither new triangle is a sliver (area >= min_quality x the smaller old area)."""
    def nrm(p, q, r):
        return np.cross(vs[q] - vs[p], vs[r] - vs[p])
    n0, n1 = nrm(A, B, C), nrm(B, A, D)
    m0, m1 = nrm(A, D, C), nrm(D, B, C)
    ref = n0 + n1
    a_old = np.minimum(np.linalg.norm(n0, axis=1), np.linalg.norm(n1, axis=1))
    a_new = np.minimum(np.linalg.norm(m0, axis=1), np.linalg.norm(m1, axis=1))
    return ((m0 * ref).sum(1) > 0) & ((m1 * ref).sum(1) > 0) & (a_new >= min_quality * a_old)


def flip_edges(vs, faces, rounds: int = 10, seed: int = 0, min_valence: int = 3, max_valence: int = 0):
    """Random manifold-preserving edge flips.  Each round picks a vertex-disjoint random set of interior edges whose flip
    keeps every valence >= ``min_valence`` (and <= ``max_valence`` if given), does not duplicate an existing edge and
    keeps the quad geometrically sane; ~10 rounds turn a valence-6 torus into a mesh with valences 3 ... 12+.
    -> new faces [F,3] (vs unchanged)."""
    faces = np.array(faces, dtype=np.int64)
    nv = len(vs)
    rng = np.random.default_rng(seed)
    for _ in range(rounds):
        f0, s0, f1, s1, A, B, C, D, ekeys = _flip_candidates(faces, nv)
        val = np.bincount(np.concatenate([ekeys // nv, ekeys % nv]), minlength=nv)
        newkey = np.minimum(C, D) * np.int64(nv) + np.maximum(C, D)
        ok = (val[A] > min_valence) & (val[B] > min_valence) & ~np.isin(newkey, ekeys) & _flip_ok(vs, A, B, C, D)
        if max_valence:
            ok &= (val[C] < max_valence) & (val[D] < max_valence)
        cand = np.flatnonzero(ok)
        chosen = []
        for _pass in range(4):                                   # Luby passes: winners = local maxima of a random priority
            if len(cand) == 0:
                break
            pr = rng.random(len(cand))
            best = np.zeros(nv)
            quad = np.stack([A[cand], B[cand], C[cand], D[cand]])
            for q in quad:
                np.maximum.at(best, q, pr)
            win = (best[quad] == pr[None, :]).all(0)
            chosen.append(cand[win])
            taken = np.zeros(nv, dtype=bool)
            taken[quad[:, win].reshape(-1)] = True
            cand = cand[~win & ~taken[quad].any(0)]
        if not chosen:
            break
        w = np.concatenate(chosen)
        faces[f0[w]] = np.stack([A[w], D[w], C[w]], 1)
        faces[f1[w]] = np.stack([D[w], B[w], C[w]], 1)
    return faces


def add_hub(vs, faces, vertex: int = 0, valence: int = 24):
    """Grow ``vertex`` to the given valence by flipping, one at a time, an edge opposite to it in one of its incident
    faces (each such flip adds the far vertex to its 1-ring).  Connectivity only: the hub's faces become long and thin,
    which is what a pole or a cone apex looks like.  -> new faces [F,3]."""
    faces = np.array(faces, dtype=np.int64)
    nv = len(vs)
    h = int(vertex)
    order = np.argsort(faces.reshape(-1), kind="stable")
    ptr = np.concatenate([[0], np.cumsum(np.bincount(faces.reshape(-1), minlength=nv))])
    vf = {}                                                      # vertex -> set of incident faces, built on demand

    def inc(u):
        st = vf.get(u)
        if st is None:
            st = vf[u] = set((order[ptr[u]:ptr[u + 1]] // 3).tolist())
        return st

    grown = True
    while grown and len(inc(h)) < valence:
        grown = False
        ring = set(int(x) for f in inc(h) for x in faces[f])
        for f in sorted(inc(h)):
            tri = faces[f]
            k = int(np.flatnonzero(tri == h)[0])
            a, b = int(tri[(k + 1) % 3]), int(tri[(k + 2) % 3])          # face (h, a, b): flip its edge a-b
            g = (inc(a) & inc(b)) - {f}
            if len(g) != 1:
                continue
            g = g.pop()
            t2 = faces[g]
            d = int(t2[(t2 != a) & (t2 != b)][0])
            if d in ring or len(inc(a)) <= 3 or len(inc(b)) <= 3:
                continue
            faces[f] = (h, a, d)
            faces[g] = (h, d, b)
            inc(a).discard(g)                                    # f keeps a, g keeps b; h and d are in both now
            inc(b).discard(f)
            inc(h).add(g)
            inc(d).add(f)
            grown = True
            break
    return faces


def uv_sphere(n_seg: int, n_ring: int, modulate: float = 0.1):
    """Latitude / longitude sphere: two poles of valence ``n_seg`` (the everyday high-valence vertex), every other vertex of
    valence 6.  V = n_seg * (n_ring - 1) + 2, F = 2 * n_seg * (n_ring - 1)."""
    th = np.arange(1, n_ring)[:, None] * (np.pi / n_ring)
    ph = np.arange(n_seg)[None, :] * (2 * np.pi / n_seg)
    r = 1.0 + modulate * np.sin(3 * th) * np.cos(2 * ph)
    body = np.stack([r * np.sin(th) * np.cos(ph), r * np.sin(th) * np.sin(ph), r * np.cos(th) + 0 * ph], -1).reshape(-1, 3)
    vs = np.concatenate([[[0.0, 0.0, 1.0]], body, [[0.0, 0.0, -1.0]]])
    north, south = 0, len(vs) - 1
    idx = lambda i, j: 1 + i * n_seg + (j % n_seg)
    faces = []
    j = np.arange(n_seg)
    faces.append(np.stack([np.full(n_seg, north), idx(0, j), idx(0, j + 1)], 1))
    for i in range(n_ring - 2):
        a, b, c, d = idx(i, j), idx(i + 1, j), idx(i + 1, j + 1), idx(i, j + 1)
        faces.append(np.stack([a, b, c], 1))
        faces.append(np.stack([a, c, d], 1))
    faces.append(np.stack([np.full(n_seg, south), idx(n_ring - 2, j + 1), idx(n_ring - 2, j)], 1))
    return vs, np.concatenate(faces).astype(np.int64)


def valence_histogram(faces, nv):
    """-> counts[k] = vertices of valence k (closed manifold: valence = incident faces)."""
    return np.bincount(np.bincount(np.asarray(faces).reshape(-1), minlength=nv))
