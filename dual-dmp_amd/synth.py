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


def laplacian_smooth(vs, vv_ptr, vv_idx, steps=30):
    deg = np.diff(vv_ptr).astype(np.float64)[:, None]
    rows = np.repeat(np.arange(len(vs)), np.diff(vv_ptr))
    p = vs.copy()
    for _ in range(steps):
        s = np.zeros_like(p)
        for c in range(3):
            s[:, c] = np.bincount(rows, weights=p[vv_idx, c], minlength=len(p))
        p = (p + 2.0 * s) / (2.0 * deg + 1.0)
    return p


def make_triplet(vs, faces, level=0.2, steps=30):
    """-> (gt_mesh, noisy_mesh, smooth_mesh) as :class:`Mesh` objects."""
    gt = Mesh(vs=vs, faces=faces)
    scale = mean_edge_length(gt.vs, gt.edges)
    gt = Mesh(vs=gt.vs / scale, faces=faces)
    nvs = gaussian_noise(gt.vs, gt.vn, level=level)
    noisy = Mesh(vs=nvs, faces=faces)
    svs = laplacian_smooth(noisy.vs, noisy.vv_ptr, noisy.vv_idx, steps=steps)
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
