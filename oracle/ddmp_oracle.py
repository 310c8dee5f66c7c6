"""CPU ORACLE for the Dual-DMP training hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this file.  Nothing under ``dual-dmp_amd/`` imports it; the product path has no CPU
fallback and fails loudly when the HIP library is missing.

It restates, in plain PyTorch-CPU / numpy, the algorithm of the reference
(astaka-pe/Dual-DMP @ /root/reference) for the path SURVEY.md §8 scopes:

  section                     follows reference                     pinned by
  --------------------------  ------------------------------------  ---------------------------
  mesh tables (python loops)  util/mesh.py:45-85,152-197            tests/golden/mesh_*.npz
  five losses, mad, fn        util/loss.py:16,37,55,86,140,261      tests/golden/loss_*.npz
  bnf (filter + vertex move)  util/loss.py:195-259                  tests/golden/bnf_*.npz
                              util/mesh.py:87-92 util/models.py:5   (values AND gradients)
  GCNConv                     torch-geometric==2.2.0 (requirements  **PARITY UNPINNED**: PyG is an
                              .txt:19): gcn_norm + lin + propagate  un-vendored dependency that is
                              as published (Kipf & Welling); call   not installable here and the
                              sites util/networks.py:15-26,51-62    reference holds no tests or
  PosNet / NormalNet          util/networks.py:8-67, :69-130        golden vectors for it.  The
  training step               main.py:88-110, main4real.py:53-75    PyG-shaped form below is checked
                                                                    against an independent float64
                                                                    dense-A-hat form only.

PyG 2.2.0 ``GCNConv(in, out)`` defaults restated here: improved=False, cached=False,
add_self_loops=True, normalize=True, bias=True; ``lin`` = Linear(in, out, bias=False) with
Glorot-uniform weight (a = sqrt(6/(in+out))), ``bias`` zeros;
forward = gcn_norm -> lin -> propagate(aggr="add", flow source_to_target) -> + bias;
gcn_norm: add one self loop (weight 1) per node, deg_i = sum of weights of edges whose TARGET
(col) is i, w_e = deg[row]^-1/2 * deg[col]^-1/2 with inf -> 0.
"""
from __future__ import annotations

import math
from collections import Counter

import numpy as np
import torch
import torch.nn as nn

POS_WIDTHS = [16, 32, 64, 128, 256, 256, 512, 512, 256, 256, 128, 64, 32, 16, 3]   # networks.py:13
NORM_WIDTHS = [7, 32, 64, 128, 256, 256, 512, 512, 256, 256, 128, 64, 32, 16, 3]   # networks.py:74


# =====================================================================  mesh tables
def mesh_tables_loops(vs: np.ndarray, faces: np.ndarray):
    """Per-face Python loops, small meshes only.  Follows util/mesh.py:45-85 (edges),
    :152-158 (vf), :176-187 (f2f, f_edges), :189-197 (v2v / v_dims)."""
    nv = len(vs)
    seen = {}
    edges = []
    for f in faces:
        for k in range(3):
            a, b = int(f[k]), int(f[(k + 1) % 3])
            e = (min(a, b), max(a, b))
            if e not in seen:
                seen[e] = len(edges)
                edges.append(e)
    edges = np.array(edges, dtype=np.int32)
    vf = [set() for _ in range(nv)]
    for i, f in enumerate(faces):
        for k in range(3):
            vf[int(f[k])].add(i)
    f2f, fe0, fe1 = [], [], []
    for i, f in enumerate(faces):
        around = list(vf[int(f[0])]) + list(vf[int(f[1])]) + list(vf[int(f[2])])
        cnt = Counter(around)
        nb = [j for j, c in cnt.items() if c == 2]
        fe0 += [i] * len(nb)
        fe1 += nb
        f2f.append(nb + [-1] * (3 - len(nb)))
    deg = np.zeros(nv, dtype=np.float32)
    for a, b in edges:
        deg[a] += 1
        deg[b] += 1
    return dict(edges=edges, vf=vf, f2f=np.array(f2f, dtype=np.int64),
                f_edges=np.array([fe0, fe1], dtype=np.int64), v_dims=deg)


def face_normals_np(vs: np.ndarray, faces: np.ndarray):
    """util/mesh.py:87-92 (float64 numpy)."""
    cr = np.cross(vs[faces[:, 1]] - vs[faces[:, 0]], vs[faces[:, 2]] - vs[faces[:, 0]])
    fa = 0.5 * np.sqrt((cr ** 2).sum(axis=1))
    fn = cr / (np.linalg.norm(cr, axis=1, keepdims=True) + 1e-24)
    return fn, fa


def bnf_np(fn: np.ndarray, vs: np.ndarray, faces: np.ndarray, f2f: np.ndarray, fc: np.ndarray, fa: np.ndarray,
           sigma_s=0.7, sigma_c=0.2, iters=1):
    """Classical bilateral normal filter + area-weighted vertex update, ``util/loss.py:195-259`` restated with one
    scatter instead of the per-vertex Python loop (:238-251).  ``fc`` / ``fa`` are the mesh's at entry and are recomputed
    after a sweep only when ``iters > 1`` (:254-256); ``f2f`` = -1 gathers the last face (numpy indexing, :209-216).
    -> (new_fn, new_vs, fc, fa)."""
    new_fn = np.asarray(fn, dtype=np.float64)
    vs = np.array(vs, dtype=np.float64)
    fc, fa = np.asarray(fc, dtype=np.float64), np.asarray(fa, dtype=np.float64)
    V = len(vs)
    cv = faces.reshape(-1)
    cf = np.repeat(np.arange(len(faces)), 3)
    for _ in range(iters):
        fc_dist = np.linalg.norm(fc[f2f] - fc[:, None, :], axis=2)
        neig_fn = new_fn[f2f]
        fn_dist = np.linalg.norm(neig_fn - new_fn[:, None, :], axis=2)
        w = np.exp(-1.0 * fc_dist ** 2 / (2 * sigma_c ** 2)) * np.exp(-1.0 * fn_dist ** 2 / (2 * sigma_s ** 2)) * fa[f2f]
        new_fn = np.sum(w[:, :, None] * neig_fn, 1)
        new_fn = new_fn / (np.linalg.norm(new_fn, axis=1, keepdims=True) + 1.0e-12)
        t = (fa[cf] * np.sum(new_fn[cf] * (fc[cf] - vs[cv]), 1))[:, None] * new_fn[cf]
        incr = np.zeros((V, 3))
        np.add.at(incr, cv, t)
        vs = vs + incr / np.bincount(cv, weights=fa[cf], minlength=V)[:, None]
        if iters > 1:
            fc = np.sum(vs[faces], 1) / 3.0
            _, fa = face_normals_np(vs, faces)
    return new_fn, vs, fc, fa


def mad_np(n1, n2) -> float:
    """util/loss.py:261-272 -- mean angular difference in degrees (float64 numpy)."""
    if isinstance(n1, torch.Tensor):
        n1 = n1.detach().cpu().numpy().copy()
    if isinstance(n2, torch.Tensor):
        n2 = n2.detach().cpu().numpy().copy()
    inner = np.sum(n1 * n2, 1)
    ang = np.rad2deg(np.arccos(np.clip(inner, -1.0, 1.0)))
    return float(np.sum(ang) / len(ang))


# =====================================================================  GCNConv (PyG 2.2.0)
def gcn_norm(edge_index: torch.Tensor, num_nodes: int, dtype=torch.float32):
    """PyG ``gcn_norm`` for an unweighted graph without pre-existing self loops."""
    row, col = edge_index[0], edge_index[1]
    loop = torch.arange(num_nodes, dtype=row.dtype, device=row.device)
    row = torch.cat([row, loop])
    col = torch.cat([col, loop])
    w = torch.ones(row.numel(), dtype=dtype, device=row.device)
    deg = torch.zeros(num_nodes, dtype=dtype, device=row.device).scatter_add_(0, col, w)
    dis = deg.pow(-0.5)
    dis[torch.isinf(dis)] = 0.0
    return row, col, dis[row] * w * dis[col]


class GCNConvRef(nn.Module):
    """PyG-shaped restatement: per call gcn_norm -> lin -> index_select * w -> scatter_add
    -> + bias.  This is also the "reference CPU path" that bench.py times."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin = nn.Linear(in_channels, out_channels, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_channels))
        a = math.sqrt(6.0 / (in_channels + out_channels))
        with torch.no_grad():
            self.lin.weight.uniform_(-a, a)

    def forward(self, x, edge_index):
        n = x.shape[0]
        row, col, w = gcn_norm(edge_index, n, x.dtype)      # cached=False: every call
        h = self.lin(x)
        msg = h.index_select(0, row) * w.unsqueeze(1)       # [M, C_out] message tensor
        out = torch.zeros(n, h.shape[1], dtype=h.dtype).index_add_(0, col, msg)
        return out + self.bias


def gcn_conv_dense(x, weight, bias, edge_index):
    """Independent float64 form: Y = D^-1/2 (A + I) D^-1/2 (X W^T) + b with a dense A
    (multi-edges counted with multiplicity, as scatter_add does)."""
    n = x.shape[0]
    A = torch.zeros(n, n, dtype=torch.float64)
    A.index_put_((edge_index[1], edge_index[0]), torch.ones(edge_index.shape[1], dtype=torch.float64),
                 accumulate=True)
    A += torch.eye(n, dtype=torch.float64)
    deg = A.sum(1)
    dis = deg.pow(-0.5)
    Ahat = dis[:, None] * A * dis[None, :]
    return Ahat @ (x.double() @ weight.double().t()) + bias.double()


# =====================================================================  networks
class _NetRef(nn.Module):
    def __init__(self, widths):
        super().__init__()
        h = widths
        for i in range(12):
            setattr(self, "conv%d" % (i + 1), GCNConvRef(h[i], h[i + 1]))
            setattr(self, "bn%d" % (i + 1), nn.BatchNorm1d(h[i + 1]))
        self.linear1 = nn.Linear(h[12], h[13])
        self.linear2 = nn.Linear(h[13], h[14])
        self.l_relu = nn.LeakyReLU()

    def trunk(self, x, edge_index):
        for i in range(12):
            x = getattr(self, "conv%d" % (i + 1))(x, edge_index)
            x = self.l_relu(getattr(self, "bn%d" % (i + 1))(x))
        return x


class PosNetRef(_NetRef):
    """util/networks.py:8-67.  The unused ``randn`` draw at :50 is omitted (it only
    advances the RNG; its value never reaches the output)."""

    def __init__(self):
        super().__init__(POS_WIDTHS)

    def forward(self, data):
        dx = self.trunk(data.z1, data.edge_index)
        dx = self.linear2(self.l_relu(self.linear1(dx)))
        return data.x_pos + dx


class NormalNetRef(_NetRef):
    """util/networks.py:69-130."""

    def __init__(self):
        super().__init__(NORM_WIDTHS)

    def forward(self, data):
        dx = self.trunk(data.z2, data.face_index)
        dx = torch.tanh(self.linear2(self.l_relu(self.linear1(dx))))
        inv = torch.reciprocal(torch.norm(dx, dim=1, keepdim=True).expand(-1, 3) + 1.0e-12)
        return dx * inv


# =====================================================================  losses
def _sq(x, dim):
    return torch.sum(x * x, dim=dim)


def pos_rec_loss(pred_pos, real_pos: np.ndarray):
    """util/loss.py:16-35 "rmse".  real_pos is float64 numpy -> the result is float64."""
    real = torch.from_numpy(np.asarray(real_pos))
    d = (real - pred_pos) ** 2
    return torch.sqrt(d.sum() / d.shape[0] + 1.0e-6)


def mesh_laplacian_loss(pred_pos, v2v_mat, v_dims):
    """util/loss.py:37-53 "rmse"."""
    lap = torch.sparse.mm(v2v_mat, pred_pos) / v_dims.reshape(-1, 1)
    r = _sq(pred_pos - lap, 1)
    return torch.sqrt(r.sum() / r.shape[0] + 1.0e-12)


def norm_rec_loss(pred_norm, real_norm: np.ndarray):
    """util/loss.py:55-84 "l1mae" (float64 by promotion)."""
    real = torch.from_numpy(np.asarray(real_norm))
    d = torch.abs(pred_norm - real).sum(1)
    return d.sum() / d.shape[0]


def fn_bnf_loss(pos, fn, faces: np.ndarray, f2f: np.ndarray, loop=5):
    """util/loss.py:86-138 "l1mae".  Quirks kept: pos detached; -1 entries of f2f gather
    the LAST face; sigma_c averages over all F*3 slots including padded ones; the padded
    slots only lose their weight through the area mask."""
    pos = pos.detach()
    faces_t = torch.from_numpy(np.asarray(faces)).long()
    p0, p1, p2 = pos[faces_t[:, 0]], pos[faces_t[:, 1]], pos[faces_t[:, 2]]
    fc = (p0 + p1 + p2) / 3.0
    cr = torch.linalg.cross(p1 - p0, p2 - p0, dim=1)
    fa = 0.5 * torch.sqrt(_sq(cr, 1) + 1.0e-12)
    nb = torch.from_numpy(np.asarray(f2f)).long()
    mask = (nb != -1).to(fn.dtype)
    nb_fc = fc[nb]                                   # negative index wraps to the last face
    nb_fa = fa[nb] * mask
    fc_dist = _sq(nb_fc - fc[:, None, :], 2)
    sigma_c = torch.sqrt(fc_dist + 1.0e-12).sum() / (fc_dist.shape[0] * fc_dist.shape[1])
    cur = fn
    for _ in range(loop):
        nb_fn = cur[nb]
        fn_dist = _sq(nb_fn - cur[:, None, :], 2)
        wc = torch.exp(-fc_dist / (2 * sigma_c ** 2))
        ws = torch.exp(-fn_dist / (2 * 0.3 ** 2))
        w = (wc * ws * nb_fa)[:, :, None]
        acc = (w * nb_fn).sum(1)
        cur = acc / (torch.sqrt(_sq(acc, 1)[:, None] + 1.0e-12) + 1.0e-12)
    d = torch.abs(cur - fn).sum(1)
    return d.sum() / d.shape[0], cur


def pos_norm_loss(pos, norm, faces: np.ndarray, num_verts: int):
    """util/loss.py:140-160 "mae": sum over the 3F (face, corner) pairs divided by V."""
    faces_t = torch.from_numpy(np.asarray(faces)).long()
    pf = pos[faces_t]                                # [F,3,3]
    fc = pf.sum(1) / 3.0
    pc = pf - fc[:, None, :]
    dots = torch.abs((pc * norm[:, None, :]).sum(2))
    return dots.sum() / num_verts


# =====================================================================  training step
class StepArgs:
    """Defaults of main.py:17-28."""

    def __init__(self, **kw):
        self.pos_lr = 0.01
        self.norm_lr = 0.01
        self.k1, self.k2, self.k3, self.k4, self.k5 = 3.0, 4.0, 4.0, 4.0, 1.0
        self.grad_crip = 0.8
        self.bnfloop = 1
        self.__dict__.update(kw)


def losses(pos, norm, mesh, args: StepArgs, epoch: int):
    l1 = pos_rec_loss(pos, mesh.vs)
    l2 = mesh_laplacian_loss(pos, mesh.v2v_mat, mesh.v_dims)
    l3 = norm_rec_loss(norm, mesh.fn)
    l4, _ = fn_bnf_loss(pos, norm, mesh.faces, mesh.f2f, loop=args.bnfloop)
    if epoch <= 100:                                  # main.py:101-102
        l4 = l4 * 0.0
    l5 = pos_norm_loss(pos, norm, mesh.faces, len(mesh.vs))
    total = args.k1 * l1 + args.k2 * l2 + args.k3 * l3 + args.k4 * l4 + args.k5 * l5
    return total, (l1, l2, l3, l4, l5)


def train_step(posnet, normnet, opt_pos, opt_norm, dataset, mesh, args: StepArgs, epoch: int):
    """main.py:88-110 (identical in main4real.py:53-75)."""
    posnet.train()
    normnet.train()
    opt_pos.zero_grad()
    opt_norm.zero_grad()
    pos = posnet(dataset)
    norm = normnet(dataset)
    total, parts = losses(pos, norm, mesh, args, epoch)
    total.backward()
    nn.utils.clip_grad_norm_(normnet.parameters(), args.grad_crip)
    opt_pos.step()
    opt_norm.step()
    return float(total.item()), pos.detach(), norm.detach(), [float(p) for p in parts]


class OracleDataset:
    """The fields the nets read from the reference ``Dataset`` (util/datamaker.py:8-21),
    built as util/datamaker.py:43-49,80-92 builds them."""

    def __init__(self, n_mesh, s_mesh):
        np.random.seed(314)
        z1 = np.random.normal(size=(n_mesh.vs.shape[0], 16))
        z2 = np.concatenate([n_mesh.fc, n_mesh.fn, n_mesh.fa.reshape(-1, 1)], axis=1)
        self.z1 = torch.tensor(z1, dtype=torch.float)
        self.z2 = torch.tensor(z2, dtype=torch.float)
        self.x_pos = torch.tensor(s_mesh.vs, dtype=torch.float)
        self.x_norm = torch.tensor(n_mesh.fn, dtype=torch.float)
        e = torch.tensor(np.asarray(n_mesh.edges).T, dtype=torch.long)
        self.edge_index = torch.cat([e, e[[1, 0], :]], dim=1)
        self.face_index = torch.from_numpy(np.asarray(n_mesh.f_edges)).long()
