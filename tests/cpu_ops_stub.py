"""TEST INFRASTRUCTURE: a PyTorch-CPU stand-in with the call signatures of ``dual_dmp_amd.ops``.

It exists so that the host logic of the multi-device path (partitioning, halo plans, collectives, the engine's
layer loop and gradient reduction) can be exercised on CPU with world_size > 1 over gloo, where no HIP kernel
can run.  Tests inject it with ``monkeypatch.setattr(engine, "ops", cpu_ops_stub)``; the product never imports
it and has no CPU fallback.  Arithmetic is float64 internally, float32 at the interfaces.
"""
import numpy as np
import torch

SLOPE = 0.01


def on_device(dev):
    import contextlib
    return contextlib.nullcontext()


def _f(x, pro, slope):
    if pro is None:
        return x
    z = x * pro[0].double() + pro[1].double()
    return torch.where(z > 0, z, slope * z)


class Graph:
    def __init__(self, rowptr, col, dinv, n_cols, row0=0):
        self.n_rows, self.n_cols, self.nnz = len(rowptr) - 1, int(n_cols), int(rowptr[-1])
        rows = np.repeat(np.arange(self.n_rows), np.diff(rowptr))
        d = torch.from_numpy(np.asarray(dinv, dtype=np.float64))
        vals = d[torch.from_numpy(rows + row0)] * d[torch.from_numpy(np.asarray(col, dtype=np.int64))]
        idx = torch.from_numpy(np.stack([rows, np.asarray(col, dtype=np.int64)]))
        self.A = torch.sparse_coo_tensor(idx, vals, size=(self.n_rows, self.n_cols)).coalesce()

    @classmethod
    def from_csr_host(cls, rowptr, col, dinv, n_cols, rows=None):
        rowptr, col = np.asarray(rowptr), np.asarray(col)
        if rows is not None:                                    # a row slice of the local graph (ops.Graph.from_csr_host)
            r0, r1 = int(rows[0]), int(rows[1])
            return cls(rowptr[r0:r1 + 1] - rowptr[r0], col[rowptr[r0]:rowptr[r1]], np.asarray(dinv), n_cols, row0=r0)
        return cls(rowptr, col, np.asarray(dinv), n_cols)

    @classmethod
    def from_edge_index(cls, edge_index, n):
        from dual_dmp_amd import ops
        rowptr, col, dinv = ops.csr_build_host(edge_index.cpu().numpy(), n)
        return cls(rowptr, col, dinv, n)


def graph_for(edge_index, n):
    return Graph.from_edge_index(edge_index, n)


def spmm(g, x, out=None, bias=None, pro=None, slope=SLOPE):
    y = torch.sparse.mm(g.A, _f(x[:g.n_cols].double(), pro, slope))
    if bias is not None:
        y = y + bias.double()
    if out is None:
        return y.float()
    out.copy_(y)
    return out


def gemm_nt(a, w, out=None, bias=None, pro=None, slope=SLOPE, n_rows=None):
    n = a.shape[0] if n_rows is None else n_rows
    y = _f(a[:n].double(), pro, slope) @ w.double().t()
    if bias is not None:
        y = y + bias.double()
    if out is None:
        return y.float()
    out[:n].copy_(y)
    return out


def gemm_nn(a, w, out=None, n_rows=None):
    n = a.shape[0] if n_rows is None else n_rows
    y = a[:n].double() @ w.double()
    if out is None:
        return y.float()
    out[:n].copy_(y)
    return out


def gemm_tn(g, z, out=None, pro=None, slope=SLOPE, n_rows=None):
    n = g.shape[0] if n_rows is None else n_rows
    y = g[:n].double().t() @ _f(z[:n].double(), pro, slope)
    if out is None:
        return y.float()
    out.copy_(y)
    return out


def bn_stats(y, sums=None, n_rows=None):
    n = y.shape[0] if n_rows is None else n_rows
    C = y.shape[1]
    yd = y[:n].double()
    r = torch.cat([yd.sum(0), (yd * yd).sum(0)])
    if sums is None:
        return r
    sums[:2 * C].copy_(r)
    return sums


def bn_prepare(sums, n_total, gamma, beta, out4, running=None, eps=1e-5, momentum=0.1):
    C = gamma.numel()
    mu = sums[:C] / n_total
    var = (sums[C:2 * C] / n_total - mu * mu).clamp_min(0)
    rs = 1.0 / torch.sqrt(var + eps)
    a = gamma.double() * rs
    out4[0].copy_(a)
    out4[1].copy_(beta.double() - mu * a)
    out4[2].copy_(mu)
    out4[3].copy_(rs)
    if running is not None:
        unb = var * n_total / (n_total - 1.0) if n_total > 1 else var
        running[0].mul_(1 - momentum).add_(momentum * mu.float())
        running[1].mul_(1 - momentum).add_(momentum * unb.float())
    return out4


def _g(dz, y, bn4, slope):
    z = y.double() * bn4[0].double() + bn4[1].double()
    return dz.double() * torch.where(z > 0, torch.ones_like(z), torch.full_like(z, slope))


def bn_bwd_reduce(dz, y, bn4, sums2=None, slope=SLOPE, n_rows=None):
    n = y.shape[0] if n_rows is None else n_rows
    C = y.shape[1]
    g = _g(dz[:n], y[:n], bn4, slope)
    yhat = (y[:n].double() - bn4[2].double()) * bn4[3].double()
    r = torch.cat([g.sum(0), (g * yhat).sum(0)])
    if sums2 is None:
        return r
    sums2[:2 * C].copy_(r)
    return sums2


def bn_bwd_prepare(sums2, n_total, bn4, dgamma, dbeta, c10):
    C = dgamma.numel()
    db, dg = sums2[:C], sums2[C:2 * C]
    a, mu, r = bn4[0].double(), bn4[2].double(), bn4[3].double()
    dbeta.copy_(db)
    dgamma.copy_(dg)
    k1 = -a * r * dg / n_total
    c10[0].copy_(k1)
    c10[1].copy_(-a * db / n_total - k1 * mu)


def bn_bwd_apply(dz, y, bn4, c10, dy, dbias_sums, slope=SLOPE, n_rows=None):
    n = y.shape[0] if n_rows is None else n_rows
    C = y.shape[1]
    g = _g(dz[:n], y[:n], bn4, slope)
    o = bn4[0].double() * g + c10[0].double() * y[:n].double() + c10[1].double()
    dy[:n].copy_(o)
    if dbias_sums is not None:
        dbias_sums[:C].copy_(o.sum(0))
    return dy


def f64_to_f32(src, dst):
    dst.copy_(src)
    return dst


def _head(y, bn4, W1, b1, W2, b2, kind, x_pos, slope):
    z = _f(y.double(), (bn4[0], bn4[1]), slope)
    t = torch.nn.functional.leaky_relu(z @ W1.double().t() + b1.double(), slope)
    u = t @ W2.double().t() + b2.double()
    if kind == 0:
        return z, x_pos.double() + u
    v = torch.tanh(u)
    return z, v * torch.reciprocal(torch.norm(v, dim=1, keepdim=True) + 1e-12)


def head_fwd(y, bn4, W1, b1, W2, b2, kind, x_pos, out, slope=SLOPE, n_rows=None):
    n = y.shape[0] if n_rows is None else n_rows
    _, o = _head(y[:n], bn4, W1, b1, W2, b2, kind, x_pos, slope)
    out.copy_(o)
    return out


def head_bwd(y, bn4, W1, b1, W2, b2, kind, dout, dz, dW1, db1, dW2, db2, slope=SLOPE, n_rows=None):
    n = y.shape[0] if n_rows is None else n_rows
    with torch.enable_grad():
        P = [p.detach().double().requires_grad_(True) for p in (W1, b1, W2, b2)]
        yy = y[:n].detach().double()
        z = _f(yy, (bn4[0], bn4[1]), slope).requires_grad_(True)
        t = torch.nn.functional.leaky_relu(z @ P[0].t() + P[1], slope)
        u = t @ P[2].t() + P[3]
        if kind == 0:
            o = u
        else:
            v = torch.tanh(u)
            o = v * torch.reciprocal(torch.norm(v, dim=1, keepdim=True) + 1e-12)
        gz, g1, g2, g3, g4 = torch.autograd.grad(o, [z] + P, dout[:n].double())
    dz[:n].copy_(gz)
    for dst, src in zip((dW1, db1, dW2, db2), (g1, g2, g3, g4)):
        dst.copy_(src)


def grad_sumsq(g, out=None):
    r = (g.double() ** 2).sum().reshape(1)
    if out is None:
        return r
    out.copy_(r)
    return out


def adam_step_(p, g, m, v, lr, step, betas=(0.9, 0.999), eps=1e-8, clip_sumsq=None, max_norm=0.0):
    c = 1.0
    if clip_sumsq is not None:
        c = min(1.0, max_norm / (float(clip_sumsq.sqrt()) + 1e-6))
    gi = g * c
    m.mul_(betas[0]).add_(gi, alpha=1 - betas[0])
    v.mul_(betas[1]).addcmul_(gi, gi, value=1 - betas[1])
    bc1, bc2 = 1 - betas[0] ** step, 1 - betas[1] ** step
    p.addcdiv_(m, v.sqrt() / bc2 ** 0.5 + eps, value=-lr / bc1)


class OracleLossEngine:
    """Replicated losses for the CPU tests: the oracle's loss functions + autograd."""

    def __init__(self, oracle, mesh, k, bnfloop):
        self.oracle, self.mesh = oracle, mesh
        self.args = oracle.StepArgs(k1=k[0], k2=k[1], k3=k[2], k4=k[3], k5=k[4], bnfloop=bnfloop)

    def forward_backward(self, pos, norm, gate4):
        with torch.enable_grad():
            p = pos.detach().clone().requires_grad_(True)
            n = norm.detach().clone().requires_grad_(True)
            total, parts = self.oracle.losses(p, n, self.mesh, self.args, 101 if gate4 else 1)
            gp, gn = torch.autograd.grad(total, [p, n])
        buf = torch.zeros(12, dtype=torch.float64)
        buf[5] = total.detach()
        return buf, gp, gn


class ShardedOracleLossEngine:
    """CPU stand-in of loss.LossEngine(shard=...) for the multi-rank tests: one rank's share of the five losses on its
    sub-mesh (dist.LossShard), written with torch ops after the oracle's formulas (oracle/ddmp_oracle.py: pos_rec_loss,
    mesh_laplacian_loss, norm_rec_loss, fn_bnf_loss, pos_norm_loss).  VALUES: partial sums over the rows the rank owns,
    all-reduced (sigma_c before the filter passes, S1..S5 before the scalars).  GRADIENTS: every local term with the global
    coefficients; exact on the owned rows because the closure holds everything those reach (the copy of the mesh's last
    face at the end of the local faces only serves the f2f == -1 gathers)."""

    def __init__(self, local_mesh, shard, k, bnfloop):
        self.m, self.sh, self.k, self.loop = local_mesh, shard, [float(x) for x in k], int(bnfloop)
        m = local_mesh
        self.real_v = torch.from_numpy(np.asarray(m.vs, dtype=np.float64))
        self.real_f = torch.from_numpy(np.asarray(m.fn, dtype=np.float64))
        self.faces = torch.from_numpy(np.asarray(m.faces, dtype=np.int64))
        self.f2f = torch.from_numpy(np.asarray(m.f2f, dtype=np.int64))
        e = torch.from_numpy(np.asarray(m.edges, dtype=np.int64))
        nv = len(m.vs)
        self.src = torch.cat([e[:, 0], e[:, 1]])
        self.dst = torch.cat([e[:, 1], e[:, 0]])
        self.deg = torch.bincount(self.src, minlength=nv).clamp(min=1).to(torch.float64)
        self.own_v = shard.own_v.to(torch.float64)
        self.own_f = shard.own_f.to(torch.float64)
        self.real_rows = torch.ones(len(m.faces), dtype=torch.float64)
        self.real_rows[-1] = 0.0                                   # the trailing copy of the mesh's last face

    def _sums(self, p, n, sigma_c):
        d1 = ((self.real_v - p) ** 2).sum(1)
        lap = torch.zeros_like(p).index_add_(0, self.src, p[self.dst]) / self.deg[:, None].to(p.dtype)
        d2 = ((p - lap) ** 2).sum(1)
        d3 = (n - self.real_f).abs().sum(1)
        pd = p.detach()
        p0, p1, p2 = pd[self.faces[:, 0]], pd[self.faces[:, 1]], pd[self.faces[:, 2]]
        fc = (p0 + p1 + p2) / 3.0
        cr = torch.linalg.cross(p1 - p0, p2 - p0, dim=1)
        fa = 0.5 * torch.sqrt((cr * cr).sum(1) + 1.0e-12)
        mask = (self.f2f != -1).to(n.dtype)
        fcd = ((fc[self.f2f] - fc[:, None, :]) ** 2).sum(2)
        dist_sum = (torch.sqrt(fcd + 1.0e-12).sum(1).double() * self.own_f).sum()
        d4 = None
        if sigma_c is not None:
            nb_fa = fa[self.f2f] * mask
            cur = n
            for _ in range(self.loop):
                nb = cur[self.f2f]
                fnd = ((nb - cur[:, None, :]) ** 2).sum(2)
                w = (torch.exp(-fcd / (2 * sigma_c ** 2)) * torch.exp(-fnd / (2 * 0.3 ** 2)) * nb_fa)[:, :, None]
                acc = (w * nb).sum(1)
                cur = acc / (torch.sqrt((acc * acc).sum(1)[:, None] + 1.0e-12) + 1.0e-12)
            d4 = (cur - n).abs().sum(1)
        pf = p[self.faces]
        pc = pf - (pf.sum(1) / 3.0)[:, None, :]
        d5 = (pc * n[:, None, :]).sum(2).abs().sum(1)
        return d1, d2, d3, d4, d5, dist_sum

    def forward_backward(self, pos, norm, gate4):
        sh, k = self.sh, self.k
        V, F = float(sh.V_glob), float(sh.F_glob)
        with torch.enable_grad():
            p = pos.detach().clone().requires_grad_(True)
            n = norm.detach().clone().requires_grad_(True)
            sig = self._sums(p, n, None)[5].reshape(1).clone()
            sh.all_reduce(sig)
            sigma_c = (sig[0] / (3.0 * F)).to(n.dtype)
            d1, d2, d3, d4, d5, _ = self._sums(p, n, sigma_c)
            S = torch.stack([(d1.double() * self.own_v).sum(), (d2.double() * self.own_v).sum(), (d3.double() * self.own_f).sum(),
                             (d4.double() * self.own_f).sum(), (d5.double() * self.own_f).sum()]).detach().clone()
            sh.all_reduce(S)
            l1, l2 = torch.sqrt(S[0] / V + 1.0e-6), torch.sqrt(S[1] / V + 1.0e-12)
            l3, l4, l5 = S[2] / F, S[3] / F * gate4, S[4] / V
            total = k[0] * l1 + k[1] * l2 + k[2] * l3 + k[3] * l4 + k[4] * l5
            # d total / d S_i, then every local term with that coefficient
            c = [k[0] / (2.0 * V * l1), k[1] / (2.0 * V * l2), k[2] / F, k[3] * gate4 / F, k[4] / V]
            G = (c[0] * d1.double().sum() + c[1] * d2.double().sum() + c[2] * (d3.double() * self.real_rows).sum()
                 + c[3] * (d4.double() * self.real_rows).sum() + c[4] * (d5.double() * self.real_rows).sum())
            gp, gn = torch.autograd.grad(G, [p, n])
        buf = torch.zeros(12, dtype=torch.float64)
        buf[5] = total.detach()
        return buf, gp, gn
