"""PosNet / NormalNet on the HIP path (host-side mirror of ``util/networks.py`` of the reference).

Two interchangeable forms, both taking the reference's ``Dataset`` object (fields ``z1``, ``z2``,
``x_pos``, ``edge_index``, ``face_index``) and returning ``pos`` [V,3] / ``norm`` [F,3]:

``PosNet(device)`` / ``NormalNet(device)``                              (``fused=True``, default)
    the whole trunk + head runs through :class:`engine.GcnEngine` (prologue-fused kernels, only conv
    outputs stored).  All parameters live in one flat float32 ``arena`` parameter, so
    ``torch.optim.Adam(net.parameters())`` and ``clip_grad_norm_(net.parameters(), ..)`` of the
    reference's loop (main.py:54-55,108-110) act on a single tensor; ``state_dict()`` /
    ``load_state_dict()`` speak the reference's names (``conv1.lin.weight`` ... ``linear2.bias``,
    ``bn1.running_mean`` ...).

``PosNet(device, fused=False)`` / ``NormalNet(device, fused=False)``
    the reference's module structure verbatim -- 12 x (``GCNConv`` -> ``nn.BatchNorm1d`` ->
    ``nn.LeakyReLU``) + two ``nn.Linear`` -- with our drop-in :class:`nn_ops.GCNConv` standing where
    ``torch_geometric.nn.GCNConv`` stands in ``util/networks.py:4``.

The reference's unused ``torch.randn(V,3)*1e-5`` draw (``util/networks.py:50``) is dropped: it only
advances the RNG.  ``z1``/``z2`` carry ``requires_grad=True`` in the reference but are never
optimised; their gradient is not computed here.
"""
from __future__ import annotations

import weakref
from collections import OrderedDict

import torch
import torch.nn as nn

from . import ops
from .engine import ArenaLayout, GcnEngine, NORM_WIDTHS, POS_WIDTHS
from .nn_ops import GCNConv


class _EngineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, arena, net):
        eng = net._engine
        with ops.on_device(eng.device):
            # nn.BatchNorm1d semantics: batch statistics (+ running-stat update) in train mode, running statistics in eval mode
            out = eng.forward(arena.detach(), update_running=net.training, use_running=not net.training)
            ctx.net = net
            ctx.trained = net.training
            return out.clone()

    @staticmethod
    def backward(ctx, dout):
        net = ctx.net
        eng = net._engine
        if not ctx.trained:
            raise RuntimeError("backward through a fused net in eval() mode is not implemented (the reference never leaves "
                               "train mode, main.py:88-89): call net.train() before the forward pass")
        with ops.on_device(eng.device):
            eng.backward(net.arena.detach(), net._grad_arena, dout)
            if eng.comm.world_size > 1:
                net._reduce_grads()
            return net._grad_arena.clone(), None


class _FusedNet(nn.Module):
    _widths = None
    _kind = 0

    def __init__(self, device, comm=None, reorder="rcb", dtype=torch.float32):
        """``reorder``: node numbering used INSIDE the engine ("rcb" = recursive coordinate bisection of the node
        coordinates -- smoothed vertex positions / noisy face centroids -- into leaves of 64 nodes, the gather kernels'
        chunk: 1.6 distinct neighbour rows per output row on the vertex graph of a 1M-face mesh where the Morton order
        ("morton", the default until round 5) has 1.9; "bfs" = breadth-first order of the graph, None = keep the
        caller's).  Results are returned in the caller's numbering either way.

        ``dtype``: ``torch.float32`` (default) or ``torch.bfloat16`` = bf16-feature mode: node features and their
        gradients are bfloat16 in HBM, parameters / optimizer state / outputs stay float32 (engine.GcnEngine)."""
        super().__init__()
        self.device = torch.device(device)
        self.reorder = reorder
        self.feature_dtype = dtype
        self.layout = ArenaLayout(self._widths)
        self.arena = nn.Parameter(torch.zeros(self.layout.total, dtype=torch.float32, device=self.device))
        self.layout.init_(self.arena.data)
        self._grad_arena = torch.zeros_like(self.arena.data)
        self._engine = None
        self._engine_key = None
        self.comm = comm

    # ------------------------------------------------------------ reference-named access
    def named_views(self, grads=False):
        src = self._grad_arena if grads else self.arena.data
        return OrderedDict((name, self.layout.view(src, name)) for name, *_ in self.layout.entries)

    def num_parameters(self):
        return self.layout.n_true_params()

    def _buffers_now(self):
        """BatchNorm buffers under the reference's names: the live engine's, else what a load left pending, else the
        nn.BatchNorm1d defaults (mean 0, var 1, 0 batches)."""
        eng, pend = self._engine, getattr(self, "_pending_running", None) or {}
        out = OrderedDict()
        for l in range(12):
            c = self.layout.cout[l]
            for j, nm in enumerate(("running_mean", "running_var")):
                k = "bn%d.%s" % (l + 1, nm)
                if eng is not None:
                    out[k] = eng.running[l][j].clone()
                elif k in pend:
                    out[k] = pend[k].clone()
                else:
                    out[k] = torch.full((c,), float(j), dtype=torch.float32, device=self.device)
            k = "bn%d.num_batches_tracked" % (l + 1)
            out[k] = (eng.num_batches_tracked[l].clone() if eng is not None else
                      pend[k].clone() if k in pend else torch.zeros((), dtype=torch.int64, device=self.device))
        return out

    def state_dict(self, *args, **kwargs):
        sd = OrderedDict()
        for name, v in self.named_views().items():
            sd[name] = v.detach().clone()
        sd.update(self._buffers_now())
        return sd

    def load_state_dict(self, sd, strict=True):
        """Reference-named keys (``conv1.lin.weight`` ... ``linear2.bias``, ``bnN.running_mean/var/num_batches_tracked``);
        returns the (missing_keys, unexpected_keys) pair like ``nn.Module.load_state_dict``.  BatchNorm buffers go into the
        live engine at once (or wait for the engine to be built)."""
        from torch.nn.modules.module import _IncompatibleKeys
        views = self.named_views()
        buf_keys = set(self._buffers_now().keys())
        missing = [k for k in list(views) + sorted(buf_keys) if k not in sd]
        unexpected = [k for k in sd if k not in views and k not in buf_keys]
        if strict and (missing or unexpected):
            raise KeyError("load_state_dict: missing keys %s, unexpected keys %s" % (missing, unexpected))
        with torch.no_grad():
            for name, v in views.items():
                if name in sd:
                    v.copy_(sd[name].to(v.device, torch.float32))
            self._pending_running = {k: sd[k].detach().clone() for k in buf_keys if k in sd}
            self._apply_pending_running()
        return _IncompatibleKeys(missing, unexpected)

    def _apply_pending_running(self):
        eng, pend = self._engine, getattr(self, "_pending_running", None)
        if eng is None or not pend:
            return
        for l in range(12):
            for j, nm in enumerate(("running_mean", "running_var")):
                k = "bn%d.%s" % (l + 1, nm)
                if k in pend:
                    eng.running[l][j].copy_(pend[k].to(eng.device, torch.float32))
            k = "bn%d.num_batches_tracked" % (l + 1)
            if k in pend:
                eng.num_batches_tracked[l] = int(pend[k])
        self._pending_running = None

    # ------------------------------------------------------------ engine plumbing
    def _inputs(self, data):
        raise NotImplementedError

    def attach_engine(self, engine):
        """Install an engine built elsewhere (dist.DistributedTrainer: this rank's shard of the graph); ``net(data)`` then
        runs on it instead of building a single-device engine for ``data``."""
        self._engine = engine
        self._engine_key = "external"
        self._apply_pending_running()

    def _get_engine(self, data):
        if self._engine is not None and self._engine_key == "external":
            return self._engine
        x0, x_pos, edge_index = self._inputs(data)
        key = self._engine_key
        same = (self._engine is not None and key is not None and key[0]() is x0 and key[1]() is edge_index
                and key[2] == (x0._version, edge_index._version))
        if not same:
            dev = self.device
            n = x0.shape[0]
            perm = self._node_order(x0, x_pos, edge_index, n)
            if perm is None:
                graph = ops.graph_for(edge_index.to(dev), n)
                x0d, xpd = x0.detach().to(dev), None if x_pos is None else x_pos.to(dev)
            else:
                inv = torch.empty_like(perm)
                inv[perm] = torch.arange(n)
                ei = inv[edge_index.detach().cpu()]
                self._relabelled = ei.to(dev)                    # keep alive: the graph cache holds a weak ref
                graph = ops.graph_for(self._relabelled, n)
                x0d = x0.detach().cpu()[perm].to(dev)
                xpd = None if x_pos is None else x_pos.detach().cpu()[perm].to(dev)
            self._engine = GcnEngine(graph, self._widths, self._kind, x0d, xpd, comm=self.comm, perm=perm,
                                     dtype=self.feature_dtype)
            self._engine_key = (weakref.ref(x0), weakref.ref(edge_index), (x0._version, edge_index._version))
            self._apply_pending_running()
        return self._engine

    def _coords(self, x0, x_pos):
        raise NotImplementedError

    def _node_order(self, x0, x_pos, edge_index, n):
        """new -> old permutation (CPU int64) or None."""
        if self.reorder is None or self.comm is not None:
            return None
        import numpy as np
        if self.reorder == "rcb":
            from .dist import rcb_order
            return torch.from_numpy(rcb_order(self._coords(x0, x_pos).detach().cpu().double().numpy(), 64))
        if self.reorder == "morton":
            from .dist import morton_order
            return torch.from_numpy(morton_order(self._coords(x0, x_pos).detach().cpu().double().numpy()).astype(np.int64))
        if self.reorder == "bfs":
            rowptr, col, _ = ops.csr_build_host(edge_index.detach().cpu().numpy(), n)
            return torch.from_numpy(ops.bfs_order_host(rowptr, col).astype(np.int64))
        raise ValueError("reorder must be 'rcb', 'morton', 'bfs' or None")

    def forward(self, data):
        if self.device.type != "cuda":
            raise ops.DdmpError("PosNet/NormalNet run on the HIP path only (device %s): there is no CPU fallback" % self.device)
        with ops.on_device(self.device):
            self._get_engine(data)
        return _EngineFn.apply(self.arena, self)

    def _reduce_grads(self):
        """Multi-device: weight gradients are sums over the row shards; BatchNorm weight/bias gradients
        were computed from already all-reduced column sums, so they are identical on every rank."""
        comm = self._engine.comm
        g = self._grad_arena
        keep = [(n, self.layout.view(g, n).clone()) for n, *_ in self.layout.entries if n.startswith("bn")]
        comm.all_reduce_sum(g)
        for n, v in keep:
            self.layout.view(g, n).copy_(v)


class PosNetFused(_FusedNet):
    """util/networks.py:8-67."""
    _widths = POS_WIDTHS
    _kind = 0

    def _inputs(self, data):
        return data.z1, data.x_pos, data.edge_index

    def _coords(self, x0, x_pos):
        return x_pos                                  # smoothed vertex positions


class NormalNetFused(_FusedNet):
    """util/networks.py:69-130."""
    _widths = NORM_WIDTHS
    _kind = 1

    def _inputs(self, data):
        return data.z2, None, data.face_index

    def _coords(self, x0, x_pos):
        return x0[:, :3]                              # z2 = [fc, fn, fa]: face centroids of the noisy mesh


# -------------------------------------------------------------------------------- operator-level form
class _ModularNet(nn.Module):
    _widths = None

    def __init__(self, device):
        super().__init__()
        self.device = torch.device(device)
        h = self._widths
        for i in range(12):
            setattr(self, "conv%d" % (i + 1), GCNConv(h[i], h[i + 1]))
        self.linear1 = nn.Linear(h[12], h[13])
        self.linear2 = nn.Linear(h[13], h[14])
        for i in range(12):
            setattr(self, "bn%d" % (i + 1), nn.BatchNorm1d(h[i + 1]))
        self.l_relu = nn.LeakyReLU()
        self.to(self.device)

    def _dev(self, data, name):
        """Device copy of a dataset tensor, kept on the dataset (the reference re-uploads on every forward,
        util/networks.py:49,110; a stable tensor object also lets the graph cache hit)."""
        t = getattr(data, name)
        if t.device == self.device:
            return t
        cache = data.__dict__.setdefault("_ddmp_dev", {})
        hit = cache.get((name, str(self.device)))
        if hit is None or hit[0] is not t or hit[1] != t._version:
            hit = (t, t._version, t.detach().to(self.device))
            cache[(name, str(self.device))] = hit
        return hit[2]

    def _trunk(self, x, edge_index):
        for i in range(1, 13):
            x = self.l_relu(getattr(self, "bn%d" % i)(getattr(self, "conv%d" % i)(x, edge_index)))
        return x


class PosNetModular(_ModularNet):
    _widths = POS_WIDTHS

    def forward(self, data):
        z1, x_pos, edge_index = self._dev(data, "z1"), self._dev(data, "x_pos"), self._dev(data, "edge_index")
        dx = self._trunk(z1, edge_index)
        dx = self.linear2(self.l_relu(self.linear1(dx)))
        return x_pos + dx


class NormalNetModular(_ModularNet):
    _widths = NORM_WIDTHS

    def forward(self, data):
        z2, edge_index = self._dev(data, "z2"), self._dev(data, "face_index")
        dx = self._trunk(z2, edge_index)
        dx = torch.tanh(self.linear2(self.l_relu(self.linear1(dx))))
        dx_norm = torch.reciprocal(torch.norm(dx, dim=1, keepdim=True).expand(-1, 3) + 1.0e-12)
        return torch.mul(dx, dx_norm)


def PosNet(device, fused=True, **kw):
    return PosNetFused(device, **kw) if fused else PosNetModular(device)


def NormalNet(device, fused=True, **kw):
    return NormalNetFused(device, **kw) if fused else NormalNetModular(device)
