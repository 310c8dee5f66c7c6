"""Drop-in ``GCNConv`` on the HIP kernels.

Same constructor / call signature, parameter names and initialisation as
``torch_geometric.nn.GCNConv`` 2.2.0 with the defaults the reference uses
(``util/networks.py:15-26``: ``GCNConv(in_channels, out_channels)``; ``:51-62``:
``conv(x, edge_index)``): ``lin.weight`` [out, in] Glorot-uniform, ``bias`` [out] zeros,
``Y = D^-1/2 (A + I) D^-1/2 (X W^T) + b``, differentiable w.r.t. x, weight and bias.

The normalised graph is built once per ``edge_index`` tensor (cached on its storage + version)
instead of on every call; aggregation runs on min(in, out) channels.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import ops


def _pad_cols(t: torch.Tensor, mult: int = 4) -> torch.Tensor:
    c = t.shape[1]
    if c % mult == 0 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0:
        return t
    cp = (c + mult - 1) // mult * mult
    out = torch.zeros((t.shape[0], cp), dtype=t.dtype, device=t.device)
    out[:, :c] = t
    return out


class _GCNConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, graph):
        cin, cout = weight.shape[1], weight.shape[0]
        xp = _pad_cols(x.detach().to(torch.float32))
        wp = _pad_cols(weight.detach())
        b = bias.detach().contiguous()
        agg_first = xp.shape[1] < cout
        if agg_first:
            p = ops.spmm(graph, xp)
            y = ops.gemm_nt(p, wp, bias=b)
            ctx.save_for_backward(p, wp)
        else:
            h = ops.gemm_nt(xp, wp)
            y = ops.spmm(graph, h, bias=b if cout % 4 == 0 else None)
            if cout % 4 != 0:
                y += b
            ctx.save_for_backward(xp, wp)
        ctx.graph, ctx.agg_first, ctx.cin = graph, agg_first, cin
        return y

    @staticmethod
    def backward(ctx, dy):
        with ops.on_device(dy):
            return _GCNConvFn._backward(ctx, dy)

    @staticmethod
    def _backward(ctx, dy):
        saved, wp = ctx.saved_tensors
        graph, cin = ctx.graph, ctx.cin
        dy = dy.contiguous()
        cout = dy.shape[1]
        dyp = _pad_cols(dy)
        if dyp.shape[1] != cout:           # ragged output width: pad the weight rows to match
            wrow = torch.zeros((dyp.shape[1], wp.shape[1]), dtype=wp.dtype, device=wp.device)
            wrow[:cout] = wp
        else:
            wrow = wp
        need_x = ctx.needs_input_grad[0]
        db = None
        if ctx.needs_input_grad[2]:
            pow2 = 8 <= cout <= 1024 and (cout & (cout - 1)) == 0
            db = ops.colsum(dy).to(torch.float32) if pow2 else dy.sum(0)
        if ctx.agg_first:
            dw = ops.gemm_tn(dyp, saved)
            dx = ops.spmm(graph, ops.gemm_nn(dyp, wrow)) if need_x else None
        else:
            dh = ops.spmm(graph, dyp)
            dw = ops.gemm_tn(dh, saved)
            dx = ops.gemm_nn(dh, wrow) if need_x else None
        dw = dw[:cout, :cin]
        if dx is not None:
            dx = dx[:, :cin]
        return dx, dw, db, None


class _Lin(nn.Module):
    """Holder so that the weight is addressed as ``conv.lin.weight`` like PyG's ``Linear``."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels))


class GCNConv(nn.Module):
    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin = _Lin(in_channels, out_channels)
        self.bias = nn.Parameter(torch.zeros(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        a = math.sqrt(6.0 / (self.in_channels + self.out_channels))     # PyG 'glorot'
        with torch.no_grad():
            self.lin.weight.uniform_(-a, a)
            self.bias.zero_()

    def forward(self, x: torch.Tensor, edge_index: torch.Tensor) -> torch.Tensor:
        if x.dim() != 2 or x.shape[1] != self.in_channels:
            raise ValueError("GCNConv: expected x of shape [N, %d]" % self.in_channels)
        if not x.is_cuda:
            raise ops.DdmpError("GCNConv runs on the HIP path only: x must be a CUDA (ROCm) tensor, there is no CPU fallback")
        with ops.on_device(x):
            graph = ops.graph_for(edge_index, x.shape[0])
            return _GCNConvFn.apply(x, self.lin.weight, self.bias, graph)

    def extra_repr(self):
        return "%d, %d" % (self.in_channels, self.out_channels)
