"""Tensor-level wrappers over the C ABI (``include/ddmp_hip.h``).

PyTorch is plumbing here: it owns device memory and the stream; every wrapper passes raw device
pointers + sizes to ``libddmp_hip.so`` and enqueues on ``torch.cuda.current_stream()``.
No wrapper has a CPU path: a non-CUDA tensor raises.
"""
from __future__ import annotations

import ctypes
import os
import threading
import weakref

import numpy as np
import torch

from . import _lib
from ._lib import DdmpError, check

SLOPE = 0.01        # nn.LeakyReLU() default, util/networks.py:44
BN_EPS = 1e-5       # nn.BatchNorm1d defaults
BN_MOMENTUM = 0.1


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    """hipStream_t of torch's current stream on the current device.  The raw getter (what torch's own extensions use) is
    ~20x cheaper than building a torch.cuda.Stream object per launch: 350 launches per iteration made that 2.7 ms of
    host time per step (scripts/host_profile.py)."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def on_device(dev):
    """Context manager: make ``dev`` (a tensor or a device) the current HIP device.  The C ABI launches on the current
    device's stream and ``ddmp_graph`` allocates there; every module-level entry point (nets, GCNConv, losses,
    trainers) enters this, so ``PosNet(torch.device("cuda:1"))`` works without a ``torch.cuda.set_device`` by the
    caller (as the reference's ``PosNet(device)`` does)."""
    if isinstance(dev, torch.Tensor):
        dev = dev.device
    dev = torch.device(dev)
    if dev.type != "cuda":
        raise DdmpError("the HIP path needs a CUDA (ROCm) device, got %s: there is no CPU fallback" % dev)
    return torch.cuda.device(dev)


class Profiler:
    """Per-call timing with HIP events recorded on the stream the kernels are launched on
    (torch's current stream), plus the ALGORITHMIC bytes / flops of each call.
    Off by default: ``ops.PROF = ops.Profiler()`` switches it on."""

    def __init__(self):
        self.records = []          # (name, key, alg_bytes, flops, survey_bytes, ev0, ev1)

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for name, key, b, f, b8, e0, e1 in self.records:
            a = agg.setdefault((name, key), dict(calls=0, ms=0.0, bytes=0.0, flops=0.0, bytes8d=0.0))
            a["calls"] += 1
            a["ms"] += e0.elapsed_time(e1)
            a["bytes"] += b
            a["flops"] += f
            a["bytes8d"] += b8
        return agg


PROF = None


class _timed:
    __slots__ = ("args", "e0")

    def __init__(self, name, key, alg_bytes=0.0, flops=0.0, survey=None):
        # survey: SURVEY.md 8d's count for the op (gather: every row read once + written once + CSR; GEMM: N (C_in + C_out) s),
        # which leaves out the extra operand streams of the fused forms that alg_bytes includes; default = alg_bytes
        self.args = (name, key, alg_bytes, flops, alg_bytes if survey is None else survey)

    def __enter__(self):
        if PROF is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if PROF is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            PROF.records.append(self.args + (self.e0, e1))
        return False


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def trace_marker():
    """Launch the library's marker kernel on the current stream (cuts a rocprofv3 trace to a region: bench.py)."""
    check(_lib.lib().ddmp_trace_marker(_stream()), "ddmp_trace_marker")


def copy_probe(src, dst, mode=0):
    """Streaming device copy on the library's own kernel (bench.py's yardstick; mode 1: nontemporal)."""
    assert src.is_cuda and dst.is_cuda and src.is_contiguous() and dst.is_contiguous()
    nbytes = src.numel() * src.element_size()
    assert dst.numel() * dst.element_size() >= nbytes
    check(_lib.lib().ddmp_copy_probe(_p(src), _p(dst), nbytes, int(mode), _stream()), "ddmp_copy_probe")


def copy_probe_rows(src, dst):
    """The copy in the gather's access pattern (64-row chunks, one 128-byte slab at a time) of a 2-D row-major tensor."""
    assert src.is_cuda and dst.is_cuda and src.is_contiguous() and dst.is_contiguous() and src.dim() == 2 and dst.shape == src.shape
    check(_lib.lib().ddmp_copy_probe_rows(_p(src), _p(dst), src.shape[0], src.shape[1] * src.element_size(), _stream()),
          "ddmp_copy_probe_rows")


def set_gemm_mode(mode: int):
    """6 = bf16x6 split MFMA (f32-class accuracy), 3 = bf16x3 (~2^-16), 0 = f32-input MFMA; 13 = f16x3 split MFMA
    in the row-panel kernels (f32-class accuracy, scaled operands: the ``scales=`` option of the GEMM calls), bf16x6 elsewhere."""
    check(_lib.lib().ddmp_set_gemm_mode(int(mode)), "ddmp_set_gemm_mode")


def get_gemm_mode() -> int:
    return int(_lib.lib().ddmp_get_gemm_mode())


def unfused(name: str) -> bool:
    """DDMP_UNFUSE=name[,name...]: fused routes switched OFF for A/B runs and the fused-vs-composed identity tests (the library
    reads the same variable: csrc/ddmp_common.h).  Engine names: stats, gather_bwd, bnbwd_l0, dgrad_red, tail, wprep, bf16_gemm,
    bf16_spmm_red, equal_width; library names: bnbwd_narrow, dgrad_red_narrow."""
    v = os.environ.get("DDMP_UNFUSE", "")
    return bool(v) and name in v.split(",")


def next_pending() -> int:
    """Diagnostic: bit mask of per-call options (1 BatchNorm coefficients, 2 GEMM scale slots, 4 prepared weight planes) still
    recorded for this host thread -- 0 between calls: an ``_o`` entry point sets its options and clears them around its own call."""
    return int(_lib.lib().ddmp_next_pending())


def gemm_forget_planes(planes=None):
    """The prepared-planes registry forgets ``planes`` (a buffer about to be freed; None: everything)."""
    _lib.lib().ddmp_gemm_forget_planes(_p(planes))


# ---------------------------------------------------------------------------------------- per-call options (ABI 3)
class _Opts(ctypes.Structure):
    """``ddmp_opts`` of include/ddmp_hip.h."""
    _fields_ = [("struct_size", ctypes.c_uint32), ("flags", ctypes.c_uint32), ("bn_n_total", ctypes.c_double),
                ("bn_C", ctypes.c_int32), ("bn_eps", ctypes.c_float), ("bn_momentum", ctypes.c_float), ("prime", ctypes.c_int32),
                ("bn_in", ctypes.c_void_p * 3), ("bn_out", ctypes.c_void_p * 6), ("slot_a", ctypes.c_void_p),
                ("slot_b", ctypes.c_void_p)]


OPT_BN_FWD, OPT_BN_BWD, OPT_SCALES, OPT_PREPARED = 1, 2, 4, 8


class BnFwd:
    """``bn=`` of a statistics-producing call (bn_stats / gemm_nt_stats / spmm_stats): the second stage of that call's
    reduction also writes what bn_prepare would (out4 rows = scale, shift, mean, rstd; running statistics updated) -- one
    launch less, bitwise the same coefficients.  Explicit per-call option (DDMP_OPT_BN_FWD)."""
    flag = OPT_BN_FWD

    def __init__(self, n_total, gamma, beta, out4, running=None, eps=BN_EPS, momentum=BN_MOMENTUM):
        rm, rv = (None, None) if running is None else running
        self.n_total, self.C, self.eps, self.momentum = float(n_total), gamma.numel(), eps, momentum
        self.ins = (gamma, beta, None)
        self.outs = (out4[0], out4[1], out4[2], out4[3], rm, rv)


class BnBwd:
    """``bn=`` of a call that produces the BatchNorm-backward reductions (bn_bwd_reduce / spmm_bnred / gemm_nn_bnred): its
    second stage also writes what bn_bwd_prepare would (dgamma, dbeta, c10 rows = c1, c0).  DDMP_OPT_BN_BWD."""
    flag = OPT_BN_BWD

    def __init__(self, n_total, bn4, dgamma, dbeta, c10):
        self.n_total, self.C, self.eps, self.momentum = float(n_total), dgamma.numel(), 0.0, 0.0
        self.ins = (bn4[0], bn4[2], bn4[3])
        self.outs = (dgamma, dbeta, c10[0], c10[1], None, None)


def _mk_opts(bn=None, scales=None, prepared=False, want_bn=False, want_scales=False):
    """-> (address | None, keep-alive) of the ddmp_opts block of ONE call.  scales = (slot_a, slot_b | None, prime).
    (want_bn / want_scales: which families of options the call accepts -- documentation at the call sites.)"""
    if bn is None and scales is None and not prepared:
        return None, None
    o = _Opts()
    o.struct_size = ctypes.sizeof(_Opts)
    flags = 0
    if bn is not None:
        flags |= bn.flag
        o.bn_n_total, o.bn_C, o.bn_eps, o.bn_momentum = bn.n_total, bn.C, bn.eps, bn.momentum
        for i, t in enumerate(bn.ins):
            o.bn_in[i] = None if t is None else t.data_ptr()
        for i, t in enumerate(bn.outs):
            o.bn_out[i] = None if t is None else t.data_ptr()
    if scales is not None:
        a, b, prime = scales
        flags |= OPT_SCALES
        o.slot_a = None if a is None else a.data_ptr()
        o.slot_b = None if b is None else b.data_ptr()
        o.prime = int(bool(prime))
    if prepared:
        flags |= OPT_PREPARED
    o.flags = flags
    return ctypes.c_void_p(ctypes.addressof(o)), (o, bn, scales)


def gemm_scales_roll(slots):
    """Once per iteration: the maxima recorded by this iteration's GEMM kernels become the next iteration's scales."""
    assert slots.dtype == torch.float32 and slots.is_contiguous() and slots.shape[-1] == 4
    check(_lib.lib().ddmp_gemm_scales_roll(_p(slots), slots.numel() // 4, _stream()), "ddmp_gemm_scales_roll")


def _chk(t, dtype=torch.float32, name="tensor"):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise DdmpError("%s must be a CUDA (ROCm) tensor: the HIP path has no CPU fallback" % name)
    if t.device.index != torch.cuda.current_device():
        # the library launches on the CURRENT device's stream and allocates graphs there: a tensor of another device
        # would be addressed from the wrong GPU.  The nets / trainers / losses enter `torch.cuda.device(their device)`
        # themselves; a bare ops call has to be made under the tensor's device.
        raise DdmpError("%s lives on cuda:%d but the current device is cuda:%d: wrap the call in "
                        "`with torch.cuda.device(t.device):`" % (name, t.device.index, torch.cuda.current_device()))
    if t.dtype != dtype:
        raise DdmpError("%s must be %s, got %s" % (name, dtype, t.dtype))
    return t


F32, BF16 = 0, 1            # DDMP_F32 / DDMP_BF16 of include/ddmp_hip.h (feature dtype of the dtype-tagged entry points)


def _dt(t) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise DdmpError("feature tensors are float32 or bfloat16, got %s" % t.dtype)


def _mat(t, name="matrix", like=None):
    """2-D float32 / bfloat16 CUDA tensor with unit inner stride -> (tensor, leading dimension).  ``like``: a tensor
    whose dtype it has to share (all feature operands of one call)."""
    if isinstance(t, torch.Tensor) and t.dtype == torch.bfloat16:
        _chk(t, torch.bfloat16, name)
    else:
        _chk(t, torch.float32, name)
    if like is not None and like.dtype != t.dtype:
        raise DdmpError("%s is %s but the call's other feature operand is %s" % (name, t.dtype, like.dtype))
    if t.dim() != 2 or t.stride(1) != 1:
        raise DdmpError("%s must be 2-D with contiguous rows" % name)
    return t, (t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0)))


class Workspace:
    """Grow-only scratch buffer for split-K partials, pre-split weights and column reductions: one per
    (device, stream, host thread) -- logical ranks emulated as threads on one stream must not share it."""

    _bufs = {}

    @classmethod
    def get(cls, nbytes: int, device) -> torch.Tensor:
        key = (torch.device(device).index, torch.cuda.current_stream().cuda_stream, threading.get_ident())
        buf = cls._bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
            cls._bufs[key] = buf
        return buf


# ---------------------------------------------------------------------------------------- graph
class Graph:
    """CSR of A + I with D^-1/2 on the device (``ddmp_graph``).  Built once per edge_index
    (the reference's GCNConv re-normalises on every call, cached=False)."""

    def __init__(self, handle, n_rows, n_cols, nnz):
        self._h = handle
        self.n_rows, self.n_cols, self.nnz = n_rows, n_cols, nnz
        self._fin = weakref.finalize(self, _lib.lib().ddmp_graph_destroy, handle)
        a, b, c, d = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int()
        check(_lib.lib().ddmp_graph_info(handle, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(d)), "ddmp_graph_info")
        self.max_row_nnz = int(d.value)                         # entries of the longest row (self loop included)

    @property
    def handle(self):
        return self._h

    @classmethod
    def from_edge_index(cls, edge_index: torch.Tensor, num_nodes: int) -> "Graph":
        if edge_index.dim() != 2 or edge_index.shape[0] != 2 or edge_index.dtype != torch.int64:
            raise DdmpError("edge_index must be a [2, nnz] int64 tensor")
        ei = edge_index.contiguous()
        h = ctypes.c_void_p()
        st = _lib.lib().ddmp_graph_create(int(num_nodes), int(ei.shape[1]), _p(ei), 1 if ei.is_cuda else 0,
                                          ctypes.byref(h))
        check(st, "ddmp_graph_create")
        return cls(h, int(num_nodes), int(num_nodes), int(ei.shape[1]) + int(num_nodes))

    @classmethod
    def from_csr_host(cls, rowptr: np.ndarray, col: np.ndarray, dinv: np.ndarray, n_cols: int, rows=None) -> "Graph":
        """``rows`` = (row0, row1): only these rows of the tables, as a graph of their own -- output row i is node row0 + i, the
        columns keep the numbering of the whole local graph (same X, the output tensor starts at row row0): the interior /
        boundary halves of a partitioned graph (dist.py)."""
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        col = np.ascontiguousarray(col, dtype=np.int32)
        dinv = np.ascontiguousarray(dinv, dtype=np.float32)
        n_rows = len(rowptr) - 1
        h = ctypes.c_void_p()
        if rows is not None:
            r0, r1 = int(rows[0]), int(rows[1])
            st = _lib.lib().ddmp_graph_create_csr_rows_host(n_rows, int(n_cols), rowptr.ctypes.data, col.ctypes.data,
                                                            dinv.ctypes.data, r0, r1, ctypes.byref(h))
            check(st, "ddmp_graph_create_csr_rows_host")
            return cls(h, r1 - r0, int(n_cols), int(rowptr[r1] - rowptr[r0]))
        st = _lib.lib().ddmp_graph_create_csr_host(n_rows, int(n_cols), rowptr.ctypes.data, col.ctypes.data,
                                                   dinv.ctypes.data, ctypes.byref(h))
        check(st, "ddmp_graph_create_csr_host")
        return cls(h, n_rows, int(n_cols), int(rowptr[-1]))


_graph_cache = {}


def graph_for(edge_index: torch.Tensor, num_nodes: int) -> Graph:
    """Graph of a static mesh, cached on the IDENTITY of the edge_index tensor (weak reference + in-place
    version counter).  A data_ptr key would be wrong: a freed tensor's address is reused by other meshes.
    A caller that builds a fresh edge_index tensor on every call (as ``data.edge_index.to(device)`` does when
    the dataset lives on the host) gets a correct but rebuilt graph each time -- keep the tensor."""
    key = id(edge_index)
    hit = _graph_cache.get(key)
    if hit is not None:
        ref, version, n, g = hit
        if ref() is edge_index and version == edge_index._version and n == int(num_nodes):
            return g
    for k in [k for k, v in _graph_cache.items() if v[0]() is None]:
        del _graph_cache[k]
    g = Graph.from_edge_index(edge_index, num_nodes)
    _graph_cache[key] = (weakref.ref(edge_index), edge_index._version, int(num_nodes), g)
    return g


def csr_build_host(edge_index: np.ndarray, num_nodes: int):
    """Host CSR (no GPU needed): -> rowptr int32[n+1], col int32[nnz'], dinv f32[n]."""
    ei = np.ascontiguousarray(edge_index, dtype=np.int64)
    nnz = ei.shape[1]
    rowptr = np.zeros(num_nodes + 1, np.int32)
    col = np.zeros(nnz + num_nodes, np.int32)
    dinv = np.zeros(num_nodes, np.float32)
    cap = ctypes.c_int64(nnz + num_nodes)
    st = _lib.lib().ddmp_csr_build_host(num_nodes, nnz, ei.ctypes.data, rowptr.ctypes.data, col.ctypes.data,
                                        dinv.ctypes.data, ctypes.byref(cap))
    check(st, "ddmp_csr_build_host")
    return rowptr, col[:cap.value].copy(), dinv


def bfs_order_host(rowptr: np.ndarray, col: np.ndarray) -> np.ndarray:
    n = len(rowptr) - 1
    order = np.zeros(n, np.int32)
    st = _lib.lib().ddmp_csr_bfs_order_host(n, rowptr.ctypes.data, col.ctypes.data, order.ctypes.data)
    check(st, "ddmp_csr_bfs_order_host")
    return order


def rcb_order_host(points: np.ndarray, leaf: int = 64) -> np.ndarray:
    """Recursive-coordinate-bisection node order (new id -> old id) of ``points`` [n,3]: leaves of ``leaf`` consecutive
    ids are compact patches of the surface (ddmp_rcb_order_host; host code, no GPU needed)."""
    p = np.ascontiguousarray(points, dtype=np.float64)
    if p.ndim != 2 or p.shape[1] != 3:
        raise DdmpError("rcb_order_host: points must be [n,3]")
    order = np.zeros(len(p), np.int32)
    check(_lib.lib().ddmp_rcb_order_host(len(p), p.ctypes.data, int(leaf), order.ctypes.data), "ddmp_rcb_order_host")
    return order


# ---------------------------------------------------------------------------------------- kernels
def spmm(g: Graph, x, out=None, bias=None, pro=None, slope=SLOPE):
    """out[i] = dinv_i * sum_j dinv_j f(x[j]) (+bias); x has g.n_cols rows, out g.n_rows rows."""
    x, ldx = _mat(x, "x")
    if x.shape[0] < g.n_cols:
        raise DdmpError("x has %d rows, graph references %d nodes" % (x.shape[0], g.n_cols))
    C = x.shape[1]
    if out is None:
        out = torch.empty((g.n_rows, C), dtype=x.dtype, device=x.device)
    out, ldy = _mat(out, "out", x)
    ps, psh = (None, None) if pro is None else pro
    es = x.element_size()
    # algorithmic bytes: every feature row read once + written once, int32 col ids, rowptr, dinv
    alg = 2.0 * g.n_rows * C * es + 4.0 * g.nnz + 4.0 * (g.n_rows + 1) + 4.0 * g.n_rows
    with _timed("spmm", (C, int(round(g.nnz / max(g.n_rows, 1)))), alg, 2.0 * g.nnz * C):      # key: width, CSR entries per row
        st = _lib.lib().ddmp_spmm(g.handle, _p(x), ldx, _p(out), ldy, C, _dt(x), _p(bias), _p(ps), _p(psh), slope,
                                  _stream())
    check(st, "ddmp_spmm")
    return out


def spmm_stats_supported(C, dtype=torch.float32):
    """Does spmm_stats take its fused form for C channels?"""
    if dtype == torch.bfloat16:
        return C % 16 == 0 and C <= 1024
    return dtype == torch.float32 and bool(_lib.lib().ddmp_spmm_stats_supported(int(C)))


def spmm_stats(g: Graph, x, out, ref, sums, bias=None, pro=None, slope=SLOPE, bn=None):
    """out = spmm(g, x) (+bias, prologue) and sums (float64 [2C]) = bn_stats(out) from the same kernel: the statistics are
    summed around ``ref`` (float32 [C], close to the column means -- the previous iteration's batch means; zeros are valid)
    in float32 over 16 rows at a time, in float64 from there on (ddmp_spmm_stats)."""
    x, ldx = _mat(x, "x")
    out, ldy = _mat(out, "out", x)
    C = x.shape[1]
    ps, psh = (None, None) if pro is None else pro
    L = _lib.lib()
    ws = Workspace.get(max(L.ddmp_spmm_bnred_ws_bytes(g.n_rows, C, _dt(x)), L.ddmp_colreduce_workspace_bytes(g.n_rows, C)), x.device)
    es = x.element_size()
    alg = float(es) * (g.n_cols + g.n_rows) * C + 4.0 * g.nnz + 8.0 * g.n_rows
    with _timed("spmm", (C, int(round(g.nnz / max(g.n_rows, 1)))), alg, 2.0 * g.nnz * C, survey=2.0 * g.n_rows * C * es + 4.0 * g.nnz + 4.0 * (g.n_rows + 1) + 4.0 * g.n_rows):
        o, keep = _mk_opts(bn, want_bn=True)
        st = L.ddmp_spmm_stats_o(g.handle, _p(x), ldx, _p(out), ldy, C, _dt(x), _p(bias), _p(ps), _p(psh), slope,
                                 _p(_chk(ref, torch.float32, "ref")), _p(sums), _p(ws), ws.numel(), _stream(), o)
    check(st, "ddmp_spmm_stats")
    return out


def spmm_bnred(g: Graph, x, out, yp, bn4, sums2, slope=SLOPE, bn=None):
    """out = spmm(g, x) (a gradient dZ) and sums2 = bn_bwd_reduce(out, yp, bn4) from the same kernel."""
    x, ldx = _mat(x, "x")
    out, ldy = _mat(out, "out", x)
    yp, ldyp = _mat(yp, "yp", x)
    C = x.shape[1]
    L = _lib.lib()
    ws = Workspace.get(L.ddmp_spmm_bnred_ws_bytes(g.n_rows, C, _dt(x)), x.device)
    alg = 3.0 * g.n_rows * C * x.element_size() + 4.0 * g.nnz + 4.0 * (g.n_rows + 1) + 4.0 * g.n_rows
    with _timed("spmm", (C, int(round(g.nnz / max(g.n_rows, 1)))), alg, 2.0 * g.nnz * C, survey=2.0 * g.n_rows * C * x.element_size() + 4.0 * g.nnz + 4.0 * (g.n_rows + 1) + 4.0 * g.n_rows):
        o, keep = _mk_opts(bn, want_bn=True)
        st = L.ddmp_spmm_bnred_o(g.handle, _p(x), ldx, _p(out), ldy, C, _dt(x), _p(yp), ldyp, _p(bn4[0]), _p(bn4[1]),
                                 _p(bn4[2]), _p(bn4[3]), slope, _p(sums2), _p(ws), ws.numel(), _stream(), o)
    check(st, "ddmp_spmm_bnred")
    return out


def spmm_bnbwd_supported(C):
    return bool(_lib.lib().ddmp_spmm_bnbwd_supported(int(C)))


def spmm_bnbwd(g: Graph, dz, yb, bn4, c10, out, slope=SLOPE):
    """out[:n_rows] = A_hat @ dY with dY = BatchNorm+LeakyReLU backward of (dz, yb) rebuilt on the gather (what
    bn_bwd_apply would have written: a*dz*lrelu'(a*yb+b) + c1*yb + c0)."""
    dz, lddz = _mat(dz, "dz")
    yb, ldyb = _mat(yb, "yb", dz)
    out, ldo = _mat(out, "out", dz)
    C = dz.shape[1]
    assert dz.shape[0] >= g.n_cols and yb.shape[0] >= g.n_cols and out.shape[0] >= g.n_rows and yb.shape[1] == C
    es = dz.element_size()
    with _timed("spmm", (C, int(round(g.nnz / max(g.n_rows, 1)))), es * (2.0 * g.n_cols + g.n_rows) * C + 4.0 * g.nnz + 8.0 * g.n_rows,
                2.0 * g.nnz * C, survey=2.0 * g.n_rows * C * es + 4.0 * g.nnz + 4.0 * (g.n_rows + 1) + 4.0 * g.n_rows):
        st = _lib.lib().ddmp_spmm_bnbwd(g.handle, _p(dz), lddz, _p(yb), ldyb, _p(out), ldo, C, _dt(dz), _p(bn4[0]),
                                        _p(bn4[1]), _p(c10[0]), _p(c10[1]), slope, _stream())
    check(st, "ddmp_spmm_bnbwd")
    return out


class WeightPlan:
    """The argument arrays of one ddmp_gemm_prepare_weights call, built once (the weight matrices are views of a parameter
    arena whose address does not change between iterations); ``run()`` re-splits the current weights."""

    def __init__(self, items, n_rows, scratch):
        n = len(items)
        assert scratch.dtype == torch.float32 and scratch.numel() >= 8 * n
        self._keep = (items, scratch)
        self.n, self.n_rows, self.scratch = n, int(n_rows), scratch
        arrs = ((ctypes.c_void_p * n)(*[w.data_ptr() for w, _, _, _ in items]),
                (ctypes.c_int64 * n)(*[w.stride(0) for w, _, _, _ in items]),
                (ctypes.c_int * n)(*[w.shape[0] for w, _, _, _ in items]),
                (ctypes.c_int * n)(*[w.shape[1] for w, _, _, _ in items]),
                (ctypes.c_int * n)(*[int(f) for _, f, _, _ in items]),
                (ctypes.c_int * n)(*[int(bool(h)) for _, _, h, _ in items]),
                (ctypes.c_void_p * n)(*[p.data_ptr() for _, _, _, p in items]),
                (ctypes.c_size_t * n)(*[p.numel() for _, _, _, p in items]))
        self._arrs = arrs
        self._args = [ctypes.cast(a, ctypes.c_void_p) for a in arrs]

    def run(self):
        check(_lib.lib().ddmp_gemm_prepare_weights(self.n, *self._args, self.n_rows, _p(self.scratch), _stream()),
              "ddmp_gemm_prepare_weights")


def gemm_prepare_weights(items, n_rows, scratch):
    """items: [(w [M,K] float32, form 0 forward | 1 dgrad, has_pro, planes uint8 buffer)]: split all these weight matrices
    into the planes their products over ``n_rows`` rows want, in two launches (ddmp_gemm_prepare_weights).  The GEMM calls
    then take ``wplanes=planes``.  ``scratch``: float32 [>= 8 len(items)]."""
    WeightPlan(items, n_rows, scratch).run()


def gemm_rows_workspace_bytes(K, M):
    return int(_lib.lib().ddmp_gemm_rows_workspace_bytes(int(K), int(M)))


def _wws(wplanes, nbytes, device):
    """The weight-plane workspace of a GEMM call -> (buffer, prepared): the caller's prepared buffer (the call is then given
    DDMP_OPT_PREPARED) or scratch."""
    if wplanes is not None:
        if wplanes.numel() < nbytes:
            raise DdmpError("wplanes: %d bytes, the product needs %d" % (wplanes.numel(), nbytes))
        return wplanes, True
    return Workspace.get(nbytes, device), False


def gemm_nt(a, w, out=None, bias=None, pro=None, slope=SLOPE, n_rows=None, wplanes=None, scales=None):
    """out[n,M] = f(a[n,K]) @ w[M,K]^T (+bias)."""
    a, lda = _mat(a, "a")
    w, ldw = _mat(_chk(w, torch.float32, "w"), "w")
    n = a.shape[0] if n_rows is None else n_rows
    K, M = a.shape[1], w.shape[0]
    if w.shape[1] != K:
        raise DdmpError("gemm_nt: inner dimensions differ (%d vs %d)" % (K, w.shape[1]))
    if out is None:
        out = torch.empty((n, M), dtype=a.dtype, device=a.device)
    out, ldy = _mat(out, "out", a)
    ps, psh = (None, None) if pro is None else pro
    L = _lib.lib()
    ws, prep = _wws(wplanes if a.dtype == torch.float32 else None, L.ddmp_gemm_rows_ws_bytes(K, M, _dt(a)), a.device)
    es = a.element_size()
    with _timed("gemm_nt", (K, M), es * n * (K + M) + 4.0 * K * M, 2.0 * n * K * M, survey=es * n * (K + M)):
        o, keep = _mk_opts(None, scales, prep, want_scales=True)
        st = L.ddmp_gemm_nt_o(_p(a), lda, _p(w), ldw, _p(out), ldy, n, K, M, _dt(a), _p(bias), _p(ps), _p(psh),
                              slope, _p(ws), ws.numel(), _stream(), o)
    check(st, "ddmp_gemm_nt")
    return out


def gemm_nt_stats(a, w, sums, out=None, bias=None, pro=None, slope=SLOPE, n_rows=None, wplanes=None, scales=None, bn=None):
    """gemm_nt that also fills ``sums`` (float64 [2M]) with the column sums of out and out^2 (= bn_stats(out)):
    produced in the row-panel kernel's epilogue where that kernel runs, by a separate pass otherwise."""
    if a.dtype == torch.bfloat16:
        # bf16 features: statistics from the row-register kernel's epilogue where that kernel runs (round 3), else the
        # GEMM followed by the streaming statistics pass
        n = a.shape[0] if n_rows is None else n_rows
        K, M = a.shape[1], w.shape[0]
        L = _lib.lib()
        if L.ddmp_gemm_fused_bf16_supported(int(M), int(K), int(n)) & 1:
            a, lda = _mat(a, "a")
            w, ldw = _mat(_chk(w, torch.float32, "w"), "w")
            if out is None:
                out = torch.empty((n, M), dtype=a.dtype, device=a.device)
            out, ldy = _mat(out, "out", a)
            ps, psh = (None, None) if pro is None else pro
            nb = (L.ddmp_gemm_rows_ws_bytes(K, M, _dt(a)) + 255) // 256 * 256
            sb = L.ddmp_gemm_nt_stats_bf16_workspace_bytes(n, M)
            ws = Workspace.get(nb + sb, a.device)
            with _timed("gemm_nt", (K, M), 2.0 * n * (K + M) + 4.0 * K * M, 2.0 * n * K * M, survey=2.0 * n * (K + M)):
                o, keep = _mk_opts(bn, want_bn=True)
                st = L.ddmp_gemm_nt_stats_bf16_o(_p(a), lda, _p(w), ldw, _p(out), ldy, n, K, M, _p(bias), _p(ps), _p(psh), slope,
                                                 _p(sums), _p(ws), nb, ws.data_ptr() + nb, ws.numel() - nb, _stream(), o)
            check(st, "ddmp_gemm_nt_stats_bf16")
            return out
        out = gemm_nt(a, w, out=out, bias=bias, pro=pro, slope=slope, n_rows=n_rows)
        bn_stats(out, sums=sums, n_rows=n_rows, bn=bn)
        return out
    a, lda = _mat(a, "a")
    w, ldw = _mat(w, "w")
    n = a.shape[0] if n_rows is None else n_rows
    K, M = a.shape[1], w.shape[0]
    if w.shape[1] != K:
        raise DdmpError("gemm_nt: inner dimensions differ (%d vs %d)" % (K, w.shape[1]))
    if out is None:
        out = torch.empty((n, M), dtype=torch.float32, device=a.device)
    out, ldy = _mat(out, "out")
    ps, psh = (None, None) if pro is None else pro
    L = _lib.lib()
    nb = (L.ddmp_gemm_rows_workspace_bytes(K, M) + 255) // 256 * 256
    sb = L.ddmp_gemm_nt_stats_workspace_bytes(n, M)
    ws = Workspace.get(nb + sb, a.device)
    wp = ws if wplanes is None else _wws(wplanes, L.ddmp_gemm_rows_workspace_bytes(K, M), a.device)[0]
    with _timed("gemm_nt", (K, M), 4.0 * n * (K + M) + 4.0 * K * M, 2.0 * n * K * M, survey=4.0 * n * (K + M)):
        o, keep = _mk_opts(bn, scales, wplanes is not None, want_bn=True, want_scales=True)
        st = L.ddmp_gemm_nt_stats_f32_o(_p(a), lda, _p(w), ldw, _p(out), ldy, n, K, M, _p(bias), _p(ps), _p(psh), slope,
                                        _p(sums), _p(wp), nb if wplanes is None else wp.numel(), ws.data_ptr() + nb,
                                        ws.numel() - nb, _stream(), o)
    check(st, "ddmp_gemm_nt_stats_f32")
    return out


def gemm_nn(a, w, out=None, n_rows=None, wplanes=None, scales=None):
    """out[n,K] = a[n,M] @ w[M,K]."""
    a, lda = _mat(a, "a")
    w, ldw = _mat(_chk(w, torch.float32, "w"), "w")
    n = a.shape[0] if n_rows is None else n_rows
    M, K = w.shape
    if a.shape[1] != M:
        raise DdmpError("gemm_nn: inner dimensions differ")
    if out is None:
        out = torch.empty((n, K), dtype=a.dtype, device=a.device)
    out, ldy = _mat(out, "out", a)
    L = _lib.lib()
    ws, prep = _wws(wplanes if a.dtype == torch.float32 else None, L.ddmp_gemm_rows_ws_bytes(K, M, _dt(a)), a.device)
    es = a.element_size()
    with _timed("gemm_nn", (M, K), es * n * (K + M) + 4.0 * K * M, 2.0 * n * K * M, survey=es * n * (K + M)):
        o, keep = _mk_opts(None, scales, prep, want_scales=True)
        st = L.ddmp_gemm_nn_o(_p(a), lda, _p(w), ldw, _p(out), ldy, n, M, K, _dt(a), _p(ws), ws.numel(), _stream(), o)
    check(st, "ddmp_gemm_nn")
    return out


def gemm_nn_bnred_supported(M, K, n_rows, dtype=torch.float32):
    """Does gemm_nn_bnred exist for a dgrad M -> K over n_rows rows (row-register kernel, float32 features; the bfloat16 twin
    of round 5 measured no gain inside the step and left the library in round 6: experiments/r05/)?"""
    return dtype == torch.float32 and bool(_lib.lib().ddmp_gemm_nn_bnred_supported(int(M), int(K), int(n_rows)))


def gemm_nn_bnred(a, w, yp, bn4, sums, out=None, slope=SLOPE, n_rows=None, wplanes=None, scales=None, bn=None):
    """out[n,K] = a[n,M] @ w[M,K] AND sums (float64 [2K]) = bn_bwd_reduce(out, yp, bn4): the BatchNorm-backward column
    reductions of `out` as the gradient behind the previous layer's BatchNorm+LeakyReLU (yp: that layer's conv output),
    from the GEMM epilogue -- one read of yp instead of a pass over out and yp."""
    a, lda = _mat(a, "a")
    w, ldw = _mat(_chk(w, torch.float32, "w"), "w")
    yp, ldyp = _mat(yp, "yp", a)
    n = a.shape[0] if n_rows is None else n_rows
    M, K = w.shape
    if out is None:
        out = torch.empty((n, K), dtype=a.dtype, device=a.device)
    out, ldo = _mat(out, "out", a)
    L = _lib.lib()
    if a.dtype != torch.float32:
        raise DdmpError("gemm_nn_bnred: float32 features only")
    nb = (L.ddmp_gemm_rows_workspace_bytes(K, M) + 255) // 256 * 256
    sb = L.ddmp_gemm_nt_stats_workspace_bytes(n, K)
    ws = Workspace.get(nb + sb, a.device)
    wp = ws if wplanes is None else _wws(wplanes, L.ddmp_gemm_rows_workspace_bytes(K, M), a.device)[0]
    with _timed("gemm_nn", (M, K), 4.0 * n * (2 * K + M) + 4.0 * K * M, 2.0 * n * K * M, survey=4.0 * n * (K + M)):
        o, keep = _mk_opts(bn, scales, wplanes is not None, want_bn=True, want_scales=True)
        st = L.ddmp_gemm_nn_bnred_f32_o(_p(a), lda, _p(w), ldw, _p(out), ldo, n, M, K, _p(yp), ldyp, _p(bn4[0]), _p(bn4[1]),
                                        _p(bn4[2]), _p(bn4[3]), slope, _p(sums), _p(wp), nb if wplanes is None else wp.numel(),
                                        ws.data_ptr() + nb, ws.numel() - nb, _stream(), o)
    check(st, "ddmp_gemm_nn_bnred_f32")
    return out


def gemm_tn(g, z, out=None, pro=None, slope=SLOPE, n_rows=None, scales=None):
    """out[M,K] = g[n,M]^T @ f(z[n,K])  (weight gradient)."""
    g, ldg = _mat(g, "g")
    z, ldz = _mat(z, "z", g)
    n = g.shape[0] if n_rows is None else n_rows
    M, K = g.shape[1], z.shape[1]
    if out is None:
        out = torch.empty((M, K), dtype=torch.float32, device=g.device)
    out, ldo = _mat(_chk(out, torch.float32, "out"), "out")
    L = _lib.lib()
    need = L.ddmp_gemm_tn_ws_bytes(n, M, K, _dt(g))
    ws = Workspace.get(need, g.device)
    ps, psh = (None, None) if pro is None else pro
    es = g.element_size()
    with _timed("gemm_tn", (M, K), es * n * (K + M) + 4.0 * K * M, 2.0 * n * K * M, survey=es * n * (K + M)):
        o, keep = _mk_opts(None, scales, want_scales=True)
        st = L.ddmp_gemm_tn_o(_p(g), ldg, _p(z), ldz, _p(out), ldo, n, M, K, _dt(g), _p(ps), _p(psh), slope, _p(ws),
                              ws.numel(), _stream(), o)
    check(st, "ddmp_gemm_tn")
    return out


def gemm_bnbwd_supported(cout, cin, n_rows, dtype=torch.float32):
    """Do the fused BatchNorm-backward GEMMs exist for a layer cin -> cout over n_rows rows in the current GEMM mode?
    (bf16 features: on the row-register kernel, round 3.)"""
    if dtype == torch.bfloat16:
        return bool(_lib.lib().ddmp_gemm_fused_bf16_supported(int(cout), int(cin), int(n_rows)) & 2)
    if dtype != torch.float32:
        return False
    return bool(_lib.lib().ddmp_gemm_bnbwd_supported(int(cout), int(cin), int(n_rows)))


def gemm_tn_bnbwd_supported(cout, cin, n_rows, dtype=torch.float32):
    """Does gemm_tn_bnbwd exist for cin -> cout over n_rows rows (float32 features; the wgrad alone)?"""
    return dtype == torch.float32 and bool(_lib.lib().ddmp_gemm_tn_bnbwd_supported(int(cout), int(cin), int(n_rows)))


def rows_gather(src, idx, out=None, n_rows=None):
    """out[r] = src[idx[r]] (halo packing on the library's kernel; float32 / bfloat16 rows of 16-byte multiples)."""
    src, lds = _mat(src, "src")
    n = idx.numel() if n_rows is None else n_rows
    if out is None:
        out = torch.empty((n, src.shape[1]), dtype=src.dtype, device=src.device)
    out, ldo = _mat(out, "out", src)
    check(_lib.lib().ddmp_rows_gather(_p(src), lds, _p(_chk(idx, torch.int64, "idx")), n, src.shape[1], _dt(src), _p(out), ldo, 0,
                                      _stream()), "ddmp_rows_gather")
    return out


def to_bf16(src, dst=None):
    """float32 -> bfloat16 (round to nearest even) on the library's kernel."""
    src = _chk(src.contiguous(), torch.float32, "src")
    if dst is None:
        dst = torch.empty(src.shape, dtype=torch.bfloat16, device=src.device)
    check(_lib.lib().ddmp_f32_to_bf16(_p(src), _p(dst), src.numel(), _stream()), "ddmp_f32_to_bf16")
    return dst


def gemm_nn_bnbwd(dz, yb, w, bn4, c10, out=None, slope=SLOPE, n_rows=None, wplanes=None, scales=None):
    """out[n,K] = dY[n,M] @ w[M,K] with dY = BatchNorm+LeakyReLU backward of (dz, yb) computed on the operand load
    (what bn_bwd_apply would have written: a*dz*lrelu'(a*yb+b) + c1*yb + c0)."""
    if dz.dtype == torch.bfloat16:
        dz, lddz = _mat(dz, "dz")
        yb, ldyb = _mat(yb, "yb", dz)
        w, ldw = _mat(_chk(w, torch.float32, "w"), "w")
        n = dz.shape[0] if n_rows is None else n_rows
        M, K = w.shape
        if out is None:
            out = torch.empty((n, K), dtype=dz.dtype, device=dz.device)
        out, ldo = _mat(out, "out", dz)
        L = _lib.lib()
        ws = Workspace.get(L.ddmp_gemm_rows_ws_bytes(K, M, _dt(dz)), dz.device)
        with _timed("gemm_nn", (M, K), 2.0 * n * (K + 2 * M) + 4.0 * K * M, 2.0 * n * K * M, survey=2.0 * n * (K + M)):
            st = L.ddmp_gemm_nn_bnbwd_bf16(_p(dz), lddz, _p(yb), ldyb, _p(w), ldw, _p(out), ldo, n, M, K, _p(bn4[0]), _p(bn4[1]),
                                           _p(c10[0]), _p(c10[1]), slope, _p(ws), ws.numel(), _stream())
        check(st, "ddmp_gemm_nn_bnbwd_bf16")
        return out
    dz, lddz = _mat(dz, "dz")
    yb, ldyb = _mat(yb, "yb")
    w, ldw = _mat(w, "w")
    n = dz.shape[0] if n_rows is None else n_rows
    M, K = w.shape
    if out is None:
        out = torch.empty((n, K), dtype=torch.float32, device=dz.device)
    out, ldo = _mat(out, "out")
    L = _lib.lib()
    ws, prep = _wws(wplanes, L.ddmp_gemm_rows_workspace_bytes(K, M), dz.device)
    with _timed("gemm_nn", (M, K), 4.0 * n * (K + 2 * M) + 4.0 * K * M, 2.0 * n * K * M, survey=4.0 * n * (K + M)):
        o, keep = _mk_opts(None, scales, prep, want_scales=True)
        st = L.ddmp_gemm_nn_bnbwd_f32_o(_p(dz), lddz, _p(yb), ldyb, _p(w), ldw, _p(out), ldo, n, M, K, _p(bn4[0]), _p(bn4[1]),
                                        _p(c10[0]), _p(c10[1]), slope, _p(ws), ws.numel(), _stream(), o)
    check(st, "ddmp_gemm_nn_bnbwd_f32")
    return out


def gemm_tn_bnbwd(dz, yb, z, bn4, c10, out=None, pro=None, slope=SLOPE, n_rows=None, scales=None):
    """out[M,K] = dY^T @ f(z) with dY as in gemm_nn_bnbwd."""
    if dz.dtype == torch.bfloat16:
        dz, lddz = _mat(dz, "dz")
        yb, ldyb = _mat(yb, "yb", dz)
        z, ldz = _mat(z, "z", dz)
        n = dz.shape[0] if n_rows is None else n_rows
        M, K = yb.shape[1], z.shape[1]
        if out is None:
            out = torch.empty((M, K), dtype=torch.float32, device=dz.device)
        out, ldo = _mat(_chk(out, torch.float32, "out"), "out")
        L = _lib.lib()
        ws = Workspace.get(L.ddmp_gemm_tn_ws_bytes(n, M, K, _dt(dz)), dz.device)
        ps, psh = (None, None) if pro is None else pro
        with _timed("gemm_tn", (M, K), 2.0 * n * (K + 2 * M) + 4.0 * K * M, 2.0 * n * K * M, survey=2.0 * n * (K + M)):
            st = L.ddmp_gemm_tn_bnbwd_bf16(_p(dz), lddz, _p(yb), ldyb, _p(z), ldz, _p(out), ldo, n, M, K, _p(bn4[0]), _p(bn4[1]),
                                           _p(c10[0]), _p(c10[1]), _p(ps), _p(psh), slope, _p(ws), ws.numel(), _stream())
        check(st, "ddmp_gemm_tn_bnbwd_bf16")
        return out
    dz, lddz = _mat(dz, "dz")
    yb, ldyb = _mat(yb, "yb")
    z, ldz = _mat(z, "z")
    n = dz.shape[0] if n_rows is None else n_rows
    M, K = yb.shape[1], z.shape[1]
    if out is None:
        out = torch.empty((M, K), dtype=torch.float32, device=dz.device)
    out, ldo = _mat(out, "out")
    L = _lib.lib()
    ws = Workspace.get(L.ddmp_gemm_tn_workspace_bytes(n, M, K), dz.device)
    ps, psh = (None, None) if pro is None else pro
    with _timed("gemm_tn", (M, K), 4.0 * n * (K + 2 * M) + 4.0 * K * M, 2.0 * n * K * M, survey=4.0 * n * (K + M)):
        o, keep = _mk_opts(None, scales, want_scales=True)
        st = L.ddmp_gemm_tn_bnbwd_f32_o(_p(dz), lddz, _p(yb), ldyb, _p(z), ldz, _p(out), ldo, n, M, K, _p(bn4[0]), _p(bn4[1]),
                                        _p(c10[0]), _p(c10[1]), _p(ps), _p(psh), slope, _p(ws), ws.numel(), _stream(), o)
    check(st, "ddmp_gemm_tn_bnbwd_f32")
    return out


def _colws(n, C, device):
    need = _lib.lib().ddmp_colreduce_workspace_bytes(n, C)
    if need == 0:
        raise DdmpError("column reduction: unsupported width %d (power of two in [8,1024])" % C)
    return Workspace.get(need, device)


def bn_stats(y, sums=None, n_rows=None, bn=None):
    """-> float64 [2C] = (column sums, column sums of squares) of y[:n_rows]."""
    y, ldy = _mat(y, "y")
    n = y.shape[0] if n_rows is None else n_rows
    C = y.shape[1]
    if sums is None:
        sums = torch.empty(2 * C, dtype=torch.float64, device=y.device)
    ws = _colws(n, C, y.device)
    with _timed("bn_stats", C, float(y.element_size()) * n * C):
        o, keep = _mk_opts(bn, want_bn=True)
        st = _lib.lib().ddmp_bn_stats_o(_p(y), ldy, n, C, _dt(y), _p(sums), _p(ws), ws.numel(), _stream(), o)
    check(st, "ddmp_bn_stats")
    return sums


def bn_prepare(sums, n_total, gamma, beta, out4, running=None, eps=BN_EPS, momentum=BN_MOMENTUM):
    """out4: float32 [4,C] rows = scale, shift, mean, rstd (written)."""
    C = gamma.numel()
    rm, rv = (None, None) if running is None else running
    st = _lib.lib().ddmp_bn_prepare_f32(_p(sums), float(n_total), C, _p(gamma), _p(beta), eps, momentum,
                                        _p(out4[0]), _p(out4[1]), _p(out4[2]), _p(out4[3]), _p(rm), _p(rv), _stream())
    check(st, "ddmp_bn_prepare_f32")
    return out4


def bn_lrelu_apply(y, scale, shift, out=None, slope=SLOPE):
    y, ldy = _mat(y, "y")
    if out is None:
        out = torch.empty_like(y)
    out, ldz = _mat(out, "out", y)
    fn = _lib.lib().ddmp_bn_lrelu_apply_bf16 if y.dtype == torch.bfloat16 else _lib.lib().ddmp_bn_lrelu_apply_f32
    st = fn(_p(y), ldy, _p(out), ldz, y.shape[0], y.shape[1], _p(scale), _p(shift), slope, _stream())
    check(st, "ddmp_bn_lrelu_apply")
    return out


def bn_bwd_reduce(dz, y, bn4, sums2=None, slope=SLOPE, n_rows=None, bn=None):
    dz, lddz = _mat(dz, "dz")
    y, ldy = _mat(y, "y", dz)
    n = y.shape[0] if n_rows is None else n_rows
    C = y.shape[1]
    if sums2 is None:
        sums2 = torch.empty(2 * C, dtype=torch.float64, device=y.device)
    ws = _colws(n, C, y.device)
    with _timed("bn_bwd_reduce", C, 2.0 * y.element_size() * n * C):
        o, keep = _mk_opts(bn, want_bn=True)
        st = _lib.lib().ddmp_bn_bwd_reduce_o(_p(dz), lddz, _p(y), ldy, n, C, _dt(y), _p(bn4[0]), _p(bn4[1]), _p(bn4[2]),
                                             _p(bn4[3]), slope, _p(sums2), _p(ws), ws.numel(), _stream(), o)
    check(st, "ddmp_bn_bwd_reduce")
    return sums2


def bn_bwd_prepare(sums2, n_total, bn4, dgamma, dbeta, c10):
    """c10: float32 [2,C] rows = c1, c0 (written)."""
    C = dgamma.numel()
    st = _lib.lib().ddmp_bn_bwd_prepare_f32(_p(sums2), float(n_total), C, _p(bn4[0]), _p(bn4[2]), _p(bn4[3]),
                                            _p(dgamma), _p(dbeta), _p(c10[0]), _p(c10[1]), _stream())
    check(st, "ddmp_bn_bwd_prepare_f32")


def bn_bwd_apply(dz, y, bn4, c10, dy, dbias_sums, slope=SLOPE, n_rows=None):
    """dy = BatchNorm+LeakyReLU backward of (dz, y); dbias_sums (float64 [C]): its column sums, None: not computed."""
    dz, lddz = _mat(dz, "dz")
    y, ldy = _mat(y, "y", dz)
    dy, lddy = _mat(dy, "dy", dz)
    n = y.shape[0] if n_rows is None else n_rows
    C = y.shape[1]
    ws = _colws(n, C, y.device)
    with _timed("bn_bwd_apply", C, 3.0 * y.element_size() * n * C):
        st = _lib.lib().ddmp_bn_bwd_apply(_p(dz), lddz, _p(y), ldy, _p(dy), lddy, n, C, _dt(y), _p(bn4[0]), _p(bn4[1]),
                                          _p(c10[0]), _p(c10[1]), slope, _p(dbias_sums), _p(ws), ws.numel(),
                                          _stream())
    check(st, "ddmp_bn_bwd_apply")
    return dy


def colsum(x, sums=None, n_rows=None):
    x, ldx = _mat(x, "x")
    n = x.shape[0] if n_rows is None else n_rows
    C = x.shape[1]
    if sums is None:
        sums = torch.empty(C, dtype=torch.float64, device=x.device)
    ws = _colws(n, C, x.device)
    st = _lib.lib().ddmp_colsum_f32(_p(x), ldx, n, C, _p(sums), _p(ws), ws.numel(), _stream())
    check(st, "ddmp_colsum_f32")
    return sums


def f64_to_f32(src, dst):
    st = _lib.lib().ddmp_f64_to_f32(_p(src), _p(dst), src.numel(), _stream())
    check(st, "ddmp_f64_to_f32")
    return dst


def head_fwd(y, bn4, W1, b1, W2, b2, kind, x_pos, out, slope=SLOPE, n_rows=None):
    y, ldy = _mat(y, "y")
    n = y.shape[0] if n_rows is None else n_rows
    with _timed("head_fwd", kind, n * (32.0 * y.element_size() + 12 + (12 if kind == 0 else 0))):
        st = _lib.lib().ddmp_head_fwd(_p(y), ldy, n, _dt(y), _p(bn4[0]), _p(bn4[1]), slope, _p(W1), _p(b1), _p(W2),
                                      _p(b2), kind, _p(x_pos), _p(out), _stream())
    check(st, "ddmp_head_fwd")
    return out


def head_bwd(y, bn4, W1, b1, W2, b2, kind, dout, dz, dW1, db1, dW2, db2, slope=SLOPE, n_rows=None):
    y, ldy = _mat(y, "y")
    dz, lddz = _mat(dz, "dz", y)
    n = y.shape[0] if n_rows is None else n_rows
    L = _lib.lib()
    ws = Workspace.get(L.ddmp_head_bwd_workspace_bytes(n), y.device)
    with _timed("head_bwd", kind, n * (64.0 * y.element_size() + 12)):
        st = L.ddmp_head_bwd(_p(y), ldy, n, _dt(y), _p(bn4[0]), _p(bn4[1]), slope, _p(W1), _p(b1), _p(W2), _p(b2), kind,
                             _p(dout), _p(dz), lddz, _p(dW1), _p(db1), _p(dW2), _p(db2), _p(ws), ws.numel(),
                             _stream())
    check(st, "ddmp_head_bwd")


def grad_sumsq(g, out=None):
    L = _lib.lib()
    if out is None:
        out = torch.empty(1, dtype=torch.float64, device=g.device)
    ws = Workspace.get(L.ddmp_sumsq_workspace_bytes(), g.device)
    with _timed("optimizer", "sumsq", 4.0 * g.numel()):
        st = L.ddmp_grad_sumsq_f32(_p(g), g.numel(), _p(out), _p(ws), ws.numel(), _stream())
    check(st, "ddmp_grad_sumsq_f32")
    return out


def grad_clip_(g, sumsq, max_norm):
    st = _lib.lib().ddmp_grad_clip_f32(_p(g), g.numel(), _p(sumsq), float(max_norm), _stream())
    check(st, "ddmp_grad_clip_f32")


def adam_step_(p, g, m, v, lr, step, betas=(0.9, 0.999), eps=1e-8, clip_sumsq=None, max_norm=0.0):
    with _timed("optimizer", "adam", 28.0 * p.numel()):
        st = _lib.lib().ddmp_adam_step_f32(_p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), betas[0], betas[1], eps,
                                           int(step), _p(clip_sumsq), float(max_norm), _stream())
    check(st, "ddmp_adam_step_f32")


def adam_prepare(counter, lr, coef, betas=(0.9, 0.999)):
    """device-side: counter += 1; coef = (lr / (1 - b1^t), sqrt(1 - b2^t))  (int32 [1], float32 [2])."""
    check(_lib.lib().ddmp_adam_prepare(_p(counter), float(lr), betas[0], betas[1], _p(coef), _stream()), "ddmp_adam_prepare")


def adam_step_dev_(p, g, m, v, coef, betas=(0.9, 0.999), eps=1e-8, clip_sumsq=None, max_norm=0.0):
    with _timed("optimizer", "adam", 28.0 * p.numel()):
        st = _lib.lib().ddmp_adam_step_dev_f32(_p(p), _p(g), _p(m), _p(v), p.numel(), betas[0], betas[1], eps, _p(coef),
                                               _p(clip_sumsq), float(max_norm), _stream())
    check(st, "ddmp_adam_step_dev_f32")
