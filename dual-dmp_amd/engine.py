"""Fused GCN-stack engine: forward / backward of one PosNet / NormalNet trunk + head on the HIP kernels.

What the reference runs per net and step (util/networks.py:47-67, :108-130) as
12 x [GCNConv -> BatchNorm1d -> LeakyReLU] + 2 Linear becomes, per layer:

  aggregate-first (C_in <  C_out):  P = A_hat . f(Y_prev)        ddmp_spmm_f32   (prologue f fused)
                                    Y = P . W^T + b              ddmp_gemm_nt_f32
  transform-first (C_in >= C_out):  H = f(Y_prev) . W^T          ddmp_gemm_nt_f32 (prologue f fused)
                                    Y = A_hat . H + b            ddmp_spmm_f32
  statistics                        (sum, sumsq) of Y  -> scale/shift   ddmp_bn_stats_f32 + ddmp_bn_prepare_f32

f = LeakyReLU(BatchNorm(.)) of the previous layer is never materialised: only the conv outputs Y_l are
stored (and P_l where the weight gradient needs it).  A_hat.(X W^T) == (A_hat.X) W^T, so the aggregation
always runs on min(C_in, C_out) channels (sum C = 2000 instead of the reference order's 2496).
P_1 = A_hat.X_0 is constant (static graph, static input) and computed once.

Multi-device: rows [0, n_rows) are owned, rows [n_rows, n_cols) are the 1-hop halo.  The tensor about
to be gathered is the only thing exchanged (``comm.halo_exchange``), BatchNorm column sums and weight
gradients are all-reduced (``comm.all_reduce_sum``).  ``comm`` defaults to the single-device no-op.

Parameter arena (float32, one flat buffer per net; every tensor 16-byte aligned):
  layer l = 1..12:  W_l [C_out, C_in_padded] | b_l [C_out] | gamma_l [C_out] | beta_l [C_out]
  head:             W1 [16,32] | b1 [16] | W2 [3,16] | b2 [3]
"""
from __future__ import annotations

import math
from typing import List, Optional

import os

import torch

from . import ops
from .ops import SLOPE

def _unfused(name):
    """ops.unfused (DDMP_UNFUSE), tolerant of the CPU tests' stand-in of ops."""
    f = getattr(ops, "unfused", None)
    return bool(f and f(name))


POS_WIDTHS = [16, 32, 64, 128, 256, 256, 512, 512, 256, 256, 128, 64, 32, 16, 3]    # util/networks.py:13
NORM_WIDTHS = [7, 32, 64, 128, 256, 256, 512, 512, 256, 256, 128, 64, 32, 16, 3]    # util/networks.py:74


def _pad4(c):
    return (c + 3) // 4 * 4


class NoComm:
    """Single-device communicator: nothing to exchange."""
    world_size = 1
    rank = 0

    def halo_exchange(self, t, n_rows):
        return t

    def all_reduce_sum(self, t):
        return t

    # asynchronous forms (see GcnEngine.forward_steps): return a handle with .wait(), or None when there is nothing to wait for
    def start_halo(self, t, n_rows):
        return None

    def start_all_reduce(self, t):
        return None


class _Both:
    """Two started collectives as one handle."""

    def __init__(self, a, b):
        self.a, self.b = a, b

    def wait(self):
        for h in (self.a, self.b):
            if h is not None:
                h.wait()


class ArenaLayout:
    """Offsets (in floats) of every parameter tensor inside the flat arena."""

    def __init__(self, widths: List[int]):
        assert len(widths) == 15 and widths[12:] == [32, 16, 3]
        self.widths = list(widths)
        self.cin = [widths[0]] + widths[1:12]                 # true fan-in per conv
        self.cin_p = [_pad4(c) for c in self.cin]             # stored (padded) fan-in
        self.cout = widths[1:13]
        self.entries = []                                     # (name, offset, stored_shape, true_shape)
        off = 0

        def add(name, stored, true=None):
            nonlocal off
            n = 1
            for s in stored:
                n *= s
            self.entries.append((name, off, tuple(stored), tuple(true or stored)))
            off += (n + 3) // 4 * 4

        for l in range(12):
            i = l + 1
            add("conv%d.lin.weight" % i, (self.cout[l], self.cin_p[l]), (self.cout[l], self.cin[l]))
            add("conv%d.bias" % i, (self.cout[l],))
            add("bn%d.weight" % i, (self.cout[l],))
            add("bn%d.bias" % i, (self.cout[l],))
        add("linear1.weight", (16, 32))
        add("linear1.bias", (16,))
        add("linear2.weight", (3, 16))
        add("linear2.bias", (3,))
        self.total = off
        self.index = {e[0]: e for e in self.entries}
        self._views = {}

    def view(self, arena: torch.Tensor, name: str, true_shape=True):
        # the engines ask for the same ~60 views of the same two arenas every iteration: cached per arena storage
        # (the cached view keeps its storage alive, so an address cannot be handed to another arena meanwhile)
        if arena.requires_grad:
            return self._make_view(arena, name, true_shape)
        key = (arena.data_ptr(), arena.numel(), name, true_shape)
        v = self._views.get(key)
        if v is None:
            v = self._make_view(arena, name, true_shape)
            if len(self._views) > 1024:
                self._views.clear()
            self._views[key] = v
        return v

    def _make_view(self, arena: torch.Tensor, name: str, true_shape=True):
        _, off, stored, true = self.index[name]
        n = 1
        for s in stored:
            n *= s
        v = arena[off:off + n].view(stored)
        if true_shape and stored != true:
            v = v[:, :true[1]]
        return v

    def n_true_params(self):
        t = 0
        for _, _, _, true in self.entries:
            n = 1
            for s in true:
                n *= s
            t += n
        return t

    def init_(self, arena: torch.Tensor, generator: Optional[torch.Generator] = None):
        """PyG / torch default initialisation: Glorot-uniform conv weights (a = sqrt(6/(in+out))), zero
        conv bias, BN weight 1 / bias 0, nn.Linear default U(+-1/sqrt(fan_in)) for weight and bias."""
        cpu = torch.zeros(self.total, dtype=torch.float32)
        for name, off, stored, true in self.entries:
            n = 1
            for s in stored:
                n *= s
            v = cpu[off:off + n].view(stored)
            if name.endswith("lin.weight"):
                a = math.sqrt(6.0 / (true[0] + true[1]))
                v[:, :true[1]].copy_(torch.empty(true).uniform_(-a, a, generator=generator))
            elif name.startswith("bn") and name.endswith("weight"):
                v.fill_(1.0)
            elif name.startswith("linear"):
                fan_in = 32 if name.startswith("linear1") else 16
                b = 1.0 / math.sqrt(fan_in)
                v.copy_(torch.empty(stored).uniform_(-b, b, generator=generator))
        with torch.no_grad():
            arena.copy_(cpu.to(arena.device))


class GcnEngine:
    """Forward/backward of one net over a fixed graph.  Holds activations and scratch; parameters and
    gradients live in caller-owned flat arenas laid out by :class:`ArenaLayout`."""

    def __init__(self, graph: ops.Graph, widths: List[int], kind: int, x0: torch.Tensor,
                 x_pos: Optional[torch.Tensor] = None, comm=None, n_total: Optional[int] = None,
                 perm: Optional[torch.Tensor] = None, dtype: torch.dtype = torch.float32, split=None, overlap: bool = False):
        """``perm`` (int64 [n_rows], new -> old): the engine works on nodes relabelled for gather locality
        (``graph`` and ``x0``/``x_pos`` must already be in the NEW numbering); :meth:`forward` returns and
        :meth:`backward` accepts rows in the caller's ORIGINAL numbering.

        ``dtype``: feature dtype in HBM.  ``torch.bfloat16`` = the bf16-feature mode (BASELINE.json configs[1]): the
        static input, every conv output ``Y_l`` / aggregate ``P_l`` kept for backward and every activation gradient
        are bfloat16; parameters, BatchNorm coefficients, the [n,3] result and everything the optimizer touches
        stay float32 (arithmetic: float32 accumulation, float64 statistics, one bf16 MFMA product per step)."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("feature dtype must be torch.float32 or torch.bfloat16")
        self.dtype = dtype
        self.g = graph
        # ``split`` = (g_int, g_bnd, n_int) (multi-device, round 6): the graph's rows [0, n_int) -- the leading 64-row chunks that
        # reference no halo row -- and [n_int, n_rows) as graphs of their own (ops.Graph.from_csr_host(rows=...)).  Every
        # aggregation then runs as two launches: the halo exchange of the gathered tensor is STARTED (comm.start_halo_overlapped),
        # the interior rows are aggregated while it travels, the boundary rows behind it; fused column sums of the two halves are
        # added in float64.  The same two launches with the exchange waited for first (DDMP_DIST_SPLIT=noverlap) give the same bits.
        # ``overlap`` is the JOB's setting (the same on every rank: it fixes the order in which the collectives are issued), ``split``
        # this rank's halves -- None where a rank has no interior chunk or no halo: that rank waits and aggregates in one launch.
        self.overlap = bool(overlap)
        self.split = None
        if self.overlap and split is not None and 0 < split[2] < graph.n_rows:
            self.split = tuple(split)
            self.n_int = int(split[2])
        self._halo_ahead = None
        self.kind = kind
        self.layout = ArenaLayout(widths)
        self.comm = comm or NoComm()
        self.n_rows, self.n_cols = graph.n_rows, graph.n_cols
        self.n_total = float(n_total if n_total is not None else graph.n_rows)
        dev = x0.device
        self.device = dev
        L = self.layout
        # static input, padded to a multiple of 4 channels (z2 has 7 columns)
        assert x0.shape[0] >= self.n_cols and x0.shape[1] == L.cin[0]
        self.x0 = torch.zeros((self.n_cols, L.cin_p[0]), dtype=dtype, device=dev)
        self.x0[:, :L.cin[0]] = x0[:self.n_cols].to(dtype)
        self.x_pos = None if x_pos is None else x_pos[:self.n_rows].contiguous().to(torch.float32)
        # aggregate on the narrower side; equal widths aggregate first too: then dY feeds only GEMMs and, where the
        # fused kernels exist, is rebuilt on their operand loads instead of being written by bn_bwd_apply
        # (DDMP_UNFUSE=equal_width: equal widths transform first, the reference's own order -- A/B)
        eq_agg = not _unfused("equal_width")
        self.agg_first = [L.cin_p[l] < L.cout[l] or (eq_agg and L.cin_p[l] == L.cout[l]) for l in range(12)]
        supported = getattr(ops, "gemm_bnbwd_supported", None)
        def _bnbwd_ok(l):
            if not supported or l == 0 or not self.agg_first[l]:
                return False
            if dtype == torch.float32:
                return supported(L.cout[l], L.cin_p[l], self.n_rows)
            try:                                                 # bf16 features: row-register kernel (round 3)
                return supported(L.cout[l], L.cin_p[l], self.n_rows, dtype) and not _unfused("bf16_gemm")
            except TypeError:                                    # (a stand-in of ops without the dtype argument)
                return False
        self.fuse_bnbwd = [bool(_bnbwd_ok(l)) for l in range(12)]
        # layer 0 (no dgrad): its dY feeds the wgrad only
        tn_ok = getattr(ops, "gemm_tn_bnbwd_supported", None)
        self.fuse_bnbwd0 = bool(tn_ok and self.agg_first[0] and tn_ok(L.cout[0], L.cin_p[0], self.n_rows, dtype)
                                and not _unfused("bnbwd_l0"))
        # transform-first layers (l > 0 always: C_in > C_out) on ONE device: BatchNorm backward rebuilt on the SpMM's
        # gather.  Across devices the halo rows of Y_l would have to travel as well (they are not exchanged forward).
        # bf16 features: rebuilding dY on the gather reads two rows per CSR entry; measured (scripts/microbench.py spmm
        # --dtype bf16, 1M faces) it beats bn_bwd_apply + plain gather on the face graph (4 entries per row: 511 vs 554 us
        # at C = 256) and loses on the vertex graph (7 entries: 386 vs 321 us)
        gather_ok = getattr(ops, "spmm_bnbwd_supported", None)
        few_entries = dtype == torch.float32 or getattr(graph, "max_row_nnz", 99) <= 4
        self.fuse_gather_bwd = [bool(gather_ok) and isinstance(self.comm, NoComm) and not self.agg_first[l] and l > 0
                                and gather_ok(L.cout[l]) and few_entries and not _unfused("gather_bwd")
                                for l in range(12)]
        cmax = max(L.cout)
        nc = self.n_cols

        def buf(c):
            return torch.empty((nc, c), dtype=dtype, device=dev)

        self.Y = [buf(L.cout[l]) for l in range(12)]                   # conv outputs (pre-BN), saved
        self.P = [buf(L.cin_p[l]) if self.agg_first[l] else None for l in range(12)]
        self._flat = [torch.empty(nc * cmax, dtype=dtype, device=dev) for _ in range(6)]   # work buffers (rotation)
        # (zeros: row 2 = the last batch mean is the reference of the gather's statistics epilogue in the NEXT iteration)
        self.bn4 = [torch.zeros((4, L.cout[l]), dtype=torch.float32, device=dev) for l in range(12)]
        # transform-first layers, float32: BatchNorm statistics from the gather's epilogue (ddmp_spmm_stats_f32)
        st_ok = getattr(ops, "spmm_stats_supported", None)
        # On by default since round 4 (DDMP_UNFUSE=stats for A/B): -0.3 ... -0.4 ms per step at 1M faces, the same sign in every
        # interleaved A/B (round 3: 46.23 -> 46.07, 46.24 -> 45.93; round 4, scripts/layer_order_ab.sh: 46.38 / 46.23 -> 45.95 /
        # 45.93 ms).  The epilogue's work lands in the gather family (+0.6 ms there, -0.7 ms of bn_stats passes) and takes ~0.02
        # off that family's achieved-bandwidth fraction at the same algorithmic bytes -- the step is what is timed.
        # bf16 features: a tie (26.87 / 26.93 vs 26.98 / 26.77 ms), no fused form.
        self.fuse_spmm_stats = [bool(st_ok) and not self.agg_first[l] and st_ok(L.cout[l], dtype)
                                and not _unfused("stats") for l in range(12)]
        self.c10s = [torch.empty((2, L.cout[l]), dtype=torch.float32, device=dev) for l in range(12)]
        # weights split into their 16-bit planes once per iteration, all layers in two launches (float32 features)
        self._wplanes = None
        self._prep_weights = (dtype == torch.float32 and hasattr(ops, "gemm_prepare_weights")
                              and not _unfused("wprep"))
        self._tail_fused = (isinstance(self.comm, NoComm) and hasattr(ops, "BnFwd")
                            and not _unfused("tail"))
        # f16 split GEMM modes: one scale slot per layer and GEMM operand (0: the forward operand X, 1: the gradient
        # operand); the kernels record the operand maxima of this iteration, backward() rolls them into the scales
        # of the next one.  The first iteration measures (prime).
        self.scale_slots = torch.zeros((12, 2, 4), dtype=torch.float32, device=dev)
        self._slot = [[self.scale_slots[l, o] for o in range(2)] for l in range(12)]
        self._prime = True
        self._f16 = False
        self.sums = torch.empty(2 * cmax, dtype=torch.float64, device=dev)
        self.sums_b = torch.empty(2 * cmax, dtype=torch.float64, device=dev) if self.split else None   # boundary half's sums
        self.running = [torch.zeros((2, L.cout[l]), dtype=torch.float32, device=dev) for l in range(12)]
        for r in self.running:
            r[1].fill_(1.0)                                            # running_var starts at 1
        self.num_batches_tracked = torch.zeros(12, dtype=torch.int64, device=dev)
        self.out = torch.empty((self.n_rows, 3), dtype=torch.float32, device=dev)
        self._p1_ready = False
        self.perm = self.inv = None
        if perm is not None:
            self.perm = perm.to(dev)
            self.inv = torch.empty_like(self.perm)
            self.inv[self.perm] = torch.arange(self.n_rows, device=dev)
            self.out_orig = torch.empty_like(self.out)

    def __del__(self):
        # the library's prepared-planes registry must not outlive the plane buffers (an allocation that reuses their address
        # would be taken for prepared planes)
        try:
            planes = getattr(self, "_wplanes", None)
            if planes and hasattr(ops, "gemm_forget_planes"):
                for buf in planes.values():
                    ops.gemm_forget_planes(buf)
        except Exception:       # noqa: BLE001  (interpreter shutdown)
            pass

    def _work(self, i, c):
        return self._flat[i][: self.n_cols * c].view(self.n_cols, c)

    def _scales(self, l, a, b=None):
        """Keyword of a GEMM call: its operands' scale slots (f16 split modes only; an explicit per-call option, ABI 3)."""
        if self._f16:
            return {"scales": (self._slot[l][a], None if b is None else self._slot[l][b], self._prime)}
        return {}

    def check_scales(self) -> int:
        """f16 split modes.  An operand that outgrows its (one iteration old) scale is never computed with clamped:
        the GEMM call re-launches itself on the device and redoes the product with the measured maximum
        (csrc/gemm_f16s.inc, "self-healing"); the per-iteration roll counts those events per slot.  Returns the number
        of healed events since the last call (and resets the counters); raises OverflowError only when an operand was
        not finite (NaN / inf activations or gradients: nothing to heal).  Syncs."""
        cnt = self.scale_slots[:, :, 3]
        c = cnt.cpu()
        if bool((c < 0).any()):
            where = [(int(l), int(o)) for l, o in (c < 0).nonzero().tolist()]
            cnt.zero_()
            raise OverflowError("non-finite GEMM operand (layer, operand): %s -- NaN or inf in the activations / "
                                "gradients of the last iterations" % where)
        healed = int(c.sum().item())
        if healed:
            cnt.zero_()
        return healed

    # ------------------------------------------------------------------ forward
    @staticmethod
    def _drain(steps):
        for h in steps:
            if h is not None:
                h.wait()

    def forward(self, params: torch.Tensor, update_running: bool = True, use_running: bool = False) -> torch.Tensor:
        self._drain(self.forward_steps(params, update_running, use_running))
        return self.result

    def backward(self, params: torch.Tensor, grads: torch.Tensor, dout: torch.Tensor) -> torch.Tensor:
        """Overwrites ``grads`` (same layout as ``params``) with d loss / d params for the last forward."""
        self._drain(self.backward_steps(params, grads, dout))
        return grads

    def _bn4_from_running(self, l, gamma, beta):
        """eval() mode of nn.BatchNorm1d: normalise with the running statistics (a handful of C-length torch ops)."""
        rm, rv = self.running[l][0], self.running[l][1]
        rstd = torch.rsqrt(rv + ops.BN_EPS)
        b = self.bn4[l]
        b[0] = gamma * rstd
        b[1] = beta - rm * b[0]
        b[2] = rm
        b[3] = rstd

    def _prepare_weights(self, params: torch.Tensor):
        """Plane buffers per (layer, forward | dgrad) and the batched split of this iteration's weights into them."""
        if not self._prep_weights:
            return
        L = self.layout
        if self._wplanes is None:
            need = ops.gemm_rows_workspace_bytes
            self._wplanes = {(l, form): torch.empty(need(L.cin_p[l], L.cout[l]), dtype=torch.uint8, device=self.device)
                             for l in range(12) for form in (0, 1) if not (form == 1 and l == 0)}
            self._wscratch = torch.empty(8 * len(self._wplanes), dtype=torch.float32, device=self.device)
            self._wplan = (None, None)
        if self._wplan[0] != params.data_ptr():                  # (argument arrays built once per parameter arena)
            items = []
            for (l, form), buf in self._wplanes.items():
                W = L.view(params, "conv%d.lin.weight" % (l + 1), true_shape=False)
                items.append((W, form, form == 0 and not self.agg_first[l] and l > 0, buf))
            self._wplan = (params.data_ptr(), ops.WeightPlan(items, self.n_rows, self._wscratch))
        self._wplan[1].run()

    def _wp(self, l, form):
        """Keyword for the GEMM call of layer l (form 0 forward, 1 dgrad): its prepared planes, if any."""
        return {} if self._wplanes is None else {"wplanes": self._wplanes[(l, form)]}

    # ------------------------------------------------------------------ split aggregation (see __init__: ``split``)
    def _start_halo(self, t):
        """Split mode: the exchange of t's halo rows, started beside the interior rows' aggregation (a handle to wait on)."""
        comm, n = self.comm, self.n_rows
        start = getattr(comm, "start_halo_overlapped", None)
        if start is None or os.environ.get("DDMP_DIST_SPLIT") == "noverlap":
            h = comm.start_halo(t, n)                            # (A/B and bit-identity check: same launches, no overlap)
            if h is not None:
                h.wait()
            return None
        return start(t, n)

    def _halves(self):
        g_int, g_bnd, ni = self.split
        return ((g_int, 0, ni, self.sums), (g_bnd, ni, self.n_rows, self.sums_b))

    def _add_sums(self, width):
        if self.split:
            self.sums[:width] += self.sums_b[:width]

    def _agg(self, launch, h):
        """Overlap mode, as a generator: ``launch(graph, r0, r1, sums)`` aggregates rows [r0, r1) (fused column sums into
        ``sums``); ``h`` = the started exchange of the gathered tensor's halo rows.  Interior rows, wait, boundary rows -- or,
        on a rank without halves, wait and one launch."""
        if self.split:
            for k, (gh, r0, r1, sums) in enumerate(self._halves()):
                if k == 1:
                    yield h
                launch(gh, r0, r1, sums)
        else:
            yield h
            launch(self.g, 0, self.n_rows, self.sums)

    def forward_steps(self, params: torch.Tensor, update_running: bool = True, use_running: bool = False):
        """The forward pass as a generator that yields at every collective it STARTS (halo exchange, BatchNorm
        statistics all-reduce): the caller waits on the handle before resuming.  A multi-device trainer runs the
        two nets' generators alternately, so that one net's collective is in flight while the other net computes
        (dist.interleave); the plain forward() above waits immediately.  The result is left in ``self.out``."""
        L, g, n, comm = self.layout, self.g, self.n_rows, self.comm
        self.n_forward = getattr(self, "n_forward", 0) + 1        # (dist: cache key of the all-gathered outputs)
        self._f16 = hasattr(ops, "gemm_scales_roll") and self.dtype == torch.float32 and ops.get_gemm_mode() == 13
        self._prepare_weights(params)
        X, pro = self.x0, None
        halo_started = False
        # one device: the coefficients ride on the second stage of the reduction that produces the sums (finalize.h);
        # across devices the sums are all-reduced first
        tail = self._tail_fused and not use_running
        for l in range(12):
            i = l + 1
            W = L.view(params, "conv%d.lin.weight" % i, true_shape=False)
            b = L.view(params, "conv%d.bias" % i)
            Y = self.Y[l]
            # one device: this layer's BatchNorm coefficients come from the second stage of whichever call reduces Y
            bnk = {"bn": ops.BnFwd(self.n_total, L.view(params, "bn%d.weight" % i), L.view(params, "bn%d.bias" % i), self.bn4[l],
                                   running=(self.running[l][0], self.running[l][1]) if update_running else None)} if tail else {}
            if self.agg_first[l]:
                P = self.P[l]
                if self.overlap and l > 0:
                    h = self._halo_ahead if halo_started else self._start_halo(X)
                    self._halo_ahead = None
                    yield from self._agg(lambda gh, r0, r1, _: ops.spmm(gh, X, out=P[r0:r1], pro=pro), h)
                elif l > 0 or not self._p1_ready:
                    if l > 0 and not halo_started:
                        yield comm.start_halo(X, n)
                    ops.spmm(g, X, out=P[:n], pro=pro)
                    self._p1_ready = True
                if hasattr(ops, "gemm_nt_stats"):                # BatchNorm statistics from the GEMM epilogue
                    ops.gemm_nt_stats(P, W, self.sums, out=Y, bias=b, n_rows=n, **self._wp(l, 0), **self._scales(l, 0), **bnk)
                else:
                    ops.gemm_nt(P, W, out=Y, bias=b, n_rows=n)
                    ops.bn_stats(Y, sums=self.sums, n_rows=n, **bnk)
            else:
                H = self._work(0, L.cout[l])
                ops.gemm_nt(X, W, out=H, pro=pro, n_rows=n, **self._wp(l, 0), **self._scales(l, 0))
                if self.overlap:
                    if self.fuse_spmm_stats[l]:
                        yield from self._agg(lambda gh, r0, r1, sums: ops.spmm_stats(gh, H, Y[r0:r1], self.bn4[l][2], sums, bias=b),
                                             self._start_halo(H))
                        self._add_sums(2 * L.cout[l])
                    else:
                        yield from self._agg(lambda gh, r0, r1, _: ops.spmm(gh, H, out=Y[r0:r1], bias=b), self._start_halo(H))
                        ops.bn_stats(Y, sums=self.sums, n_rows=n, **bnk)
                else:
                    yield comm.start_halo(H, n)
                    if self.fuse_spmm_stats[l]:
                        ops.spmm_stats(g, H, Y[:n], self.bn4[l][2], self.sums, bias=b, **bnk)
                    else:
                        ops.spmm(g, H, out=Y[:n], bias=b)
                        ops.bn_stats(Y, sums=self.sums, n_rows=n, **bnk)
            # the halo rows of Y (raw, pre-BatchNorm: the consumer applies the prologue) do not depend on the statistics:
            # when the next layer gathers Y directly, its halo exchange travels together with the all-reduce
            halo_started = l < 11 and self.agg_first[l + 1]
            fused = None if self.overlap else getattr(comm, "halo_and_sums", None)
            if self.overlap:
                # the sums first (the next kernel's prologue needs them), the halo rows of Y behind them and beside the
                # interior rows' aggregation of the next layer
                yield comm.start_all_reduce(self.sums[: 2 * L.cout[l]])
                if halo_started:
                    self._halo_ahead = self._start_halo(Y)
            elif halo_started and fused is not None and fused(Y, n, self.sums[: 2 * L.cout[l]]):
                yield None                                       # one grouped launch carried both (native RCCL backend)
            else:
                h_stats = comm.start_all_reduce(self.sums[: 2 * L.cout[l]])
                h_halo = comm.start_halo(Y, n) if halo_started else None
                yield _Both(h_stats, h_halo)
            if use_running:
                self._bn4_from_running(l, L.view(params, "bn%d.weight" % i), L.view(params, "bn%d.bias" % i))
            elif not tail:
                ops.bn_prepare(self.sums, self.n_total, L.view(params, "bn%d.weight" % i), L.view(params, "bn%d.bias" % i),
                               self.bn4[l], running=(self.running[l][0], self.running[l][1]) if update_running else None)
            X, pro = Y, (self.bn4[l][0], self.bn4[l][1])
        if update_running:
            self.num_batches_tracked += 1
        ops.head_fwd(self.Y[11], self.bn4[11], L.view(params, "linear1.weight"), L.view(params, "linear1.bias"),
                     L.view(params, "linear2.weight"), L.view(params, "linear2.bias"), self.kind, self.x_pos,
                     self.out, n_rows=n)
        self.result = self.out
        if self.inv is not None:
            torch.index_select(self.out, 0, self.inv, out=self.out_orig)
            self.result = self.out_orig

    # ------------------------------------------------------------------ backward
    def backward_steps(self, params: torch.Tensor, grads: torch.Tensor, dout: torch.Tensor):
        """Backward pass as a generator (see forward_steps).  (A further stream for the weight gradients -- GEMMs nothing in
        this pass waits for -- was an option until round 5: -1.6 ... -2.5 ms per step with the round-4 wgrad kernel at 1M faces,
        nothing at 13k faces, never capturable beside the two nets' streams; removed in round 6: profiles/r04_stream_overlap_ab.txt.)"""
        L, g, n, comm = self.layout, self.g, self.n_rows, self.comm
        free = list(range(len(self._flat)))                     # work buffers, FIFO

        def take(c):
            k = free.pop(0)
            return k, self._work(k, c)

        def release(k):
            free.append(k)

        def wgrad(l, fn, *bufs):
            """fn() launches the weight-gradient GEMM of layer l; bufs = work buffers it reads."""
            fn()

        kz, dZ = take(32)
        if self.perm is not None:
            dout = dout.index_select(0, self.perm)
        tail = self._tail_fused

        def arm(l):
            """Keyword of the call that reduces for layer l: its BatchNorm-backward coefficients come from that call's second
            stage (one device; an explicit per-call option, ABI 3)."""
            if tail:
                return {"bn": ops.BnBwd(self.n_total, self.bn4[l], L.view(grads, "bn%d.weight" % (l + 1)),
                                        L.view(grads, "bn%d.bias" % (l + 1)), self.c10s[l])}
            return {}

        # the conv-bias gradients: zero after BatchNorm in exact arithmetic (its backward output has zero column mean), where
        # the reference's autograd leaves float32 summation noise.  Written as 0 on every route, in one launch
        torch._foreach_zero_([L.view(grads, "conv%d.bias" % (l + 1)) for l in range(12)])
        ops.head_bwd(self.Y[11], self.bn4[11], L.view(params, "linear1.weight"), L.view(params, "linear1.bias"),
                     L.view(params, "linear2.weight"), L.view(params, "linear2.bias"), self.kind, dout.contiguous(), dZ,
                     L.view(grads, "linear1.weight"), L.view(grads, "linear1.bias"),
                     L.view(grads, "linear2.weight"), L.view(grads, "linear2.bias"), n_rows=n)
        have_sums = False
        # the backward column reductions from the SpMM epilogue.  bf16 features: the epilogue (one more row stream + 16 cross-lane
        # sums per 128-byte slab) used to cost as much as the separate pass it replaces (C = 512: 939 us fused vs 457 + 390 us);
        # with one partial record per chunk (round 4) the step gains 0.27 ms (26.21 / 26.22 -> 25.97 / 25.92 ms, interleaved):
        # on by default, DDMP_UNFUSE=bf16_spmm_red for A/B
        fuse_red = hasattr(ops, "spmm_bnred") and (self.dtype == torch.float32
                                                   or not _unfused("bf16_spmm_red"))

        def spmm_to_dz(src, dst, l):
            """dZ of layer l-1 = A^T src; with its BatchNorm-backward column reductions where the kernel can."""
            if fuse_red and l > 0:
                ops.spmm_bnred(g, src, dst[:n], self.Y[l - 1], self.bn4[l - 1], self.sums, **arm(l - 1))
                return True
            ops.spmm(g, src, out=dst[:n])
            return False

        def spmm_to_dz_split(src, dst, l):
            """The same in overlap mode, as a generator: exchange started, interior rows, wait, boundary rows."""
            red = fuse_red and l > 0
            if red:
                yield from self._agg(lambda gh, r0, r1, sums: ops.spmm_bnred(gh, src, dst[r0:r1], self.Y[l - 1][r0:r1],
                                                                            self.bn4[l - 1], sums), self._start_halo(src))
                self._add_sums(2 * L.cout[l - 1])
            else:
                yield from self._agg(lambda gh, r0, r1, _: ops.spmm(gh, src, out=dst[r0:r1]), self._start_halo(src))
            return red

        # transform-first dgrads with the next BatchNorm-backward reductions in their epilogue (row-register kernel, float32).
        # Measured at 1M faces (interleaved A/B, 10 steps each): 47.49 / 47.35 ms with, 47.75 / 47.78 ms without.  (A first
        # version lost 0.9 ms: the 32 per-column coefficients of the epilogue were hoisted out of the tile loop as invariants and
        # SPILLED the main loop -- scratch reloads inside the counted-vmcnt pipeline; they now live in LDS and
        # scripts/check_rr_asm.py audits that form too.)  DDMP_UNFUSE=dgrad_red for A/B.
        fuse_dgrad_red = (getattr(ops, "gemm_nn_bnred_supported", None) is not None
                          and not _unfused("dgrad_red"))

        def dgrad_to_dz(dH, W, dZ, l):
            """dZ of layer l-1 = dH . W (transform-first layer l > 0); with that layer's BatchNorm-backward column reductions
            from the GEMM epilogue where the kernel exists (across devices too: the epilogue sums this rank's owned rows,
            which is what n selects, and the sums are all-reduced as after the separate pass)."""
            if fuse_dgrad_red and ops.gemm_nn_bnred_supported(L.cout[l], L.cin_p[l], n, self.dtype):
                ops.gemm_nn_bnred(dH, W, self.Y[l - 1], self.bn4[l - 1], self.sums, out=dZ, n_rows=n, **self._wp(l, 1),
                                  **self._scales(l, 1), **arm(l - 1))
                return True
            ops.gemm_nn(dH, W, out=dZ, n_rows=n, **self._wp(l, 1), **self._scales(l, 1))
            return False

        for l in range(11, -1, -1):
            i = l + 1
            co, ci = L.cout[l], L.cin_p[l]
            W = L.view(params, "conv%d.lin.weight" % i, true_shape=False)
            dW = L.view(grads, "conv%d.lin.weight" % i, true_shape=False)
            Y, bn4, c10 = self.Y[l], self.bn4[l], self.c10s[l]
            if not have_sums:                                    # else: produced by the SpMM that wrote dZ
                ops.bn_bwd_reduce(dZ, Y, bn4, sums2=self.sums, n_rows=n, **arm(l))
            have_sums = False
            yield comm.start_all_reduce(self.sums[: 2 * co])
            if not tail:
                ops.bn_bwd_prepare(self.sums, self.n_total, bn4, L.view(grads, "bn%d.weight" % i),
                                   L.view(grads, "bn%d.bias" % i), c10)
            if self.fuse_bnbwd[l]:
                # dY is never written: the two GEMMs that consume it rebuild it from (dZ, Y) on their operand loads
                kp, dP = take(ci)
                ops.gemm_nn_bnbwd(dZ, Y, W, bn4, c10, out=dP, n_rows=n, **self._wp(l, 1), **self._scales(l, 1))
                # after the dgrad GEMM: two panel GEMMs cannot share a CU (LDS), the wgrad's partners are the SpMM and
                # the BatchNorm passes that follow
                wgrad(l, lambda: ops.gemm_tn_bnbwd(dZ, Y, self.P[l], bn4, c10, out=dW, n_rows=n, **self._scales(l, 1, 0)), kz)
                release(kz)
                if self.overlap:
                    kz, dZ = take(ci)
                    have_sums = yield from spmm_to_dz_split(dP, dZ, l)
                else:
                    yield comm.start_halo(dP, n)
                    kz, dZ = take(ci)
                    have_sums = spmm_to_dz(dP, dZ, l)
                release(kp)
                continue
            if self.fuse_gather_bwd[l]:
                # transform-first layer on one device: dY is rebuilt by the SpMM on its gather of (dZ, Y) rows
                kh, dH = take(co)
                ops.spmm_bnbwd(g, dZ, Y, bn4, c10, dH[:n])
                release(kz)
                Xp, pro = self.Y[l - 1], (self.bn4[l - 1][0], self.bn4[l - 1][1])
                kz, dZ = take(ci)
                have_sums = dgrad_to_dz(dH, W, dZ, l)
                wgrad(l, lambda: ops.gemm_tn(dH, Xp, out=dW, pro=pro, n_rows=n, **self._scales(l, 1, 0)), kh)
                release(kh)
                continue
            if l == 0 and self.fuse_bnbwd0:
                wgrad(0, lambda: ops.gemm_tn_bnbwd(dZ, Y, self.P[0], bn4, c10, out=dW, n_rows=n, **self._scales(0, 1, 0)), kz)
                release(kz)
                continue
            ky, dY = take(co)
            ops.bn_bwd_apply(dZ, Y, bn4, c10, dY, None, n_rows=n)
            release(kz)                                          # dZ is dead once dY exists
            if l > 0:
                Xp, pro = self.Y[l - 1], (self.bn4[l - 1][0], self.bn4[l - 1][1])
            else:
                Xp, pro = self.x0, None
            if self.agg_first[l]:
                if l > 0:
                    kp, dP = take(ci)
                    ops.gemm_nn(dY, W, out=dP, n_rows=n, **self._wp(l, 1), **self._scales(l, 1))
                wgrad(l, lambda: ops.gemm_tn(dY, self.P[l], out=dW, n_rows=n, **self._scales(l, 1, 0)), ky)
                if l > 0:
                    release(ky)
                    if self.overlap:
                        kz, dZ = take(ci)
                        have_sums = yield from spmm_to_dz_split(dP, dZ, l)
                    else:
                        yield comm.start_halo(dP, n)
                        kz, dZ = take(ci)
                        have_sums = spmm_to_dz(dP, dZ, l)
                    release(kp)
                else:
                    release(ky)
            else:
                if self.overlap:
                    h = self._start_halo(dY)
                    kh, dH = take(co)
                    yield from self._agg(lambda gh, r0, r1, _: ops.spmm(gh, dY, out=dH[r0:r1]), h)
                else:
                    yield comm.start_halo(dY, n)
                    kh, dH = take(co)
                    ops.spmm(g, dY, out=dH[:n])
                release(ky)
                if l > 0:
                    kz, dZ = take(ci)
                    have_sums = dgrad_to_dz(dH, W, dZ, l)
                wgrad(l, lambda: ops.gemm_tn(dH, Xp, out=dW, pro=pro, n_rows=n, **self._scales(l, 1, 0)), kh)
                release(kh)
        if self._f16:
            ops.gemm_scales_roll(self.scale_slots)
            self._prime = False
