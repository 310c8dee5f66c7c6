"""One training iteration of Dual-DMP on the HIP path (``main.py:88-110`` == ``main4real.py:53-75``).

    zero_grad; pos = posnet(data); norm = normnet(data)
    loss = k1*pos_rec + k2*laplacian + k3*norm_rec + k4*[epoch>100]*bnf + k5*pos_norm
    loss.backward(); clip_grad_norm_(normnet, grad_crip); Adam.step() x2

:class:`FusedTrainer` runs exactly that sequence with everything device-resident and no autograd
graph: two engine forwards, :class:`loss.LossEngine` (values + analytic gradients + device-side
coefficients), two engine backwards, one global-norm reduction and two fused clip+Adam updates over the
flat arenas.  ``step()`` returns the loss as a 0-dim float64 device tensor; calling ``.item()`` on it is
the reference's only per-step sync (``main.py:113``).
"""
from __future__ import annotations

import torch

from . import ops
from .loss import LossEngine


def nonfinite_check(nets):
    """Raise OverflowError when a parameter or gradient arena holds a NaN / inf (see FusedTrainer.check_scales)."""
    for name, net in nets:
        for what, t in (("parameters", net.arena.data), ("gradients", getattr(net, "_grad_arena", None))):
            if t is not None and not bool(torch.isfinite(t).all()):
                raise OverflowError("non-finite %s in %s: NaN or inf in the activations / gradients of the last "
                                    "iterations" % (what, name))


class FusedTrainer:
    def __init__(self, posnet, normnet, dataset, n_mesh, pos_lr=0.01, norm_lr=0.01, k=(3.0, 4.0, 4.0, 4.0, 1.0),
                 grad_crip=0.8, bnfloop=1, betas=(0.9, 0.999), eps=1e-8, bnf_start_epoch=100, use_graph=False,
                 overlap=False):
        """``use_graph``: replay the whole iteration as one hipGraph (captured on the second call, re-captured
        when the BNF gate opens at ``bnf_start_epoch``).  Worth it for launch-bound small meshes (13k faces:
        ~560 launches per iteration); the Adam step count then lives on the device.

        ``overlap``: PosNet runs on a second HIP stream beside NormalNet (forward, then backward + its Adam
        update); the two nets only meet in the losses.  Same kernels, same results."""
        with ops.on_device(posnet.device):       # graphs and engines allocate on the CURRENT device (ddmp_graph_create)
            self._init(posnet, normnet, dataset, n_mesh, pos_lr, norm_lr, k, grad_crip, bnfloop, betas, eps,
                       bnf_start_epoch, use_graph, overlap)

    def _init(self, posnet, normnet, dataset, n_mesh, pos_lr, norm_lr, k, grad_crip, bnfloop, betas, eps,
              bnf_start_epoch, use_graph, overlap):
        self.posnet, self.normnet = posnet, normnet
        self.dataset = dataset
        dev = posnet.device
        self.device = dev
        self.pos_lr, self.norm_lr = pos_lr, norm_lr
        self.grad_crip = grad_crip
        self.betas, self.eps = betas, eps
        self.bnf_start_epoch = bnf_start_epoch
        self.loss_engine = LossEngine(n_mesh, dev, bnfloop=bnfloop, k=k)
        self.peng = posnet._get_engine(dataset)
        self.neng = normnet._get_engine(dataset)
        self.m = [torch.zeros_like(posnet.arena.data), torch.zeros_like(normnet.arena.data)]
        self.v = [torch.zeros_like(posnet.arena.data), torch.zeros_like(normnet.arena.data)]
        self.sumsq = torch.zeros(1, dtype=torch.float64, device=dev)
        self.epoch = 0          # drives the BNF gate (main.py:101)
        self.t = 0              # optimiser step count (Adam bias correction)
        self.lossbuf = None
        self.use_graph = use_graph
        self.overlap = overlap and self.peng.comm.world_size == 1
        self._side = torch.cuda.Stream(device=dev) if self.overlap else None
        self._graphs = {}       # gate -> torch.cuda.CUDAGraph
        self._warm = False
        self._t_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self._coef = [torch.zeros(2, dtype=torch.float32, device=dev) for _ in range(2)]

    def _fork(self):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._side.wait_event(ev)

    def _join(self):
        ev = torch.cuda.Event()
        ev.record(self._side)
        torch.cuda.current_stream().wait_event(ev)

    def _iteration(self, gate, dev_adam):
        pa, na = self.posnet.arena.data, self.normnet.arena.data
        pg, ng = self.posnet._grad_arena, self.normnet._grad_arena
        if dev_adam:
            ops.adam_prepare(self._t_dev, self.pos_lr, self._coef[0], self.betas)
            self._t_dev -= 1                                                # one counter, two learning rates
            ops.adam_prepare(self._t_dev, self.norm_lr, self._coef[1], self.betas)

        def pos_update():
            if dev_adam:
                ops.adam_step_dev_(pa, pg, self.m[0], self.v[0], self._coef[0], self.betas, self.eps)
            else:
                ops.adam_step_(pa, pg, self.m[0], self.v[0], self.pos_lr, self.t, self.betas, self.eps)

        if self.overlap:
            self._fork()
            with torch.cuda.stream(self._side):
                pos = self.peng.forward(pa, update_running=True)
            norm = self.neng.forward(na, update_running=True)
            self._join()
        else:
            pos = self.peng.forward(pa, update_running=True)
            norm = self.neng.forward(na, update_running=True)
        lossbuf, dpos, dnorm = self.loss_engine.forward_backward(pos, norm, gate)
        if self.overlap:
            self._fork()
            with torch.cuda.stream(self._side):
                self.peng.backward(pa, pg, dpos)
                pos_update()
            self.neng.backward(na, ng, dnorm)
        else:
            self.peng.backward(pa, pg, dpos)
            self.neng.backward(na, ng, dnorm)
            if self.peng.comm.world_size > 1:
                self.posnet._reduce_grads()
                self.normnet._reduce_grads()
            pos_update()
        ops.grad_sumsq(ng, out=self.sumsq)                                  # clip NormalNet only (main.py:108)
        if dev_adam:
            ops.adam_step_dev_(na, ng, self.m[1], self.v[1], self._coef[1], self.betas, self.eps,
                               clip_sumsq=self.sumsq, max_norm=self.grad_crip)
        else:
            ops.adam_step_(na, ng, self.m[1], self.v[1], self.norm_lr, self.t, self.betas, self.eps,
                           clip_sumsq=self.sumsq, max_norm=self.grad_crip)
        if self.overlap:
            self._join()
        return lossbuf, pos, norm

    @torch.no_grad()
    def check_scales(self) -> int:
        """f16 split GEMM modes: number of GEMM operands that outgrew their scale since the last call -- each was redone
        on the device with the measured scale before anything consumed it (GcnEngine.check_scales); raises
        OverflowError for non-finite operands.  Syncs.

        The scale slots record the operands' maxima with ``fmaxf``, which drops NaN: an inf operand raises through the
        slots, a NaN operand does not.  NaN activations or activation gradients reach every weight gradient of their layer
        and, through Adam, the parameters; so the parameter and gradient arenas are tested for finiteness here as well
        (two reductions over 0.75 M floats each, at the cadence of this call: every 10 epochs in the CLI)."""
        healed = self.peng.check_scales() + self.neng.check_scales()
        nonfinite_check((("PosNet", self.posnet), ("NormalNet", self.normnet)))
        return healed

    def step(self):
        with ops.on_device(self.device):
            return self._step()

    def _step(self):
        self.epoch += 1
        self.t += 1
        gate = 0.0 if self.epoch <= self.bnf_start_epoch else 1.0
        if not self.use_graph:
            self.lossbuf, self.pos, self.norm = self._iteration(gate, dev_adam=False)
            return self.lossbuf[5]
        self._t_dev.fill_(self.t - 1)                                       # keeps load_adam_state / eager steps in sync
        if not self._warm:                                                  # first call: eager (allocates workspaces)
            self._warm = True
            self.lossbuf, self.pos, self.norm = self._iteration(gate, dev_adam=True)
            return self.lossbuf[5]
        g = self._graphs.get(gate)
        if g is None:
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(g):
                self._cap = self._iteration(gate, dev_adam=True)
            self._graphs[gate] = (g, self._cap)
            g = self._graphs[gate]
        graph, (self.lossbuf, self.pos, self.norm) = g
        graph.replay()
        return self.lossbuf[5]

    # ------------------------------------------------------------------ state injection (parity harness)
    def load_adam_state(self, which: int, exp_avg: dict, exp_avg_sq: dict, t: int):
        """Set the Adam moments of net `which` (0 = PosNet, 1 = NormalNet) from reference-named tensors
        (as found in ``torch.optim.Adam.state``) and the shared step count."""
        net = (self.posnet, self.normnet)[which]
        self.m[which].zero_()
        self.v[which].zero_()
        for name, *_ in net.layout.entries:
            if name in exp_avg:
                net.layout.view(self.m[which], name).copy_(exp_avg[name].to(self.device))
                net.layout.view(self.v[which], name).copy_(exp_avg_sq[name].to(self.device))
        self.t = int(t)
