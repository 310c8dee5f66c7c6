"""Multi-GPU execution of the training step: one process per GPU, torch.distributed over RCCL/xGMI.

The reference is single-device; this is the build's own row (e).  What shards and how:

* The two GCN stacks (99 % of the step) are ROW-partitioned: the face graph over P contiguous chunks of a
  Morton order of the face centroids, the vertex graph by "owner of the vertex's first incident face".
  Rank r keeps its owned rows plus a 1-hop halo (rows [n_rows, n_cols)) and a local CSR whose columns
  index [owned | halo]; ``dinv`` comes from the GLOBAL degrees.
* Per layer and direction the ONLY feature exchange is the tensor about to be gathered (width
  min(C_in, C_out)): ``all_to_all_single`` of the boundary rows straight into the halo rows.  Because
  A_hat is symmetric the backward aggregation is the same gather, so no reverse scatter-add exists.
* BatchNorm statistics: float64 column sums all-reduced (2*C values) before ``bn_prepare`` /
  ``bn_bwd_prepare``; weight gradients: one all-reduce of the flat gradient arena per net and step
  (BN weight/bias gradients are already global and are excluded from the sum).
* The five losses (<2 % of the step) are SHARDED by recomputation, not by exchange: a rank sums the loss terms of the
  rows it owns and keeps their gradient rows, and evaluates everything those read on a ghost closure
  (:class:`LossShard`: ``2*loop+1`` face rings for the bilateral filter, two vertex rings for the Laplacian, the faces
  around owned vertices, and the mesh's LAST face -- what f2f == -1 gathers).  Per step: two ghost exchanges (vertex
  positions, face normals: the same grouped send/recv as a layer's halo) and two all-reduces of float64 partial sums
  (sigma_c, a mean over ALL faces, before the filter passes; S1..S5 before the loss scalars).  ``losses="replicated"``
  keeps the round-1 form (all-gather pos | norm, every rank runs the whole-mesh :class:`loss.LossEngine`).
* Parameters are replicated; identical reduced gradients + identical Adam state keep them bit-identical.

Communicators: :class:`TorchDistComm` (nccl = RCCL on GPUs, gloo in the CPU tests) and :class:`ThreadComm`
(P logical ranks as threads in ONE process / on ONE device, used to check the partitioned HIP path against
the unpartitioned one on a single GPU).
"""
from __future__ import annotations

import threading
from typing import List, Optional

import numpy as np
import torch

from . import ops


# ------------------------------------------------------------------------------------ partitioning
def morton_order(points: np.ndarray) -> np.ndarray:
    """argsort of 3-D Morton codes (21 bits per axis) of `points` [n,3]."""
    p = np.asarray(points, dtype=np.float64)
    lo, hi = p.min(0), p.max(0)
    q = ((p - lo) / np.maximum(hi - lo, 1e-30) * (2 ** 21 - 1)).astype(np.uint64)

    def spread(x):
        x &= np.uint64(0x1FFFFF)
        x = (x | (x << np.uint64(32))) & np.uint64(0x1F00000000FFFF)
        x = (x | (x << np.uint64(16))) & np.uint64(0x1F0000FF0000FF)
        x = (x | (x << np.uint64(8))) & np.uint64(0x100F00F00F00F00F)
        x = (x | (x << np.uint64(4))) & np.uint64(0x10C30C30C30C30C3)
        x = (x | (x << np.uint64(2))) & np.uint64(0x1249249249249249)
        return x

    code = spread(q[:, 0].copy()) | (spread(q[:, 1].copy()) << np.uint64(1)) | (spread(q[:, 2].copy()) << np.uint64(2))
    return np.argsort(code, kind="stable")


def rcb_order(points: np.ndarray, leaf: int = 64) -> np.ndarray:
    """Recursive coordinate bisection of `points` [n,3] into leaves of `leaf` consecutive ids (new id -> old id): every
    leaf -- and every aligned group of 2, 4, ... leaves -- is a compact patch of the surface.  Host code of the library
    (``ddmp_rcb_order_host``, csrc/graph.hip); called through ctypes directly so that it also serves the CPU tests that
    replace ``ops`` by a stand-in."""
    import ctypes  # noqa: F401
    from . import _lib
    p = np.ascontiguousarray(points, dtype=np.float64)
    assert p.ndim == 2 and p.shape[1] == 3
    order = np.zeros(len(p), np.int32)
    _lib.check(_lib.lib().ddmp_rcb_order_host(len(p), p.ctypes.data, int(leaf), order.ctypes.data), "ddmp_rcb_order_host")
    return order.astype(np.int64)


def face_owner_morton(fc: np.ndarray, P: int) -> np.ndarray:
    """P contiguous chunks of the Morton order of the face centroids, balanced on faces."""
    order = morton_order(fc)
    owner = np.empty(len(fc), dtype=np.int32)
    bounds = np.linspace(0, len(fc), P + 1).astype(np.int64)
    for r in range(P):
        owner[order[bounds[r]:bounds[r + 1]]] = r
    return owner


def face_owner_rcb(fc: np.ndarray, P: int) -> np.ndarray:
    """P contiguous chunks of the RCB order of the face centroids (:func:`rcb_order`), balanced on faces: compact parts (a
    chunk of the coordinate-bisection order is a union of few spatial boxes; a chunk of the Morton curve can be a ragged
    staircase), so fewer halo rows per owned row.  Default partition since round 5 (METIS is not available)."""
    order = rcb_order(fc, 64)
    owner = np.empty(len(fc), dtype=np.int32)
    bounds = np.linspace(0, len(fc), P + 1).astype(np.int64)
    for r in range(P):
        owner[order[bounds[r]:bounds[r + 1]]] = r
    return owner


def vertex_owner_from_faces(faces: np.ndarray, face_owner: np.ndarray, n_verts: int) -> np.ndarray:
    """A vertex belongs to the rank owning its lowest-numbered incident face."""
    flat = faces.reshape(-1)
    fid = np.repeat(np.arange(len(faces), dtype=np.int64), 3)
    first = np.full(n_verts, np.iinfo(np.int64).max, dtype=np.int64)
    np.minimum.at(first, flat, fid)
    owner = np.zeros(n_verts, dtype=np.int32)
    has = first < np.iinfo(np.int64).max
    owner[has] = face_owner[first[has]]
    return owner


CHUNK = 64                                                       # rows per gather chunk (csrc: ddmp::kChunkRows)


def interior_first_keys(rowptr: np.ndarray, col: np.ndarray, owner: np.ndarray, key: np.ndarray, P: int) -> np.ndarray:
    """Per owner, a new local order of its rows: the 64-row chunks of the order `key` are kept as they are (a chunk is a compact
    patch of the surface: gather locality), but the chunks in which no row has a neighbour on another rank come FIRST -- the
    rows a rank can aggregate before its halo rows have arrived (the halo exchange is overlapped with them).  A
    ragged last chunk stays last.  Returns the new key (only its order within an owner matters)."""
    n = len(rowptr) - 1
    owner = owner.astype(np.int64)
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rowptr).astype(np.int64))
    touches = np.zeros(n, dtype=bool)
    touches[rows[owner[rows] != owner[col.astype(np.int64)]]] = True
    order = np.lexsort((key, owner))
    off = np.concatenate([[0], np.cumsum(np.bincount(owner, minlength=P))])
    new = np.empty(n, dtype=np.int64)
    for r in range(P):
        ids = order[off[r]:off[r + 1]]
        m = len(ids)
        if m == 0:
            continue
        ch = np.arange(m) // CHUNK
        nch = int(ch[-1]) + 1
        bnd = np.zeros(nch, dtype=bool)
        np.logical_or.at(bnd, ch, touches[ids])
        if m % CHUNK:
            bnd[-1] = True
        pos = np.empty(nch, dtype=np.int64)
        pos[np.argsort(bnd, kind="stable")] = np.arange(nch)
        new[ids] = pos[ch] * CHUNK + np.arange(m) % CHUNK
    return new


class HaloPlan:
    """Everything rank `rank` needs for one graph: local CSR over [owned | halo], global ids, and the
    all-to-all schedule.  Built identically (deterministically) on every rank from the global CSR."""

    def __init__(self, rowptr: np.ndarray, col: np.ndarray, dinv: np.ndarray, owner: np.ndarray, rank: int, P: int,
                 order_key: Optional[np.ndarray] = None):
        """``order_key`` [n]: owned rows are stored in increasing key (e.g. the Morton rank of the node) so that
        consecutive local rows are spatial neighbours; the exchange schedule is keyed on GLOBAL ids and is the
        same on every rank whatever the local order."""
        n = len(rowptr) - 1
        owner = owner.astype(np.int64)
        rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rowptr).astype(np.int64))
        colg = col.astype(np.int64)
        cut = owner[rows] != owner[colg]
        # (dest rank, needed node) pairs, unique; dest d receives node j from owner[j]
        key = owner[rows[cut]] * np.int64(n) + colg[cut]
        key = np.unique(key)
        dest, node = key // n, key % n
        src = owner[node]
        self.rank, self.P = rank, P
        self.owned = np.flatnonzero(owner == rank).astype(np.int64)            # ascending global ids ...
        if order_key is not None:                                               # ... or along the locality key
            self.owned = self.owned[np.argsort(order_key[self.owned], kind="stable")]
        mine = dest == rank
        o = np.lexsort((node[mine], src[mine]))                                 # grouped by source rank, then id
        self.halo = node[mine][o]
        self.recv_counts = np.bincount(src[mine], minlength=P).astype(np.int64).tolist()
        g2l = np.full(n, -1, dtype=np.int64)
        g2l[self.owned] = np.arange(len(self.owned))
        g2l[self.halo] = len(self.owned) + np.arange(len(self.halo))
        out = src == rank
        o = np.lexsort((node[out], dest[out]))                                  # grouped by dest rank, then id
        self.send_idx = g2l[node[out][o]]                                       # local owned indices
        self.send_counts = np.bincount(dest[out], minlength=P).astype(np.int64).tolist()
        self.n_rows, self.n_cols = len(self.owned), len(self.owned) + len(self.halo)
        # local CSR over owned rows
        cnt = np.diff(rowptr).astype(np.int64)[self.owned]
        self.rowptr = np.zeros(self.n_rows + 1, dtype=np.int32)
        np.cumsum(cnt, out=self.rowptr[1:])
        starts = rowptr[self.owned].astype(np.int64)
        idx = np.repeat(starts - self.rowptr[:-1].astype(np.int64), cnt) + np.arange(int(cnt.sum()), dtype=np.int64)
        self.col = g2l[colg[idx]].astype(np.int32)
        assert (self.col >= 0).all()
        self.local_ids = np.concatenate([self.owned, self.halo])
        self.dinv = dinv[self.local_ids].astype(np.float32)
        self.n_global = n
        # rows [0, n_int): the leading 64-row chunks none of whose rows references a halo row -- what can be aggregated while the
        # halo rows are still travelling (interior_first_keys puts every such chunk in front)
        touches = np.zeros(n, dtype=bool)
        touches[rows[cut]] = True
        nch = (self.n_rows + CHUNK - 1) // CHUNK
        flags = np.zeros(max(nch, 1), dtype=bool)
        if self.n_rows:
            np.logical_or.at(flags, np.arange(self.n_rows) // CHUNK, touches[self.owned])
        lead = int(np.argmax(flags)) if flags.any() else nch
        self.n_int = int(min(lead * CHUNK, self.n_rows))


# ------------------------------------------------------------------------------------ communicators
class GraphComm:
    """Per-graph view of a backend: what :class:`engine.GcnEngine` calls."""

    def __init__(self, backend, plan: HaloPlan, device):
        self.backend, self.plan = backend, plan
        self.world_size, self.rank = backend.world_size, backend.rank
        self.send_idx = torch.from_numpy(plan.send_idx).to(device)

    def halo_exchange(self, t: torch.Tensor, n_rows: int):
        p = self.plan
        assert t.shape[0] >= p.n_cols and n_rows == p.n_rows and t.is_contiguous()
        native = getattr(self.backend, "halo_exchange_native", None)
        if native is not None:                                  # pack + grouped send/recv from C (csrc/comm.hip)
            return native(p, t)
        send = self._pack(t, n_rows)
        recv = t[n_rows:p.n_cols]
        self.backend.all_to_all(recv, send, p.recv_counts, p.send_counts)
        return t

    def _pack(self, t, n_rows):
        """Boundary rows in plan order: the library's gather kernel on the GPU (ddmp_rows_gather), torch on the CPU stub."""
        if t.is_cuda and (t.shape[1] * t.element_size()) % 16 == 0 and self.send_idx.numel() > 0:
            return ops.rows_gather(t[:n_rows], self.send_idx)
        return t[:n_rows].index_select(0, self.send_idx)

    def all_reduce_sum(self, t: torch.Tensor):
        return self.backend.all_reduce_sum(t)

    # ---- asynchronous forms: start the collective, return a handle (or None) to wait on before the data is used
    def start_halo(self, t: torch.Tensor, n_rows: int):
        p = self.plan
        assert t.shape[0] >= p.n_cols and n_rows == p.n_rows and t.is_contiguous()
        native = getattr(self.backend, "halo_exchange_native", None)
        if native is not None:                                  # stream-ordered: nothing to wait for on the host
            native(p, t)
            return None
        send = self._pack(t, n_rows)
        recv = t[n_rows:p.n_cols]
        start = getattr(self.backend, "all_to_all_start", None)
        if start is None:
            self.backend.all_to_all(recv, send, p.recv_counts, p.send_counts)
            return None
        return start(recv, send, p.recv_counts, p.send_counts)

    def start_halo_overlapped(self, t: torch.Tensor, n_rows: int):
        """Start the exchange of `t`'s halo rows so that it runs BESIDE what the caller enqueues next (the aggregation of the
        interior rows, which reference no halo row: GcnEngine split mode); the returned handle's wait() orders the caller's
        stream behind it.  Native RCCL backend with an exchange stream (NativeComm.use_xs): pack + grouped send/recv go to the
        communicator's exchange stream behind an event of the current stream (the producer of `t`); other backends: start_halo
        (asynchronous where the backend can)."""
        native = getattr(self.backend, "halo_exchange_native", None)
        if native is None or not getattr(self.backend, "use_xs", False) or torch.cuda.is_current_stream_capturing():
            return self.start_halo(t, n_rows)
        p = self.plan
        assert t.shape[0] >= p.n_cols and n_rows == p.n_rows and t.is_contiguous()
        return native(p, t, defer=True)

    def halo_and_sums(self, t: torch.Tensor, n_rows: int, sums: torch.Tensor) -> bool:
        """The halo rows of `t` and the all-reduce of the BatchNorm column sums in ONE grouped RCCL launch (native
        backend only; False = not available, the caller issues the two collectives separately)."""
        native = getattr(self.backend, "halo_exchange_native", None)
        if native is None:
            return False
        native(self.plan, t, sums)
        return True

    def start_all_reduce(self, t: torch.Tensor):
        start = getattr(self.backend, "all_reduce_start", None)
        if start is None:
            self.backend.all_reduce_sum(t)
            return None
        return start(t)


class _StreamWait:
    """An exchange enqueued on another stream: wait() orders the CURRENT stream behind it (no host wait)."""

    def __init__(self, done):
        self.done = done

    def wait(self):
        torch.cuda.current_stream().wait_event(self.done)


class _Pending:
    """A started torch.distributed collective; keeps its buffers alive until it has been waited for."""

    def __init__(self, work, *keep):
        self.work, self.keep = work, keep

    def wait(self):
        self.work.wait()
        self.keep = None


def interleave(a, b):
    """Run two step generators (GcnEngine.forward_steps / backward_steps) alternately: each runs until it STARTS a
    collective, then the other one gets the device while that collective is in flight.  Every rank executes the same
    sequence, so the collectives are issued in the same order everywhere."""
    ha = hb = None
    done_a = done_b = False
    while not (done_a and done_b):
        if not done_a:
            if ha is not None:
                ha.wait()
            try:
                ha = next(a)
            except StopIteration:
                done_a, ha = True, None
        if not done_b:
            if hb is not None:
                hb.wait()
            try:
                hb = next(b)
            except StopIteration:
                done_b, hb = True, None


class TorchDistComm:
    """torch.distributed backend: "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world_size, self.rank = dist.get_world_size(group), dist.get_rank(group)

    def all_to_all(self, recv, send, recv_counts, send_counts):
        self.dist.all_to_all_single(recv, send, output_split_sizes=recv_counts, input_split_sizes=send_counts,
                                    group=self.group)

    def all_reduce_sum(self, t):
        self.dist.all_reduce(t, group=self.group)
        return t

    def all_to_all_start(self, recv, send, recv_counts, send_counts):
        work = self.dist.all_to_all_single(recv, send, output_split_sizes=recv_counts, input_split_sizes=send_counts,
                                           group=self.group, async_op=True)
        return _Pending(work, recv, send)

    def all_reduce_start(self, t):
        return _Pending(self.dist.all_reduce(t, group=self.group, async_op=True), t)

    def all_gather_rows(self, out, local):
        """out [P * m, c] <- every rank's `local` [m, c] (equal shapes), rank-major."""
        self.dist.all_gather_into_tensor(out, local.contiguous(), group=self.group)
        return out

    def barrier(self):
        self.dist.barrier(group=self.group)


class NativeComm:
    """RCCL through the library's OWN communicator and halo plans (csrc/comm.hip, include/ddmp_hip.h "Communicator +
    halo plan + exchange"): the pack kernel, the grouped ncclSend/ncclRecv that lands halo rows in place and the
    BatchNorm-sum all-reduce are enqueued from C on the caller's stream -- no Python collective objects, capturable.
    torch.distributed is only the bootstrap channel (the 128-byte unique id) and the barrier.  Default on the GPU since
    round 3 (DDMP_DIST_NATIVE=0 opts out); no multi-GPU box exists in the build loop: hardware runs at world size 1 only."""

    def __init__(self, device):
        import ctypes
        import torch.distributed as dist
        from . import _lib
        self.dist, self.L, self.ct = dist, _lib.lib(), ctypes
        self.world_size, self.rank = dist.get_world_size(), dist.get_rank()
        self.device = torch.device(device)
        buf = ctypes.create_string_buffer(128)
        if self.rank == 0:
            _lib.check(self.L.ddmp_comm_unique_id(buf), "ddmp_comm_unique_id")
        box = [bytes(buf.raw)]
        dist.broadcast_object_list(box, src=0)
        h = ctypes.c_void_p()
        with ops.on_device(self.device):
            _lib.check(self.L.ddmp_comm_create(self.rank, self.world_size, ctypes.c_char_p(box[0]), ctypes.byref(h)), "ddmp_comm_create")
        self.h = h
        self._plans = {}
        # Exchange stream (round 6, DistributedTrainer(overlap_halo=True)): an OVERLAPPED halo exchange (start_halo_overlapped) is
        # enqueued on a stream of the communicator's own, behind an event of the caller's stream; the caller's stream waits for
        # its completion event right before it aggregates the boundary rows.  Blocking collectives (all-reduces, all-gathers,
        # ghost exchanges) stay on the caller's stream: routing them through the exchange stream as well was measured at one
        # rank with the RCCL loopback -- two cross-queue hand-offs of ~10 us per collective, ~100 collectives per step: 10.5
        # instead of 8.5 ms at 125k faces (profiles/r06_dist_overhead.txt).  No two operations of the communicator are ever in
        # flight at once, whatever RCCL does across streams: the exchange stream waits for an event recorded on the caller's
        # stream AFTER every earlier collective was enqueued there, and the caller's stream waits for the exchange's completion
        # event (before the boundary rows) before it enqueues the next collective -- the events serialise them in issue order, and
        # the issue order is the same on every rank.  Never under capture.
        self.use_xs = False
        self.xs = None

    def _on_xs(self, fn, defer=False):
        if not self.use_xs or torch.cuda.is_current_stream_capturing():
            fn()
            return None
        if self.xs is None:
            self.xs = torch.cuda.Stream(device=self.device)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        self.xs.wait_event(ready)
        with torch.cuda.stream(self.xs):
            fn()
            done = torch.cuda.Event()
            done.record(self.xs)
        if defer:
            return _StreamWait(done)
        torch.cuda.current_stream().wait_event(done)
        return None

    def plan_handle(self, plan: "HaloPlan"):
        key = id(plan)
        if key not in self._plans:
            ct = self.ct
            idx = np.ascontiguousarray(plan.send_idx, dtype=np.int64)
            sc = np.asarray(plan.send_counts, dtype=np.int64)
            rc = np.asarray(plan.recv_counts, dtype=np.int64)
            h = ct.c_void_p()
            with ops.on_device(self.device):
                _lib_check(self.L.ddmp_halo_plan_create(self.world_size, self.rank, plan.n_rows, plan.n_cols, idx.ctypes.data,
                                                        sc.ctypes.data, rc.ctypes.data, ct.byref(h)), "ddmp_halo_plan_create")
            self._plans[key] = (h, plan)
        return self._plans[key][0]

    def halo_exchange_native(self, plan, t, sums=None, defer=False):
        """``defer``: return a handle whose wait() orders the caller's stream behind the exchange (exchange stream only)."""
        h = self.plan_handle(plan)
        dt = ops._dt(t)

        def go():
            ws = ops.Workspace.get(self.L.ddmp_halo_pack_bytes(h, t.shape[1], dt), t.device)     # (per stream: the exchange stream's own)
            _lib_check(self.L.ddmp_halo_exchange(self.h, h, ops._p(t), t.stride(0), t.shape[1], dt, ops._p(ws), ws.numel(),
                                                 ops._p(sums), 0 if sums is None else sums.numel(), ops._stream()), "ddmp_halo_exchange")
        if defer:
            return self._on_xs(go, True)
        go()
        return t

    def all_reduce_sum(self, t):
        assert t.is_contiguous() and t.dtype in (torch.float32, torch.float64)
        _lib_check(self.L.ddmp_comm_allreduce_sum(self.h, ops._p(t), t.numel(), 1 if t.dtype == torch.float64 else 0, ops._stream()),
                   "ddmp_comm_allreduce_sum")
        return t

    def all_gather_rows(self, out, local):
        local = local.contiguous()
        _lib_check(self.L.ddmp_comm_allgather(self.h, ops._p(local), ops._p(out), local.numel() * local.element_size(), ops._stream()),
                   "ddmp_comm_allgather")
        return out

    def barrier(self):
        torch.cuda.synchronize(self.device)
        self.dist.barrier()

    def _check_inputs(self, plan):
        """Test tensors of the self-check + what torch.distributed delivers for them (reference collectives: run by EVERY rank,
        unconditionally and in the same order, before anything native can fail)."""
        dev = self.device
        ids = torch.from_numpy(np.asarray(plan.local_ids, dtype=np.float64)).to(dev)
        t = torch.zeros((plan.n_cols, 4), dtype=torch.float32, device=dev)
        t[: plan.n_rows, 0] = (ids[: plan.n_rows] % 8191.0).float()
        t[: plan.n_rows, 1] = torch.div(ids[: plan.n_rows], 8191.0, rounding_mode="floor").float()
        t[: plan.n_rows, 2] = float(self.rank)
        ref = t.clone()
        send = ref[: plan.n_rows].index_select(0, torch.from_numpy(plan.send_idx).to(dev)).contiguous()
        recv = ref[plan.n_rows: plan.n_cols]
        self.dist.all_to_all_single(recv, send, output_split_sizes=list(plan.recv_counts),
                                    input_split_sizes=list(plan.send_counts))
        sums = torch.arange(8, dtype=torch.float64, device=dev) + self.rank
        sums_ref = sums.clone()
        self.dist.all_reduce(sums_ref)
        return ids, t, ref, sums, sums_ref

    @staticmethod
    def _check_verdict(plan, ids, t, ref, sums, sums_ref):
        want = ids[plan.n_rows: plan.n_cols]
        got = t[plan.n_rows: plan.n_cols, 0].double() + 8191.0 * t[plan.n_rows: plan.n_cols, 1].double()
        return bool(torch.equal(t, ref) and torch.equal(got, want) and torch.equal(sums, sums_ref))

    def self_check(self, plan: "HaloPlan", other: "NativeComm" = None, other_plan: "HaloPlan" = None) -> bool:
        """World size > 1, before the first step: push a tensor of GLOBAL row ids through this communicator's grouped halo
        exchange and a float64 vector through its all-reduce, and compare with what torch.distributed delivers for the
        same plan (all_to_all_single / all_reduce: the path the gloo world-2 tests cover).  Every rank returns the same
        verdict (the flags are all-reduced through torch.distributed): False = do not use this backend.  The native
        send/recv path cannot be run with peers in the build loop (one-GPU boxes; RCCL refuses two ranks on one device),
        so it proves itself on the job's own plan before it is trusted.

        ``other`` / ``other_plan``: a second communicator that the trainer will drive CONCURRENTLY on a side stream (PosNet
        beside NormalNet); its exchange is enqueued on a second stream before the first one is waited for, i.e. both
        communicators are in flight on the device at once, as in the step.

        The torch.distributed reference collectives run outside the guarded region and the verdict all-reduce is reached at
        the same collective index on every rank whatever the native calls do.  A native exchange that HANGS (mismatched
        order, a peer that never posts) cannot be recovered in-process: a watchdog ends the process with exit code 17
        after DDMP_SELFCHECK_TIMEOUT seconds (default 180) and says which switch selects the torch.distributed backend."""
        import os
        import sys
        import threading
        dev = self.device
        with ops.on_device(dev):
            a = self._check_inputs(plan)
            b = other._check_inputs(other_plan) if other is not None else None
            torch.cuda.synchronize(dev)
        limit = float(os.environ.get("DDMP_SELFCHECK_TIMEOUT", "180"))

        def _give_up():
            sys.stderr.write("ddmp: the native RCCL self-check did not complete within %.0f s on rank %d -- exiting (17).  "
                             "DDMP_DIST_NATIVE=0 selects the torch.distributed backend, DDMP_DIST_STREAMS=0 a single "
                             "communicator.\n" % (limit, self.rank))
            sys.stderr.flush()
            os._exit(17)

        dog = threading.Timer(limit, _give_up)
        dog.daemon = True
        dog.start()
        ok = 1.0
        try:
            with ops.on_device(dev):
                side = torch.cuda.Stream(device=dev) if other is not None else None
                if other is not None:
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        other.halo_exchange_native(other_plan, b[1], b[3])
                self.halo_exchange_native(plan, a[1], a[3])
                torch.cuda.synchronize(dev)
                if not self._check_verdict(plan, *a):
                    ok = 0.0
                if other is not None and not self._check_verdict(other_plan, *b):
                    ok = 0.0
        except Exception:       # noqa: BLE001
            ok = 0.0
        finally:
            dog.cancel()
        flag = torch.tensor([ok], dtype=torch.float32, device=dev)
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN)
        return bool(flag.item() > 0.5)

    def close(self):
        """Destroy the halo plans and the communicator (idempotent; also run by the finalizer)."""
        plans, self._plans = self._plans, {}
        for h, _ in plans.values():
            self.L.ddmp_halo_plan_destroy(h)
        h, self.h = self.h, None
        if h is not None:
            self.L.ddmp_comm_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _lib_check(st, what):
    from . import _lib
    _lib.check(st, what)


class ThreadComm:
    """P logical ranks = P threads of one process sharing one device.  Collectives are rendezvous through a
    barrier; used to run the partitioned path against the unpartitioned one on a single GPU (or on CPU)."""

    class _Shared:
        def __init__(self, P):
            self.P = P
            self.barrier = threading.Barrier(P)
            self.slots = [None] * P

    def __init__(self, shared, rank):
        self.s, self.rank, self.world_size = shared, rank, shared.P

    @classmethod
    def make(cls, P) -> List["ThreadComm"]:
        sh = cls._Shared(P)
        return [cls(sh, r) for r in range(P)]

    def all_to_all(self, recv, send, recv_counts, send_counts):
        if send.is_cuda:
            torch.cuda.current_stream().synchronize()
        self.s.slots[self.rank] = (send, send_counts)
        self.s.barrier.wait()
        off = 0
        for src in range(self.world_size):
            n = recv_counts[src]
            if n:
                sbuf, scounts = self.s.slots[src]
                s0 = sum(scounts[:self.rank])
                recv[off:off + n].copy_(sbuf[s0:s0 + n])
            off += n
        if recv.is_cuda:
            torch.cuda.current_stream().synchronize()
        self.s.barrier.wait()

    def all_reduce_sum(self, t):
        if t.is_cuda:
            torch.cuda.current_stream().synchronize()
        self.s.slots[self.rank] = t.clone()
        self.s.barrier.wait()
        acc = self.s.slots[0].clone()
        for r in range(1, self.world_size):
            acc += self.s.slots[r]
        if t.is_cuda:
            torch.cuda.current_stream().synchronize()
        self.s.barrier.wait()
        t.copy_(acc)
        return t

    def all_gather_rows(self, out, local):
        if local.is_cuda:
            torch.cuda.current_stream().synchronize()
        self.s.slots[self.rank] = local
        self.s.barrier.wait()
        m = local.shape[0]
        for r in range(self.world_size):
            out[r * m:(r + 1) * m].copy_(self.s.slots[r])
        if out.is_cuda:
            torch.cuda.current_stream().synchronize()
        self.s.barrier.wait()
        return out

    def barrier(self):
        self.s.barrier.wait()


# ------------------------------------------------------------------------------------ distributed nets
def global_csr(edge_index: np.ndarray, n: int):
    return ops.csr_build_host(edge_index, n)


_global_tables = {}
_global_lock = threading.Lock()


def _shared_global_tables(dataset, n_mesh, P, face_owner):
    """The rank-independent part of the partition (owners, the two global CSRs, Morton keys), built ONCE per process and
    (dataset, P): logical ranks that live in one process (ThreadComm: tests, single-GPU emulation of an 8-way run) share
    it instead of each repeating O(global mesh) numpy work; one process per GPU builds it once anyway."""
    key = (id(dataset), id(n_mesh), P, None if face_owner is None else hash(np.asarray(face_owner).tobytes()))
    with _global_lock:
        hit = _global_tables.get(key)
        if hit is not None and hit[0] is dataset:
            return hit[1]
        V, F = len(n_mesh.vs), len(n_mesh.faces)
        if face_owner is None:
            face_owner = face_owner_rcb(np.asarray(n_mesh.fc, dtype=np.float64), P)
        vert_owner = vertex_owner_from_faces(n_mesh.faces, face_owner, V)
        ei = dataset.edge_index.cpu().numpy()
        fi = dataset.face_index.cpu().numpy()
        # local row order = RCB rank of the node (smoothed vertex positions / noisy face centroids), the same
        # locality numbering the single-device engine applies (round 5; Morton before)
        vkey = np.empty(V, dtype=np.int64)
        vkey[rcb_order(dataset.x_pos.detach().cpu().double().numpy(), 64)] = np.arange(V)
        fkey = np.empty(F, dtype=np.int64)
        fkey[rcb_order(np.asarray(n_mesh.fc, dtype=np.float64), 64)] = np.arange(F)
        vcsr, fcsr = global_csr(ei, V), global_csr(fi, F)
        if P > 1:                                                # chunks without a remote neighbour first (round 6)
            vkey = interior_first_keys(vcsr[0], vcsr[1], vert_owner, vkey, P)
            fkey = interior_first_keys(fcsr[0], fcsr[1], face_owner, fkey, P)
        # where every rank's owned rows (in its local order: increasing Morton key) go in the global arrays: the
        # replicated losses are fed by ONE all-gather of the owned pos | norm rows, padded to the largest shard
        vord, ford = np.lexsort((vkey, vert_owner)), np.lexsort((fkey, face_owner))
        vcnt, fcnt = np.bincount(vert_owner, minlength=P), np.bincount(face_owner, minlength=P)
        m = int((vcnt + fcnt).max())
        gather_dst = np.full((P, m), V + F, dtype=np.int64)            # pads land in a scratch row behind the arrays
        vo = np.concatenate([[0], np.cumsum(vcnt)])
        fo = np.concatenate([[0], np.cumsum(fcnt)])
        for r in range(P):
            gather_dst[r, :vcnt[r]] = vord[vo[r]:vo[r + 1]]
            gather_dst[r, vcnt[r]:vcnt[r] + fcnt[r]] = V + ford[fo[r]:fo[r + 1]]
        tables = dict(face_owner=face_owner, vert_owner=vert_owner, vcsr=vcsr, fcsr=fcsr,
                      vkey=vkey, fkey=fkey, gather_dst=gather_dst.reshape(-1), gather_rows=m)
        if len(_global_tables) > 4:
            _global_tables.clear()
        _global_tables[key] = (dataset, tables)
        return tables


# ------------------------------------------------------------------------------------ sharded losses
def _csr_rows(ptr: np.ndarray, idx: np.ndarray, rows: np.ndarray) -> np.ndarray:
    """Concatenated CSR rows `rows` of (ptr, idx)."""
    cnt = (ptr[rows + 1] - ptr[rows]).astype(np.int64)
    if cnt.sum() == 0:
        return np.empty(0, dtype=idx.dtype)
    base = np.repeat(ptr[rows].astype(np.int64) - np.concatenate([[0], np.cumsum(cnt)[:-1]]), cnt)
    return idx[base + np.arange(int(cnt.sum()), dtype=np.int64)]


class GhostPlan:
    """Exchange schedule of a ghost closure, with the fields :class:`GraphComm` / :class:`NativeComm` read from a
    :class:`HaloPlan` (owned rows first, then the ghost rows grouped by source rank and id)."""

    def __init__(self, ext_by_rank, owner: np.ndarray, rank: int, P: int, owned: np.ndarray):
        """``ext_by_rank[d]``: sorted global ids rank d needs (a superset of what it owns is fine); ``owned``: this rank's
        owned ids in the row order of its engine."""
        n = len(owner)
        self.rank, self.P, self.owned = rank, P, owned
        mine = ext_by_rank[rank]
        ghost = mine[owner[mine] != rank]
        self.halo = ghost[np.lexsort((ghost, owner[ghost]))]
        self.recv_counts = np.bincount(owner[self.halo], minlength=P).astype(np.int64).tolist()
        g2l = np.full(n, -1, dtype=np.int64)
        g2l[owned] = np.arange(len(owned))
        send, counts = [], []
        for d in range(P):
            ids = ext_by_rank[d] if d != rank else np.empty(0, dtype=np.int64)
            ids = ids[owner[ids] == rank]                               # ascending ids = the receiver's order
            send.append(g2l[ids])
            counts.append(len(ids))
        self.send_idx = np.concatenate(send).astype(np.int64) if send else np.empty(0, dtype=np.int64)
        assert (self.send_idx >= 0).all()
        self.send_counts = counts
        self.n_rows, self.n_cols = len(owned), len(owned) + len(self.halo)
        self.local_ids = np.concatenate([owned, self.halo])
        self.n_global = n


def _loss_closures(n_mesh, face_owner, vert_owner, P, loop):
    """For every rank: the faces and vertices its share of the losses (util/loss.py:16-160) reads.

    Owned rows = the rows whose loss terms the rank SUMS and whose gradient rows it keeps.  A loss term's gradient reaches
    rows of other ranks and, for the bilateral filter (fn_bnf_loss, util/loss.py:86-137), other ranks' terms reach owned
    rows; instead of exchanging inside the losses a rank recomputes what it needs on a ghost closure:

    * faces: ``2*loop + 1`` rings (f2f) around the owned faces -- the filtered normal n^t is exact where its ring is
      present, loss terms within ``loop`` rings send gradient to an owned normal, and their backward reads one more ring
      -- plus every face incident to an owned vertex (pos_norm_loss gradient, util/loss.py:140-160) and the mesh's LAST
      face (what f2f == -1 gathers in the reference: ``fn[f2f]`` with index -1);
    * vertices: the corners of those faces and two rings around the owned vertices (the Laplacian residual of a
      neighbour reads that neighbour's neighbours, util/loss.py:37-52).
    """
    faces = np.ascontiguousarray(n_mesh.faces, dtype=np.int64)
    f2f = np.ascontiguousarray(n_mesh.f2f, dtype=np.int64)
    V, F = len(n_mesh.vs), len(faces)
    e = np.asarray(n_mesh.edges, dtype=np.int64)
    src, dst = np.concatenate([e[:, 0], e[:, 1]]), np.concatenate([e[:, 1], e[:, 0]])
    o = np.argsort(src, kind="stable")
    vv_ptr = np.zeros(V + 1, dtype=np.int64)
    np.cumsum(np.bincount(src, minlength=V), out=vv_ptr[1:])
    vv_idx = dst[o]
    flat = faces.reshape(-1)
    vf_ptr = np.zeros(V + 1, dtype=np.int64)
    np.cumsum(np.bincount(flat, minlength=V), out=vf_ptr[1:])
    vf_idx = np.argsort(flat, kind="stable") // 3
    rings = 2 * int(loop) + 1
    fext, vext = [], []
    for r in range(P):
        fo, vo = np.flatnonzero(face_owner == r), np.flatnonzero(vert_owner == r)
        fin = np.zeros(F, dtype=bool)
        fin[fo] = True
        front = fo
        for _ in range(rings):
            nb = f2f[front].reshape(-1)
            nb = nb[nb >= 0]
            nb = np.unique(nb[~fin[nb]])
            if nb.size == 0:
                break
            fin[nb] = True
            front = nb
        fin[_csr_rows(vf_ptr, vf_idx, vo)] = True
        fin[F - 1] = True
        vin = np.zeros(V, dtype=bool)
        vin[vo] = True
        front = vo
        for _ in range(2):
            nb = _csr_rows(vv_ptr, vv_idx, front)
            nb = np.unique(nb[~vin[nb]])
            if nb.size == 0:
                break
            vin[nb] = True
            front = nb
        f_ids = np.flatnonzero(fin)
        vin[faces[f_ids].reshape(-1)] = True
        fext.append(f_ids)
        vext.append(np.flatnonzero(vin))
    return fext, vext


class _LocalMesh:
    """The attributes :class:`loss.LossEngine` reads from a mesh (vs, fn, faces, f2f, edges), for a rank's sub-mesh."""


class LossShard:
    """One rank's sub-mesh for the sharded losses: local connectivity over [owned | ghost] rows (+ a copy of the mesh's
    last face as the last local face, which is what the loss kernels gather for f2f == -1), the owned-row masks of the
    partial sums, and the two ghost exchange plans (vertex positions, face normals)."""

    def __init__(self, n_mesh, g: dict, rank: int, P: int, loop: int, owned_v: np.ndarray, owned_f: np.ndarray):
        key = "loss_closures_%d" % loop
        with _global_lock:
            if key not in g:
                g[key] = _loss_closures(n_mesh, g["face_owner"], g["vert_owner"], P, loop)
        fext, vext = g[key]
        faces = np.ascontiguousarray(n_mesh.faces, dtype=np.int64)
        f2f = np.ascontiguousarray(n_mesh.f2f, dtype=np.int64)
        V, F = len(n_mesh.vs), len(faces)
        self.vplan = GhostPlan(vext, g["vert_owner"].astype(np.int64), rank, P, owned_v)
        self.fplan = GhostPlan(fext, g["face_owner"].astype(np.int64), rank, P, owned_f)
        v_ids, f_ids = self.vplan.local_ids, self.fplan.local_ids
        nvl, nfl = len(v_ids), len(f_ids)
        gv = np.full(V, -1, dtype=np.int64)
        gv[v_ids] = np.arange(nvl)
        gf = np.full(F, -1, dtype=np.int64)
        gf[f_ids] = np.arange(nfl)
        self.last_row = int(gf[F - 1])                                     # where the mesh's last face lives locally
        f_all = np.concatenate([f_ids, [F - 1]])                           # + its copy, the last local face
        lf = gv[faces[f_all]]
        assert (lf >= 0).all()
        l2 = f2f[f_all]
        l2 = np.where(l2 >= 0, gf[np.maximum(l2, 0)], -1)                  # neighbours outside the closure: -1 (outermost
        l2[-1] = -1                                                        # ghost ring only; those rows are never counted)
        e = np.asarray(n_mesh.edges, dtype=np.int64)
        le = gv[e]
        le = le[(le >= 0).all(axis=1)]
        m = _LocalMesh()
        m.vs = np.asarray(n_mesh.vs, dtype=np.float64)[v_ids]
        m.fn = np.asarray(n_mesh.fn, dtype=np.float64)[f_all]
        m.faces, m.f2f, m.edges = lf, l2, le
        m.vf_faces = nfl                                                   # the copy is not incident to anything
        self.mesh = m
        self.own_v = np.zeros(nvl, dtype=np.uint8)
        self.own_v[: len(owned_v)] = 1
        self.own_f = np.zeros(nfl + 1, dtype=np.uint8)
        self.own_f[: len(owned_f)] = 1
        self.V_glob, self.F_glob = V, F


class ShardedData:
    """What a rank's two engines read: local slices of the static inputs + the two halo plans."""

    def __init__(self, dataset, n_mesh, rank: int, P: int, face_owner=None):
        V, F = len(n_mesh.vs), len(n_mesh.faces)
        g = _shared_global_tables(dataset, n_mesh, P, face_owner)
        self.face_owner, self.vert_owner = g["face_owner"], g["vert_owner"]
        self.vplan = HaloPlan(*g["vcsr"], self.vert_owner, rank, P, order_key=g["vkey"])
        self.fplan = HaloPlan(*g["fcsr"], self.face_owner, rank, P, order_key=g["fkey"])
        self.gather_dst, self.gather_rows = g["gather_dst"], g["gather_rows"]
        self.z1 = dataset.z1.detach().cpu()[torch.from_numpy(self.vplan.local_ids)]
        self.z2 = dataset.z2.detach().cpu()[torch.from_numpy(self.fplan.local_ids)]
        self.x_pos = dataset.x_pos.detach().cpu()[torch.from_numpy(self.vplan.owned)]
        self.V, self.F = V, F
        self._g, self._n_mesh, self.rank, self.P = g, n_mesh, rank, P

    def loss_shard(self, loop: int) -> "LossShard":
        return LossShard(self._n_mesh, self._g, self.rank, self.P, loop, self.vplan.owned, self.fplan.owned)


class DistributedTrainer:
    """main.py:88-110 on P ranks (see module docstring)."""

    def __init__(self, posnet, normnet, sharded: ShardedData, n_mesh, backend, device, pos_lr=0.01, norm_lr=0.01,
                 k=(3.0, 4.0, 4.0, 4.0, 1.0), grad_crip=0.8, bnfloop=1, betas=(0.9, 0.999), eps=1e-8,
                 bnf_start_epoch=100, ops_mod=None, loss_engine=None, losses=None, backend_pos=None, use_graph=None,
                 overlap_halo=None):
        """``overlap_halo`` (default: on with more than one rank; env DDMP_DIST_SPLIT=0 switches it off, =1 forces it at one rank
        too, =noverlap keeps the two launches but waits for every exchange first -- the bit-identity check): every aggregation runs as two launches -- the rows that reference no halo row (HaloPlan.n_int: the leading chunks of
        the interior-first local order) while the layer's halo exchange travels on the communicator's exchange stream, the
        boundary rows behind it (engine.GcnEngine ``split``).  Eager path only: a captured iteration keeps whole-graph launches.

        ``use_graph`` (default: env DDMP_DIST_GRAPH=1; it OVERRIDES DDMP_DIST_STREAMS: a captured iteration uses one
        communicator on one stream; with more than one rank it is refused unless DDMP_DIST_GRAPH_PEERS=1, see below): replay the
        partitioned iteration as ONE hipGraph -- every kernel and,
        with the native RCCL backend, every collective is enqueued on the capturing stream(s) from C (csrc/comm.hip); the Adam
        step count lives on the device as in FusedTrainer(use_graph=True).  First call eager, second captured, then replayed;
        re-captured when the BNF gate opens.  Native backend only (torch.distributed's collectives are issued from Python).

        ``losses``: "sharded" (every rank evaluates the loss terms of its own rows on a ghost closure; two small
        all-reduces of partial sums; default on the GPU) or "replicated" (one all-gather of pos | norm, every rank
        runs the whole-mesh losses; what a caller-supplied ``loss_engine`` implies).  Env: DDMP_DIST_LOSSES.

        ``backend_pos``: a SECOND communicator for PosNet.  Given one (make_distributed_trainer creates it for the native
        RCCL backend), the step runs PosNet on a second HIP stream beside NormalNet -- forward, then backward + gradient
        all-reduce + Adam update -- exactly like the single-device FusedTrainer(overlap=True); each stream owns its
        communicator, so the two nets' collectives never share one.  Off: DDMP_DIST_STREAMS=0."""
        import contextlib
        # graphs and engines allocate on the CURRENT device (ddmp_graph_create): enter the device context here too
        # (the CPU tests run this class on a torch stand-in of ops: nothing to enter there)
        ctx = ops.on_device(device) if torch.device(device).type == "cuda" else contextlib.nullcontext()
        import os
        if use_graph is None:
            use_graph = os.environ.get("DDMP_DIST_GRAPH", "0") == "1"
        self.use_graph = bool(use_graph) and isinstance(backend, NativeComm) and torch.device(device).type == "cuda"
        if self.use_graph and backend.world_size > 1 and os.environ.get("DDMP_DIST_GRAPH_PEERS") != "1":
            # capturing RCCL send / recv and all-reduces into a hipGraph has only ever run on a one-rank loopback
            # communicator (no multi-GPU box in the build loop), the start-up self-check validates EAGER exchanges only, and a hang
            # inside a capture or a replay is not covered by its watchdog.  With peers the capture is refused until someone has
            # validated it on hardware: DDMP_DIST_GRAPH_PEERS=1 takes that responsibility.
            import warnings
            warnings.warn("DDMP_DIST_GRAPH / use_graph with %d ranks: the captured iteration is unvalidated with peers -- running "
                          "eagerly (set DDMP_DIST_GRAPH_PEERS=1 to capture anyway)" % backend.world_size)
            self.use_graph = False
        if self.use_graph:
            # ONE communicator on ONE stream under capture.  Measured on ROCm 7.2 with DDMP_COMM_LOOPBACK=1 (tests/test_gpu_multi.py,
            # profiles/r06_dist_overhead.txt): the captured iteration replays correctly with one communicator on one stream; a
            # second communicator in the same capture ends in a SIGSEGV (two streams) or never returns (one stream), and -- round
            # 6 -- so does ONE communicator whose operations are captured on two forked streams (SIGSEGV inside the capture): RCCL
            # operations cannot be captured on a forked stream here.  Two streams under capture are therefore taken only where
            # no RCCL call is issued at all (one rank without DDMP_COMM_LOOPBACK: 8.02 instead of 8.30 ms at 125k faces).
            want2 = backend.world_size == 1 and os.environ.get("DDMP_COMM_LOOPBACK") != "1"
            backend_pos = backend if want2 else None
        if overlap_halo is None:
            e = os.environ.get("DDMP_DIST_SPLIT")
            overlap_halo = (backend.world_size > 1) if e is None else e != "0"
        self.overlap_halo = bool(overlap_halo) and not self.use_graph
        with ctx:
            self._init(posnet, normnet, sharded, n_mesh, backend, device, pos_lr, norm_lr, k, grad_crip, bnfloop, betas,
                       eps, bnf_start_epoch, ops_mod, loss_engine, losses, backend_pos)
        self._graphs, self._warm = {}, False
        if self.use_graph:
            self.interleaved = False
            self._t_dev = torch.zeros(1, dtype=torch.int32, device=device)
            self._coef = [torch.zeros(2, dtype=torch.float32, device=device) for _ in range(2)]

    def _init(self, posnet, normnet, sharded, n_mesh, backend, device, pos_lr, norm_lr, k, grad_crip, bnfloop, betas, eps,
              bnf_start_epoch, ops_mod, loss_engine, losses, backend_pos):
        from .engine import GcnEngine, POS_WIDTHS, NORM_WIDTHS
        self.ops = ops_mod or ops
        self.backend, self.device = backend, device
        self.posnet, self.normnet = posnet, normnet
        sd = sharded
        self.sd = sd
        vg = self.ops.Graph.from_csr_host(sd.vplan.rowptr, sd.vplan.col, sd.vplan.dinv, sd.vplan.n_cols)
        fg = self.ops.Graph.from_csr_host(sd.fplan.rowptr, sd.fplan.col, sd.fplan.dinv, sd.fplan.n_cols)
        import os
        self.two_streams = (backend_pos is not None and torch.device(device).type == "cuda"
                            and os.environ.get("DDMP_DIST_STREAMS", "1") != "0")
        self.backend_pos = backend_pos if self.two_streams else backend
        self._side = torch.cuda.Stream(device=device) if self.two_streams else None
        def halves(plan):
            """The interior / boundary halves of a rank's graph (None: nothing to split -- no interior chunk, or no boundary)."""
            if not getattr(self, "overlap_halo", False) or not 0 < plan.n_int < plan.n_rows:
                return None
            mk = self.ops.Graph.from_csr_host
            return (mk(plan.rowptr, plan.col, plan.dinv, plan.n_cols, rows=(0, plan.n_int)),
                    mk(plan.rowptr, plan.col, plan.dinv, plan.n_cols, rows=(plan.n_int, plan.n_rows)), plan.n_int)
        # (the interleaved driver keeps one net's collective in flight while the other net issues its own: with ONE native
        #  communicator that would put two of its operations in flight at once -- the exchange stream is for the plain drivers)
        interleaved = os.environ.get("DDMP_DIST_INTERLEAVE", "0") == "1"
        for b in {id(backend): backend, id(self.backend_pos): self.backend_pos}.values():
            if hasattr(b, "use_xs"):
                b.use_xs = bool(getattr(self, "overlap_halo", False)) and not interleaved and torch.device(device).type == "cuda"
        self.peng = GcnEngine(vg, POS_WIDTHS, 0, sd.z1.to(device), sd.x_pos.to(device),
                              comm=GraphComm(self.backend_pos, sd.vplan, device), n_total=sd.V,
                              dtype=getattr(posnet, "feature_dtype", torch.float32), split=halves(sd.vplan),
                              overlap=getattr(self, "overlap_halo", False))
        self.neng = GcnEngine(fg, NORM_WIDTHS, 1, sd.z2.to(device), None,
                              comm=GraphComm(backend, sd.fplan, device), n_total=sd.F,
                              dtype=getattr(normnet, "feature_dtype", torch.float32), split=halves(sd.fplan),
                              overlap=getattr(self, "overlap_halo", False))
        for net, eng in ((posnet, self.peng), (normnet, self.neng)):
            if hasattr(net, "attach_engine"):
                net.attach_engine(eng)                          # net(data) then runs on this rank's shard
            else:
                net._engine = eng
        self.owned_v = torch.from_numpy(sd.vplan.owned).to(device)
        self.owned_f = torch.from_numpy(sd.fplan.owned).to(device)
        import os
        if losses is None:
            losses = "replicated" if loss_engine is not None else os.environ.get("DDMP_DIST_LOSSES", "sharded")
        # (with losses="sharded", ``loss_engine`` may be a factory (local_mesh, shard) -> engine: the CPU tests' stand-in)
        if losses not in ("sharded", "replicated"):
            raise ValueError("losses must be 'sharded' or 'replicated', got %r" % (losses,))
        self.losses = losses
        self._full = None                                        # pos | norm of the whole mesh: built when asked for
        self._full_epoch = -1
        if losses == "sharded":
            from .loss import LossEngine
            ls = sd.loss_shard(bnfloop)
            self.lshard = ls
            self.vghost, self.fghost = GraphComm(backend, ls.vplan, device), GraphComm(backend, ls.fplan, device)
            f = dict(dtype=torch.float32, device=device)
            # exchange buffers are 4 floats wide (16-byte rows: what the pack kernel / in-place receives move)
            self.vbuf, self.fbuf = torch.zeros((ls.vplan.n_cols, 4), **f), torch.zeros((ls.fplan.n_cols, 4), **f)
            self.pos_ext, self.norm_ext = torch.empty((ls.vplan.n_cols, 3), **f), torch.empty((ls.fplan.n_cols + 1, 3), **f)

            class _Shard:
                pass
            sh = _Shard()
            sh.own_v, sh.own_f = torch.from_numpy(ls.own_v).to(device), torch.from_numpy(ls.own_f).to(device)
            sh.V_glob, sh.F_glob, sh.all_reduce = ls.V_glob, ls.F_glob, backend.all_reduce_sum
            sh.rank_shard = ls
            loss_engine = loss_engine(ls.mesh, sh) if callable(loss_engine) else LossEngine(ls.mesh, device, bnfloop=bnfloop, k=k, shard=sh)
        elif loss_engine is None:
            from .loss import LossEngine
            loss_engine = LossEngine(n_mesh, device, bnfloop=bnfloop, k=k)
        self.loss_engine = loss_engine
        self.pos_lr, self.norm_lr, self.grad_crip, self.betas, self.eps = pos_lr, norm_lr, grad_crip, betas, eps
        self.bnf_start_epoch = bnf_start_epoch
        self.m = [torch.zeros_like(posnet.arena.data), torch.zeros_like(normnet.arena.data)]
        self.v = [torch.zeros_like(posnet.arena.data), torch.zeros_like(normnet.arena.data)]
        self.sumsq = torch.zeros(1, dtype=torch.float64, device=device)
        self.epoch = 0
        self.t = 0
        self.lossbuf = None
        # DDMP_DIST_INTERLEAVE=1: the two nets alternate at their collectives (async_op=True).  Off by default: that
        # path has only run at world size 1 on RCCL and through gloo on CPU (no multi-GPU box in the build loop);
        # the default is the blocking form, which the threaded-rank GPU tests exercise kernel for kernel.
        self.interleaved = os.environ.get("DDMP_DIST_INTERLEAVE", "0") == "1"

    def barrier(self):
        self.backend.barrier()

    def _assemble_full(self):
        """pos | norm of the whole mesh on every rank: ONE all-gather of the owned rows (padded to the largest shard),
        scattered to their global places.  COLLECTIVE: every rank has to call it (gather_pos / gather_norm do) at the same point; cached until the next step."""
        key = (getattr(self.peng, "n_forward", 0), getattr(self.neng, "n_forward", 0))    # any forward (step, net(data),
        if self._full_epoch == key and self._full is not None:                            # eval) invalidates the cache
            return self._full
        sd, dev = self.sd, self.device
        if self._full is None:
            self._full = torch.zeros((sd.V + sd.F + 1, 3), dtype=torch.float32, device=dev)   # + the scratch row of the pads
            self._gather_dst = torch.from_numpy(sd.gather_dst).to(dev)
            self._gather_send = torch.zeros((sd.gather_rows, 3), dtype=torch.float32, device=dev)
            self._gather_recv = torch.empty((self.backend.world_size * sd.gather_rows, 3), dtype=torch.float32, device=dev)
        pos_loc, norm_loc = self.peng.result, self.neng.result
        nv, nf = pos_loc.shape[0], norm_loc.shape[0]
        self._gather_send[:nv].copy_(pos_loc)
        self._gather_send[nv:nv + nf].copy_(norm_loc)
        self.backend.all_gather_rows(self._gather_recv, self._gather_send)
        self._full.index_copy_(0, self._gather_dst, self._gather_recv)
        self._full_epoch = key
        return self._full

    def gather_pos(self):
        with self.ops.on_device(self.device):
            return self._assemble_full()[: self.sd.V]

    def gather_norm(self):
        with self.ops.on_device(self.device):
            return self._assemble_full()[self.sd.V: self.sd.V + self.sd.F]

    # (no .pos / .norm attributes here: reading them would be a COLLECTIVE -- `if rank == 0: tr.pos` deadlocks; FusedTrainer
    #  has the attributes, the partitioned trainer the two methods above)

    def _sharded_losses(self, pos_loc, norm_loc, gate):
        """Ghost rows of pos / norm from their owners (two grouped exchanges), then this rank's share of the losses."""
        ls = self.lshard
        nv, nf = ls.vplan.n_rows, ls.fplan.n_rows
        self.vbuf[:nv, :3].copy_(pos_loc)
        self.fbuf[:nf, :3].copy_(norm_loc)
        self.vghost.halo_exchange(self.vbuf, nv)
        self.fghost.halo_exchange(self.fbuf, nf)
        self.pos_ext.copy_(self.vbuf[:, :3])
        self.norm_ext[:-1].copy_(self.fbuf[:, :3])
        self.norm_ext[-1].copy_(self.norm_ext[ls.last_row])               # the copy of the mesh's last face
        lossbuf, dpos, dnorm = self.loss_engine.forward_backward(self.pos_ext, self.norm_ext, gate)
        return lossbuf, dpos[:nv], dnorm[:nf]

    def check_scales(self) -> int:
        from .trainer import nonfinite_check
        healed = self.peng.check_scales() + self.neng.check_scales()
        nonfinite_check((("PosNet", self.posnet), ("NormalNet", self.normnet)))
        return healed

    @torch.no_grad()
    def step(self):
        with self.ops.on_device(self.device):
            return self._step()

    def _step(self):
        self.epoch += 1
        self.t += 1
        gate = 0.0 if self.epoch <= self.bnf_start_epoch else 1.0
        if not self.use_graph:
            self.lossbuf = self._iteration(gate, False)
            return self.lossbuf[5]
        self._t_dev.fill_(self.t - 1)
        if not self._warm:                                       # first call: eager (allocates workspaces, primes the scales)
            self._warm = True
            self.lossbuf = self._iteration(gate, True)
            return self.lossbuf[5]
        g = self._graphs.get(gate)
        if g is None:
            graph = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(graph):
                cap = self._iteration(gate, True)
            g = self._graphs[gate] = (graph, cap)
        g[0].replay()
        self.peng.n_forward = getattr(self.peng, "n_forward", 0) + 1      # (a replay is a forward: the all-gather cache key)
        self.neng.n_forward = getattr(self.neng, "n_forward", 0) + 1
        self.lossbuf = g[1]
        return self.lossbuf[5]

    def _iteration(self, gate, dev_adam):
        o = self.ops
        V = self.sd.V
        pa, na = self.posnet.arena.data, self.normnet.arena.data
        pg, ng = self.posnet._grad_arena, self.normnet._grad_arena
        if dev_adam:
            o.adam_prepare(self._t_dev, self.pos_lr, self._coef[0], self.betas)
            self._t_dev -= 1                                                # one counter, two learning rates
            o.adam_prepare(self._t_dev, self.norm_lr, self._coef[1], self.betas)

        def pos_update():
            if dev_adam:
                o.adam_step_dev_(pa, pg, self.m[0], self.v[0], self._coef[0], self.betas, self.eps)
            else:
                o.adam_step_(pa, pg, self.m[0], self.v[0], self.pos_lr, self.t, self.betas, self.eps)
        # the two nets alternate at their collectives: one net's halo exchange / BatchNorm all-reduce is in flight
        # while the other net's kernels run
        if self.two_streams:
            self._fork()
            with torch.cuda.stream(self._side):
                self.peng.forward(pa, update_running=True)
            self.neng.forward(na, update_running=True)
            self._join()
        elif self.interleaved:
            interleave(self.peng.forward_steps(pa, update_running=True), self.neng.forward_steps(na, update_running=True))
        else:
            self.peng.forward(pa, update_running=True)
            self.neng.forward(na, update_running=True)
        if self.losses == "sharded":
            lossbuf, dpos_loc, dnorm_loc = self._sharded_losses(self.peng.result, self.neng.result, gate)
        else:
            # replicated: every rank runs the whole-mesh losses on the all-gathered pos | norm and keeps its rows
            full = self._assemble_full()
            lossbuf, dpos, dnorm = self.loss_engine.forward_backward(full[:V], full[V:V + self.sd.F], gate)
            dpos_loc, dnorm_loc = dpos.index_select(0, self.owned_v), dnorm.index_select(0, self.owned_f)
        if self.two_streams:
            self._fork()
            if dpos_loc.is_cuda and not torch.cuda.is_current_stream_capturing():
                dpos_loc.record_stream(self._side)               # (allocated on this stream, read on the other)
            with torch.cuda.stream(self._side):                  # PosNet: backward, gradient all-reduce, Adam -- beside NormalNet
                self.peng.backward(pa, pg, dpos_loc)
                self.posnet._reduce_grads()
                pos_update()
            self.neng.backward(na, ng, dnorm_loc)
            self.normnet._reduce_grads()
        else:
            if self.interleaved:
                interleave(self.peng.backward_steps(pa, pg, dpos_loc), self.neng.backward_steps(na, ng, dnorm_loc))
            else:
                self.peng.backward(pa, pg, dpos_loc)
                self.neng.backward(na, ng, dnorm_loc)
            self.posnet._reduce_grads()
            self.normnet._reduce_grads()
            pos_update()
        o.grad_sumsq(ng, out=self.sumsq)
        if dev_adam:
            o.adam_step_dev_(na, ng, self.m[1], self.v[1], self._coef[1], self.betas, self.eps,
                             clip_sumsq=self.sumsq, max_norm=self.grad_crip)
        else:
            o.adam_step_(na, ng, self.m[1], self.v[1], self.norm_lr, self.t, self.betas, self.eps,
                         clip_sumsq=self.sumsq, max_norm=self.grad_crip)
        if self.two_streams:
            self._join()
        return lossbuf

    def _fork(self):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._side.wait_event(ev)

    def _join(self):
        ev = torch.cuda.Event()
        ev.record(self._side)
        torch.cuda.current_stream().wait_event(ev)


def make_distributed_trainer(n_mesh, s_mesh, dataset, device, rank, world, bnfloop=1, backend=None, nets=None, **kw):
    from .networks import PosNet, NormalNet
    backend_pos = kw.pop("backend_pos", None)
    if backend is None:
        # Default on the GPU: the library's own RCCL communicators (csrc/comm.hip: pack + ONE grouped send/recv per layer
        # with the BatchNorm sums in the same group, enqueued from C on the caller's stream) -- two of them, so that PosNet
        # runs on a second stream beside NormalNet.  DDMP_DIST_NATIVE=0 (or RCCL not resolvable) falls back to
        # torch.distributed's process group: blocking collectives on one stream.
        import os
        backend = None
        if torch.device(device).type == "cuda" and os.environ.get("DDMP_DIST_NATIVE", "1") != "0":
            try:
                backend = NativeComm(device)
                # Two communicators driven concurrently on two streams have run at world size 1 only (no multi-GPU box in
                # the build loop): with peers they are OPT-IN (DDMP_DIST_STREAMS=1) and then have to pass the concurrent
                # self-check below; the default with peers is one communicator, one stream.
                graph = kw.get("use_graph") if kw.get("use_graph") is not None else os.environ.get("DDMP_DIST_GRAPH", "0") == "1"
                if os.environ.get("DDMP_DIST_STREAMS", "1" if world == 1 else "0") != "0" and not graph:
                    backend_pos = NativeComm(device)                 # (a captured iteration uses ONE communicator)
            except Exception as e:      # noqa: BLE001  (every rank takes the same branch: RCCL is there for all or none)
                import warnings
                warnings.warn("native RCCL backend unavailable (%s): falling back to torch.distributed" % (e,))
                backend = backend_pos = None
        if backend is None:
            backend = TorchDistComm()
    if nets is None:
        torch.manual_seed(0)                   # identical initial parameters on every rank
        nets = (PosNet(device), NormalNet(device))
    posnet, normnet = nets
    sharded = ShardedData(dataset, n_mesh, rank, world)
    if isinstance(backend, NativeComm) and backend.world_size > 1 and not kw.get("_skip_self_check"):
        # the job will route every RCCL call of a communicator through its exchange stream (overlap_halo, the default with peers):
        # the start-up check then runs through that stream too
        import os
        ov = kw.get("overlap_halo")
        if ov is None:
            ov = os.environ.get("DDMP_DIST_SPLIT", "1") != "0"
        graph = kw.get("use_graph") if kw.get("use_graph") is not None else os.environ.get("DDMP_DIST_GRAPH", "0") == "1"
        if ov and not graph:
            backend.use_xs = True
            if backend_pos is not None:
                backend_pos.use_xs = True
        good = backend.self_check(sharded.fplan)
        if good and backend_pos is not None:                     # each alone, then both in flight at once
            good = backend_pos.self_check(sharded.vplan) and backend.self_check(sharded.fplan, backend_pos, sharded.vplan)
        if not good:
            import warnings
            warnings.warn("the native RCCL backend failed its self-check against torch.distributed on this job's halo plan: "
                          "using torch.distributed (blocking collectives, one stream)")
            backend, backend_pos = TorchDistComm(), None
    kw.pop("_skip_self_check", None)
    return DistributedTrainer(posnet, normnet, sharded, n_mesh, backend, device, bnfloop=bnfloop, backend_pos=backend_pos, **kw)
