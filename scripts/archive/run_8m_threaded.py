#!/usr/bin/env python3
"""BASELINE.json configs[3] on ONE GPU: the 8,000,000-face / 4,000,000-vertex mesh as 8 logical ranks (threads, ThreadComm)
with Morton face partition + 1-hop halos, bf16 features (288 GB hold 8M faces only in that mode: ~170 GB), against the
unpartitioned run.  usage: run_8m_threaded.py [faces_u faces_v] [P]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dual_dmp_amd import dist as D, synth
from dual_dmp_amd.datamaker import dataset_from_meshes
from dual_dmp_amd.networks import PosNet, NormalNet
from dual_dmp_amd.trainer import FusedTrainer

nu, nv = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2500, 1600)
P = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda:0")
t0 = time.perf_counter()
v, f = synth.torus(nu, nv)
gt, noisy, smooth = synth.make_triplet(v, f)
data = dataset_from_meshes(noisy, smooth)
print("mesh: %d faces / %d verts, built in %.1f s" % (len(noisy.faces), len(noisy.vs), time.perf_counter() - t0), flush=True)
BF = torch.bfloat16

t0 = time.perf_counter()
torch.manual_seed(0)
tr = FusedTrainer(PosNet(dev, dtype=BF), NormalNet(dev, dtype=BF), data, noisy)
base = tr.step().item()
torch.cuda.synchronize()
t1 = time.perf_counter()
l2 = tr.step().item()
torch.cuda.synchronize()
print("unpartitioned bf16: first iteration (incl. set-up) %.1f s, second %.3f s, loss %.6f -> %.6f, peak memory %.1f GB"
      % (t1 - t0, time.perf_counter() - t1, base, l2, torch.cuda.max_memory_allocated() / 1e9), flush=True)
base_pos = tr.pos.clone()
del tr
torch.cuda.empty_cache()
torch.cuda.reset_peak_memory_stats()

nets = []
for _ in range(P):
    torch.manual_seed(0)
    nets.append((PosNet(dev, dtype=BF), NormalNet(dev, dtype=BF)))
comms = D.ThreadComm.make(P)
res, errs = {}, []
t0 = time.perf_counter()

def work(r):
    try:
        torch.cuda.set_device(0)
        t = D.make_distributed_trainer(noisy, smooth, data, dev, r, P, backend=comms[r], nets=nets[r])
        ta = time.perf_counter()
        loss = t.step().item()
        tb = time.perf_counter()
        loss2 = t.step().item()
        res[r] = (loss, loss2, tb - ta, time.perf_counter() - tb, t.peng.n_rows, t.peng.n_cols, t.neng.n_rows, t.neng.n_cols,
                  (lambda g: g.clone() if r == 0 else None)(t.gather_pos()))       # collective: every rank calls it
    except BaseException as e:      # noqa: BLE001
        import traceback; traceback.print_exc()
        errs.append(e)
        comms[r].s.barrier.abort()

ths = [threading.Thread(target=work, args=(r,)) for r in range(P)]
[t.start() for t in ths]
[t.join() for t in ths]
assert not errs, errs
print("%d logical ranks: set-up + 2 iterations in %.1f s, peak memory %.1f GB" % (P, time.perf_counter() - t0, torch.cuda.max_memory_allocated() / 1e9))
for r in range(P):
    l, l2_, d1, d2, vr, vc, fr, fc, _ = res[r]
    print("rank %d: loss %.6f -> %.6f (unpartitioned %.6f -> %.6f)  iteration %.2f / %.2f s  owned verts %d (+%d halo)  faces %d (+%d halo)"
          % (r, l, l2_, base, l2, d1, d2, vr, vc - vr, fr, fc - fr))
    assert abs(l - base) <= 2e-3 * abs(base), (l, base)
print("max |pos - unpartitioned| after 2 iterations: %.3e" % float((res[0][8] - base_pos).abs().max()))
print("OK")
