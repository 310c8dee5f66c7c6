#!/usr/bin/env python3
"""The reference's own mesh sizes (fandisk: 12,946 faces; BASELINE.json configs[0] / [4]): ms per iteration of the
13,068-face cube (synth.cube_cad(33)) eager on one stream / as one replayed hipGraph / with PosNet on a second stream, and
(MODE=trace) a few eager iterations for rocprofv3 --kernel-trace.
    python scripts/small_mesh_probe.py [faces-side n, default 33]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dual_dmp_amd import synth
from dual_dmp_amd.datamaker import dataset_from_meshes
from dual_dmp_amd.networks import PosNet, NormalNet
from dual_dmp_amd.trainer import FusedTrainer

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 33
v, f = synth.cube_cad(n)
gt, noisy, smooth = synth.make_triplet(v, f)
data = dataset_from_meshes(noisy, smooth); data.to(dev)


def run(graph, overlap, iters=200):
    torch.manual_seed(0)
    tr = FusedTrainer(PosNet(dev), NormalNet(dev), data, noisy, use_graph=graph, overlap=overlap)
    for _ in range(5):
        tr.step().item()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        loss = tr.step().item()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3, loss


if os.environ.get("MODE") == "trace":
    print(run(False, False, iters=3))
elif os.environ.get("MODE") == "trace_graph":                      # one replayed hipGraph per iteration, two streams
    print(run(True, os.environ.get("STREAMS", "2") == "2", iters=20))
else:
    print("%d faces / %d verts" % (len(f), len(v)))
    for graph, overlap in ((False, False), (True, False), (True, True)):
        ms, loss = run(graph, overlap)
        print("graph=%d two_streams=%d  %.3f ms/iteration  loss %.6f" % (graph, overlap, ms, loss), flush=True)
