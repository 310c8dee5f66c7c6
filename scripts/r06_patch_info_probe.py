#!/usr/bin/env python3
"""How the LDS-patch gather takes the chunks of the bench mesh's two graphs (as the engines build them: RCB order)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dual_dmp_amd import _lib
from dual_dmp_amd.networks import PosNet, NormalNet
from dual_dmp_amd.trainer import FusedTrainer

faces = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
dev = torch.device("cuda:0")
gt, noisy, smooth, data = bench.build_case(faces, "native")
torch.manual_seed(0)
data.to(dev)
tr = FusedTrainer(PosNet(dev), NormalNet(dev), data, noisy, use_graph=False, overlap=False)
L = _lib.lib()
for name, eng in (("vertex (PosNet)", tr.peng), ("face (NormalNet)", tr.neng)):
    g = eng.g
    kd, nh, ns = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    L.ddmp_graph_patch_info(g._h, ctypes.byref(kd), ctypes.byref(nh), ctypes.byref(ns))
    print("%s graph: %d rows, %d chunks, patch_kd %d, heavy %d, split %d" % (name, g.n_rows, (g.n_rows + 63) // 64, kd.value, nh.value, ns.value))
