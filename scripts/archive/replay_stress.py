"""Eager / hipGraph / two-stream / two-stream hipGraph iterations of one small mesh, N times over in one process: losses, outputs
and parameters must be bit-identical across all of them (tests/test_gpu_path.py::test_graph_replay_is_bit_identical_to_eager runs it
once; this is the stress form that found a 5 % flake).  usage: replay_stress.py [repetitions] [f32|bf16] [big]
`big`: a 48,400-face torus instead of the 532-face grid -- both nets above the 20k-row threshold of the row-register / row-panel
GEMMs and of the fused BatchNorm routes, so the two-stream graph runs those kernels beside each other."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dual_dmp_amd import synth
from dual_dmp_amd.datamaker import dataset_from_meshes
from dual_dmp_amd.networks import PosNet, NormalNet
from dual_dmp_amd.trainer import FusedTrainer
dev = torch.device("cuda:0")
BIG = len(sys.argv) > 3 and sys.argv[3] == "big"
v, f = synth.torus(220, 110) if BIG else synth.open_grid(20, 15)
v, f = synth.permute_vertices(v, f, 4)
gt, noisy, smooth = synth.make_triplet(v, f)
data = dataset_from_meshes(noisy, smooth); data.to(dev)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
DT = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float32
modes = (False, True, "overlap", "overlap+graph")
ref = None
fails = 0
for rep in range(N):
    for graph in modes:
        torch.manual_seed(5)
        posnet, normnet = PosNet(dev, dtype=DT), NormalNet(dev, dtype=DT)
        tr = FusedTrainer(posnet, normnet, data, noisy, bnfloop=2, bnf_start_epoch=4,
                          use_graph=graph in (True, "overlap+graph"), overlap=str(graph).startswith("overlap"))
        losses = [tr.step().item() for _ in range(9)]
        cur = (losses, tr.pos.clone(), tr.norm.clone(), posnet.arena.data.clone(), normnet.arena.data.clone())
        if ref is None:
            ref = cur
        else:
            same = ref[0] == cur[0] and all(torch.equal(a, b) for a, b in zip(ref[1:], cur[1:]))
            if not same:
                fails += 1
                first = next((i for i, (a, b) in enumerate(zip(ref[0], cur[0])) if a != b), None)
                print("rep %d mode %s differs: first differing loss at step %s, max|dpos| %.3e, max|dW_pos| %.3e max|dW_norm| %.3e"
                      % (rep, graph, first, float((ref[1] - cur[1]).abs().max()), float((ref[3] - cur[3]).abs().max()),
                         float((ref[4] - cur[4]).abs().max())), flush=True)
print("failures: %d of %d runs" % (fails, N * len(modes) - 1))
