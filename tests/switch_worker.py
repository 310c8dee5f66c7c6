"""Worker of tests/test_gpu_switches.py: ONE training iteration of the fused trainer on the 144,400-face torus (every large-mesh
route on) under the environment it was started with; writes loss, outputs and Adam first moments (= (1 - beta1) * gradients) to
the .npz named on the command line.  The library reads most of its A/B switches once per process, hence a process per setting."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    out, dtype = sys.argv[1], (torch.bfloat16 if len(sys.argv) > 2 and sys.argv[2] == "bf16" else torch.float32)
    dev = torch.device("cuda:0")
    from dual_dmp_amd import synth
    from dual_dmp_amd.datamaker import dataset_from_meshes
    from dual_dmp_amd.networks import PosNet, NormalNet
    from dual_dmp_amd.trainer import FusedTrainer
    v, f = synth.torus(380, 190)
    v, f = synth.permute_vertices(v, f, 3)
    gt, noisy, smooth = synth.make_triplet(v, f)
    data = dataset_from_meshes(noisy, smooth)
    data.to(dev)
    torch.manual_seed(5)
    posnet, normnet = PosNet(dev, dtype=dtype), NormalNet(dev, dtype=dtype)
    tr = FusedTrainer(posnet, normnet, data, noisy, bnfloop=1)
    loss = tr.step().item()
    torch.cuda.synchronize()
    np.savez(out, loss=np.float64(loss), pos=tr.pos.float().cpu().numpy(), norm=tr.norm.float().cpu().numpy(),
             m0=tr.m[0].cpu().numpy(), m1=tr.m[1].cpu().numpy())


if __name__ == "__main__":
    main()
