"""150 iterations (through the BNF gate at 101) at 144k faces: float32 (f16x3 GEMMs), float32 (bf16x6), bf16 features --
loss trajectory and final MAD per seed.  usage: long_run_dtypes.py [seeds, comma separated]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dual_dmp_amd import ops, synth                                       # noqa: E402
from dual_dmp_amd.datamaker import dataset_from_meshes                    # noqa: E402
from dual_dmp_amd.networks import PosNet, NormalNet                       # noqa: E402
from dual_dmp_amd.trainer import FusedTrainer                             # noqa: E402
from dual_dmp_amd.evaluate import Evaluator                               # noqa: E402

dev = torch.device("cuda:0")
v, f = synth.torus(380, 190)
gt, noisy, smooth = synth.make_triplet(v, f)
ev = Evaluator(noisy, gt.fn, dev)
seeds = [int(m) for m in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 3]
print("torus %d faces; MAD of the noisy input %.4f deg" % (len(noisy.faces), ev.mad(torch.from_numpy(np.asarray(noisy.vs, dtype=np.float32)).to(dev))))
for seed in seeds:
    for name, mode, dt in (("f32 f16x3", 13, torch.float32), ("f32 bf16x6", 6, torch.float32), ("bf16 features", 13, torch.bfloat16)):
        ops.set_gemm_mode(mode)
        data = dataset_from_meshes(noisy, smooth)
        data.to(dev)
        torch.manual_seed(seed)
        tr = FusedTrainer(PosNet(dev, dtype=dt), NormalNet(dev, dtype=dt), data, noisy, use_graph=True, overlap=True)
        ls, mads, healed = [], {}, 0
        for it in range(150):
            ls.append(tr.step().item())
            if it % 10 == 9:
                healed += tr.check_scales()
            if it + 1 in (50, 100, 150):
                mads[it + 1] = ev.mad(tr.pos)
        print("seed %d %-14s loss @1 %.5f @10 %.5f @50 %.5f @100 %.5f @101 %.5f @150 %.5f | MAD @50 %.4f @100 %.4f @150 %.4f | healed %d"
              % (seed, name, ls[0], ls[9], ls[49], ls[99], ls[100], ls[149], mads[50], mads[100], mads[150], healed), flush=True)
ops.set_gemm_mode(13)
