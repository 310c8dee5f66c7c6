"""cProfile of the eager (no hipGraph) iteration at a shard-sized mesh: where the HOST time of the multi-GPU path goes.
usage: python scripts/host_profile.py [faces] [steps]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

faces = int(sys.argv[1]) if len(sys.argv) > 1 else 125000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
import torch.distributed as dist  # noqa: E402
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from dual_dmp_amd.dist import make_distributed_trainer  # noqa: E402
from dual_dmp_amd.networks import PosNet, NormalNet  # noqa: E402
gt, noisy, smooth, data = bench.build_case(faces, "native")
torch.manual_seed(0)
tr = make_distributed_trainer(noisy, smooth, data, dev, 0, 1, nets=(PosNet(dev), NormalNet(dev)))
for _ in range(5):
    tr.step().item()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("faces %d: host issue %.2f ms/step, wall %.2f ms/step" % (faces, t_issue / steps * 1e3, t_all / steps * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    tr.step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
dist.destroy_process_group()
