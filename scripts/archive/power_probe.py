import sys, os
sys.path.insert(0, "/root/repo")
import torch
from dual_dmp_amd import ops
dev = torch.device("cuda:0")
n, K, M = 1000000, 512, 512
def timeit(fn, iters=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
Y = torch.empty(n, M, device=dev)
for name, A, W in (("randn", torch.randn(n, K, device=dev), torch.randn(M, K, device=dev) / K ** 0.5),
                   ("zeros", torch.zeros(n, K, device=dev), torch.zeros(M, K, device=dev)),
                   ("ones", torch.ones(n, K, device=dev), torch.ones(M, K, device=dev)),
                   ("small-int", torch.randint(0, 4, (n, K), device=dev).float(), torch.randint(0, 4, (M, K), device=dev).float())):
    us = timeit(lambda: ops.gemm_nt(A, W, out=Y))
    print("%-10s %8.0f us  %6.1f TF-eq" % (name, us, 2.0 * n * K * M / us / 1e6))
