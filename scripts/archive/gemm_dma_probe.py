"""What would an all-DMA f16x3 row-panel GEMM cost?  The bf16 row-panel kernel (A and W planes by LDS-DMA, one MFMA product) run at
K' = 2K moves the bytes of the float32 problem at K (A: 4 B per element, W: two 16-bit planes) and issues 2/3 of the f16x3 kernel's
MFMAs; next to it the float32 kernels as they are.  usage: gemm_dma_probe.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dual_dmp_amd import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
dev = torch.device("cuda:0")


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for K, M in ((512, 512), (256, 512), (512, 256), (256, 256)):
    A = torch.randn(n, K, device=dev)
    W = torch.randn(M, K, device=dev) / K ** 0.5
    Y = torch.empty(n, M, device=dev)
    t32 = timeit(lambda: ops.gemm_nt(A, W, out=Y))
    Ab = torch.randn(n, 2 * K, device=dev).to(torch.bfloat16)
    Wb = torch.randn(M, 2 * K, device=dev) / K ** 0.5
    Yb = torch.empty(n, M, device=dev, dtype=torch.bfloat16)
    t16 = timeit(lambda: ops.gemm_nt(Ab, Wb, out=Yb))
    Ab1 = Ab[:, :K].contiguous()
    Wb1 = Wb[:, :K].contiguous()
    t16k = timeit(lambda: ops.gemm_nt(Ab1, Wb1, out=Yb))
    gb = n * (K + M) * 4 / 1e9
    print("K=%3d M=%3d: f32 f16x3 panel %6.0f us (%.2f TB/s) | bf16 K'=2K %6.0f us (A bytes of f32, 2/3 of the MFMAs, Y half) | bf16 K %6.0f us"
          % (K, M, t32, gb / t32 * 1e3, t16, t16k))
