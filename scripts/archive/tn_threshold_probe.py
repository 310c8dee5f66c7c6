import sys; sys.path.insert(0,"/root/repo")
import torch
from dual_dmp_amd import ops
dev=torch.device("cuda:0")
def timeit(fn, iters=30):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/iters*1e3
for n in (62296, 125000):
    for M,K in ((512,512),(256,256),(512,256)):
        G=torch.randn(n,M,device=dev); Z=torch.randn(n,K,device=dev); dW=torch.empty(M,K,device=dev)
        t=timeit(lambda: ops.gemm_tn(G,Z,out=dW))
        print("n=%6d tn M=%3d K=%3d %7.1f us %6.1f TF"%(n,M,K,t,2.0*n*M*K/t/1e6))
