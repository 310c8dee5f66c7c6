import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dual_dmp_amd import ops
dev=torch.device("cuda:0"); n=1000000
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/iters*1e3
for M,K in ((256,128),(128,256),(64,128),(128,64),(32,64),(64,32)):
    G=torch.randn(n,M,device=dev); Z=torch.randn(n,K,device=dev); dW=torch.empty(M,K,device=dev)
    t1=timeit(lambda: ops.gemm_tn(G,Z,out=dW))
    ref=(G[:20000].double().t()@Z[:20000].double())
    print("tn M=%3d K=%3d  %6.0f us (%4.0f GB/s)"%(M,K,t1,4.0*n*(K+M)/t1/1e3))
