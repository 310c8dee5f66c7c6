import sys; sys.path.insert(0,"/root/repo")
import torch
from dual_dmp_amd import ops
dev=torch.device("cuda:0"); n=1000000
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/iters*1e3
for mode in (6,0):
    ops.set_gemm_mode(mode)
    print("== mode", mode)
    for M,K in ((256,128),(128,256),(64,128),(128,64),(32,64),(64,32),(32,8)):
        G=torch.randn(n,M,device=dev); Z=torch.randn(n,K,device=dev); dW=torch.empty(M,K,device=dev)
        W=torch.randn(M,K,device=dev); Y=torch.empty(n,M,device=dev); X=torch.empty(n,K,device=dev)
        t1=timeit(lambda: ops.gemm_tn(G,Z,out=dW)); t2=timeit(lambda: ops.gemm_nt(Z,W,out=Y)); t3=timeit(lambda: ops.gemm_nn(G,W,out=X))
        by=4.0*n*(K+M)
        print("M=%3d K=%3d  tn %6.0f us (%4.0f GB/s)  nt(K->M) %6.0f us  nn(M->K) %6.0f us"%(M,K,t1,by/t1/1e3,t2,t3))
