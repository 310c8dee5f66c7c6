#!/usr/bin/env python3
"""A/B probe of the f16x3 operand conversion (round 5): every wide GEMM form of the step at 1M rows with caller scale slots, the
time per call and a checksum of the output BITS.  Run once per library (DDMP_LIB=build_abl/libddmp_head.so | unset) -- the
checksums of two libraries that convert to the same f16 terms are equal."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dual_dmp_amd import ops
dev = torch.device("cuda:0")
n = int(os.environ.get("ROWS", "1000000"))
torch.manual_seed(1)
tag = os.environ.get("DDMP_LIB", "tree").split("/")[-1]


def bits(t):
    v = t.contiguous().view(torch.int32).to(torch.int64).flatten()
    w = (torch.arange(v.numel(), device=v.device, dtype=torch.int64) % 1021) + 1
    return "%016x" % (int((v * w).sum().item()) & 0xFFFFFFFFFFFFFFFF)


def run(name, call, out, slots):
    slots.zero_()
    call(dict(scales=(slots[0], slots[1], True)))
    ops.gemm_scales_roll(slots)
    kw = dict(scales=(slots[0], slots[1], False))
    call(kw); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        call(kw)
    e1.record(); torch.cuda.synchronize()
    print("%-14s %-22s %7.0f us  %s" % (tag, name, e0.elapsed_time(e1) / 5 * 1e3, bits(out)), flush=True)


for M, K in ((512, 512), (256, 256), (512, 256), (256, 128)):
    dz = torch.randn(n, M, device=dev); yb = torch.randn(n, M, device=dev); z = torch.randn(n, K, device=dev) * 3
    # a wide dynamic range per column, as real activations have
    z *= torch.exp(torch.randn(K, device=dev) * 2)
    bn4 = torch.rand(4, M, device=dev) + 0.5; c10 = torch.rand(2, M, device=dev) * 0.1
    sc = torch.rand(K, device=dev) + 0.5; sh = torch.randn(K, device=dev)
    dW = torch.empty(M, K, device=dev)
    slots = torch.zeros(2, 4, device=dev)
    run("tn %dx%d bnbwd" % (M, K), lambda kw: ops.gemm_tn_bnbwd(dz, yb, z, bn4, c10, out=dW, **kw), dW, slots)
    run("tn %dx%d pro" % (M, K), lambda kw: ops.gemm_tn(dz, z, out=dW, pro=(sc, sh), **kw), dW, slots)
    run("tn %dx%d plain" % (M, K), lambda kw: ops.gemm_tn(dz, z, out=dW, **kw), dW, slots)
    del dz, yb, z
for K, M in ((512, 512), (256, 512), (256, 256), (128, 256), (512, 256)):
    a = torch.randn(n, K, device=dev) * torch.exp(torch.randn(K, device=dev) * 2)
    w = torch.randn(M, K, device=dev) / K ** 0.5
    b = torch.randn(M, device=dev)
    y = torch.empty(n, M, device=dev)
    sums = torch.empty(2 * 512, dtype=torch.float64, device=dev)
    sc = torch.rand(K, device=dev) + 0.5; sh = torch.randn(K, device=dev)
    slots = torch.zeros(2, 4, device=dev)
    one = lambda kw: dict(scales=(kw["scales"][0], None, kw["scales"][2]))
    run("nt %d->%d stats" % (K, M), lambda kw: ops.gemm_nt_stats(a, w, sums, out=y, bias=b, **one(kw)), y, slots)
    run("nt %d->%d pro" % (K, M), lambda kw: ops.gemm_nt(a, w, out=y, pro=(sc, sh), **one(kw)), y, slots)
    # dgrads: dX[n, K] = g[n, M] . w[M, K]
    g = torch.randn(n, M, device=dev) * torch.exp(torch.randn(M, device=dev))
    yb = torch.randn(n, M, device=dev)
    bn4 = torch.rand(4, M, device=dev) + 0.5; c10 = torch.rand(2, M, device=dev) * 0.1
    dx = torch.empty(n, K, device=dev)
    if ops.gemm_bnbwd_supported(M, K, n):
        run("nn %d<-%d bnbwd" % (K, M), lambda kw: ops.gemm_nn_bnbwd(g, yb, w, bn4, c10, out=dx, **one(kw)), dx, slots)
    if ops.gemm_nn_bnred_supported(M, K, n):
        yp = torch.randn(n, K, device=dev); bn4p = torch.rand(4, K, device=dev) + 0.5
        run("nn %d<-%d bnred" % (K, M), lambda kw: ops.gemm_nn_bnred(g, w, yp, bn4p, sums, out=dx, **one(kw)), dx, slots)
        del yp
    run("nn %d<-%d plain" % (K, M), lambda kw: ops.gemm_nn(g, w, out=dx, **one(kw)), dx, slots)
    del a, y, g, yb, dx
