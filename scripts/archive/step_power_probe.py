#!/usr/bin/env python3
"""Average socket power / shader clock (rocm-smi samples) while (a) the training step, (b) only its gathers, (c) only wide
f16x3 GEMMs, (d) only BatchNorm passes run in a loop: is the step power-capped, and in which phases?"""
import json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dual_dmp_amd import ops, synth
from dual_dmp_amd.datamaker import dataset_from_meshes
from dual_dmp_amd.networks import PosNet, NormalNet
from dual_dmp_amd.trainer import FusedTrainer

dev = torch.device("cuda:0")
samples, stop = [], [False]


def sampler():
    while not stop[0]:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
            d = json.loads(out)
            c = d[sorted(d)[0]]
            p = [v for k, v in c.items() if "ower" in k and "W" in k]
            s = [v for k, v in c.items() if k.startswith("sclk")]
            samples.append((time.time(), float(p[0]) if p else float("nan"), s[0] if s else "?"))
        except Exception as e:      # noqa: BLE001
            samples.append((time.time(), float("nan"), repr(e)[:60]))
        time.sleep(0.15)


def phase(name, fn, seconds=6.0):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        n += 5
    t1 = time.time()
    ss = [s for s in samples if t0 + 1.0 <= s[0] <= t1]
    pw = [s[1] for s in ss if s[1] == s[1]]
    clk = sorted(set(str(s[2]) for s in ss))
    print("%-28s %7.2f ms/iter   power avg %.0f W (min %.0f max %.0f, %d samples)  sclk %s" % (
        name, (t1 - t0) / n * 1e3, sum(pw) / max(len(pw), 1), min(pw or [0]), max(pw or [0]), len(pw), clk[:4]), flush=True)


th = threading.Thread(target=sampler, daemon=True)
th.start()
faces = int(os.environ.get("FACES", "1000000"))
nv = int(round((faces / 4.0) ** 0.5)); nu = faces // (2 * nv)
v, f = synth.torus(nu, nv)
gt, noisy, smooth = synth.make_triplet(v, f)
data = dataset_from_meshes(noisy, smooth); data.to(dev)
torch.manual_seed(0)
tr = FusedTrainer(PosNet(dev), NormalNet(dev), data, noisy, use_graph=True, overlap=True)
for _ in range(3):
    tr.step().item()
phase("idle (sleep)", lambda: time.sleep(0.05), 3.0)
phase("training step (graph)", lambda: tr.step())
n = 1000000
A = torch.randn(n, 512, device=dev); W = torch.randn(512, 512, device=dev) / 512 ** 0.5; Y = torch.empty(n, 512, device=dev)
slots = torch.zeros(1, 4, device=dev)
ops.gemm_next_scales(slots[0], None, prime=True); ops.gemm_nt(A, W, out=Y); ops.gemm_scales_roll(slots)
def g():
    ops.gemm_next_scales(slots[0], None); ops.gemm_nt(A, W, out=Y)
phase("gemm_nt 1M x 512 x 512", g)
G = torch.randn(n, 512, device=dev); dW = torch.empty(512, 512, device=dev)
phase("gemm_tn 1M x 512 x 512", lambda: ops.gemm_tn(G, A, out=dW))
eng = tr.neng
X = torch.randn(eng.n_cols, 512, device=dev); O = torch.empty(eng.n_rows, 512, device=dev)
phase("spmm face graph C=512", lambda: ops.spmm(eng.g, X, out=O))
sums = torch.zeros(1024, dtype=torch.float64, device=dev)
phase("bn_stats 1M x 512", lambda: ops.bn_stats(X, sums=sums, n_rows=eng.n_rows))
src = torch.empty(256 * 1024 * 1024, dtype=torch.float32, device=dev).normal_(); dst = torch.empty_like(src)
phase("device copy 1 GiB", lambda: dst.copy_(src))
stop[0] = True
