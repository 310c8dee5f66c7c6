#!/usr/bin/env python3
"""Loop of wide f16x3 GEMMs (row-register kernel) with power / clock samples: see rr_ablation.sh."""
import json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dual_dmp_amd import ops

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


threading.Thread(target=sampler, daemon=True).start()
n = 1000000
for K, M in ((512, 512), (256, 256), (256, 512)):
    A = torch.randn(n, K, device=dev); W = torch.randn(M, K, device=dev) / K ** 0.5; Y = torch.empty(n, M, device=dev)
    slots = torch.zeros(1, 4, device=dev)
    ops.gemm_next_scales(slots[0], None, prime=True); ops.gemm_nt(A, W, out=Y); ops.gemm_scales_roll(slots)
    torch.cuda.synchronize()
    t0 = time.time(); it = 0
    while time.time() - t0 < 5.0:
        for _ in range(20):
            ops.gemm_next_scales(slots[0], None); ops.gemm_nt(A, W, out=Y)
        torch.cuda.synchronize(); it += 20
    t1 = time.time()
    ss = [s for s in samples if t0 + 1.0 <= s[0] <= t1]
    pw = [s[1] for s in ss if s[1] == s[1]]
    clk = sorted(set(str(s[2]) for s in ss))
    print("gemm_nt 1M x %d -> %d   %7.1f us/launch   power %.0f W   sclk %s .. %s" % (
        K, M, (t1 - t0) / it * 1e6, sum(pw) / max(len(pw), 1), clk[0] if clk else "?", clk[-1] if clk else "?"), flush=True)
stop[0] = True
