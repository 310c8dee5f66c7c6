#!/usr/bin/env python3
"""Long-horizon MAD of the CPU oracle (oracle/ddmp_oracle.py, the restatement of main.py:86-149) as a committed fixture:
the reference's learning loop on a 320-face mesh (icosphere-2 + the noisemaker's noise, 150 iterations: the BNF gate of
main.py:101-102 opens at 101), five weight seeds, ``Loss.mad(o1_mesh.fn, gt_mesh.fn)`` every 10 epochs as main.py:117-127
evaluates it.  tests/test_gpu_path.py::test_long_horizon_mad_matches_oracle_fixture runs the HIP path from the same weights.

Adam on this problem is chaotic (the oracle's own float32 and float64 runs separate ~10x per iteration), so what is pinned is
the STATISTIC the reference reports -- the MAD reached -- per seed and on average, not a trajectory to rounding; the fixture
also holds the oracle's float64 run of the same seeds, whose distance from its float32 run is the noise floor of the comparison.

    python tests/golden/make_mad_fixture.py            (CPU, ~2 min; needs nothing outside the repo)
"""
import importlib.util
import os
import sys
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("ddmp_oracle", os.path.join(ROOT, "oracle", "ddmp_oracle.py"))
oracle = importlib.util.module_from_spec(spec)
sys.modules["ddmp_oracle"] = oracle
spec.loader.exec_module(oracle)
from dual_dmp_amd import synth                      # noqa: E402  (mesh generator + Mesh tables: inputs, golden-pinned)
from dual_dmp_amd.mesh import Mesh                  # noqa: E402

ITERS, SEEDS, EVERY = 150, [100, 101, 102, 103, 104], 10


def weights_crc(sd):
    c = 0
    for k in sorted(sd):
        c = zlib.crc32(sd[k].detach().cpu().numpy().tobytes(), c)
    return c


def mad_of(pos, noisy, gt):
    o = Mesh.__new__(Mesh)
    o.vs, o.faces = np.asarray(pos, dtype=np.float64), noisy.faces
    Mesh.compute_face_normals(o)
    return float(oracle.mad_np(o.fn, gt.fn))


def run(seed, noisy, smooth, gt, dtype):
    torch.manual_seed(seed)
    rp, rn = oracle.PosNetRef(), oracle.NormalNetRef()
    crc = (weights_crc(rp.state_dict()), weights_crc(rn.state_dict()))
    if dtype == torch.float64:
        rp, rn = rp.double(), rn.double()
    odata = oracle.OracleDataset(noisy, smooth)
    if dtype == torch.float64:
        for k in ("z1", "z2", "x_pos", "x_norm"):
            setattr(odata, k, getattr(odata, k).double())
    mesh = noisy
    if dtype == torch.float64:                       # (the loss tables the reference keeps in float32: util/mesh.py:189-197)
        import copy
        mesh = copy.copy(noisy)
        mesh.v2v_mat, mesh.v_dims = noisy.v2v_mat.double(), noisy.v_dims.double()
    args = oracle.StepArgs()
    op = torch.optim.Adam(rp.parameters(), lr=args.pos_lr)
    on = torch.optim.Adam(rn.parameters(), lr=args.norm_lr)
    mads, losses = [], []
    for ep in range(1, ITERS + 1):
        lo, p, n, _ = oracle.train_step(rp, rn, op, on, odata, mesh, args, ep)
        losses.append(float(lo))
        if ep % EVERY == 0:
            mads.append(mad_of(p.detach().double().numpy(), noisy, gt))
    return crc, mads, losses


def main():
    torch.set_num_threads(4)
    v, f = synth.icosphere(2)
    gt, noisy, smooth = synth.make_triplet(v, f)
    out = dict(iters=ITERS, every=EVERY, seeds=np.array(SEEDS), noisy_mad=mad_of(noisy.vs, noisy, gt))
    m32, m64, l32, crcs = [], [], [], []
    for s in SEEDS:
        crc, mads, losses = run(s, noisy, smooth, gt, torch.float32)
        m32.append(mads); l32.append(losses); crcs.append(crc)
        try:
            _, mads64, _ = run(s, noisy, smooth, gt, torch.float64)
        except Exception as e:      # noqa: BLE001
            print("float64 run failed:", e)
            mads64 = [np.nan] * len(mads)
        m64.append(mads64)
        print("seed %d: f32 MAD %s | f64 final %.3f" % (s, " ".join("%.2f" % x for x in mads), mads64[-1]), flush=True)
    out.update(mad_f32=np.array(m32), mad_f64=np.array(m64), loss_f32=np.array(l32), weights_crc=np.array(crcs, dtype=np.int64))
    np.savez_compressed(os.path.join(HERE, "mad_oracle_ico2.npz"), **out)
    a = out["mad_f32"][:, -1]
    print("noisy MAD %.3f; final MAD f32 mean %.3f std %.3f; f64 mean %.3f" % (out["noisy_mad"], a.mean(), a.std(), np.nanmean(out["mad_f64"][:, -1])))


if __name__ == "__main__":
    main()
