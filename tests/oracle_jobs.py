"""The CPU-oracle half of tests/test_gpu_path.py's teacher-forced comparison, as a picklable job: at 144,400 faces it is 50-100 s
of PyTorch-CPU work during which the GPU idles, so the two big cases are started in worker processes when the GPU session begins
(tests/conftest.py) and the tests pick the records up; every other case computes its record inline.  Test infrastructure only."""
import copy
import importlib.util
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def load_oracle():
    mod = sys.modules.get("ddmp_oracle")
    if mod is None:
        spec = importlib.util.spec_from_file_location("ddmp_oracle", os.path.join(ROOT, "oracle", "ddmp_oracle.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules["ddmp_oracle"] = mod
        spec.loader.exec_module(mod)
    return mod


def flipped(vf, hub):
    from dual_dmp_amd import synth
    v, f = vf
    f = synth.flip_edges(v, f, rounds=10, seed=1)
    f = synth.add_hub(v, f, hub, 24)
    hist = synth.valence_histogram(f, len(v))
    assert len(hist) - 1 == 24 and hist[3] > 0 and hist[10:].sum() > 1, hist
    return v, f


def case(which="ico3"):
    from dual_dmp_amd import synth
    from dual_dmp_amd.datamaker import dataset_from_meshes
    v, f = {"ico3": lambda: synth.icosphere(3), "grid": lambda: synth.open_grid(12, 9),
            "cad33": lambda: synth.cube_cad(33),         # 13,068 faces: the fandisk stand-in (README.md:57 of the reference)
            "grid24": lambda: synth.open_grid(24, 17),
            "torus48k": lambda: synth.torus(220, 110),   # 48,400 faces / 24,200 verts: row-panel routes on both graphs
            "torus144k": lambda: synth.torus(380, 190),  # 144,400 faces / 72,200 verts: every bench route is on
            # irregular valence (round 5): random edge flips (valence 3 ... 12+) and a valence-24 hub
            "flip": lambda: flipped(synth.torus(30, 14), 17),
            "flip144k": lambda: flipped(synth.torus(380, 190), 1000)}[which]()
    v, f = synth.permute_vertices(v, f, 3)
    gt, noisy, smooth = synth.make_triplet(v, f)
    return gt, noisy, smooth, dataset_from_meshes(noisy, smooth)


def oracle_nets(oracle, sd_pos, sd_norm, dtype=None):
    import torch
    posnet, normnet = oracle.PosNetRef(), oracle.NormalNetRef()
    posnet.load_state_dict(sd_pos)
    normnet.load_state_dict(sd_norm)
    if dtype == torch.float64:
        posnet.double()
        normnet.double()
    return posnet, normnet


def oracle_inputs(oracle, noisy, smooth, dtype):
    import torch
    odata = oracle.OracleDataset(noisy, smooth)
    mesh = noisy
    if dtype == torch.float64:
        for k in ("z1", "z2", "x_pos"):
            setattr(odata, k, getattr(odata, k).double())
        mesh = types.SimpleNamespace(vs=noisy.vs, fn=noisy.fn, faces=noisy.faces, f2f=noisy.f2f,
                                     v2v_mat=noisy.v2v_mat.double(), v_dims=noisy.v_dims.double())
    return odata, mesh


def oracle_grads(oracle, rp, rn, noisy, smooth, args, epoch, dtype):
    """Pre-clip gradients of one iteration from the current oracle state, evaluated in `dtype`."""
    import torch
    p2, n2 = copy.deepcopy(rp), copy.deepcopy(rn)
    if dtype == torch.float64:
        p2.double()
        n2.double()
    odata, omesh = oracle_inputs(oracle, noisy, smooth, dtype)
    p2.train(); n2.train()
    p2.zero_grad(); n2.zero_grad()
    total, _ = oracle.losses(p2(odata), n2(odata), omesh, args, epoch)
    total.backward()
    return ({n: p.grad.detach().clone() for n, p in p2.named_parameters()},
            {n: p.grad.detach().clone() for n, p in n2.named_parameters()})


def snapshot(net, opt):
    names = {p: n for n, p in net.named_parameters()}
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    m = {names[p]: st["exp_avg"].clone() for p, st in opt.state.items()}
    v = {names[p]: st["exp_avg_sq"].clone() for p, st in opt.state.items()}
    return sd, m, v


def teacher_forced_oracle(which, k, bnfloop, ep0, iters, check_at, threads=0, meshes=None):
    """The oracle's `iters` iterations of main.py:88-110 from seed 11; for every iteration in `check_at` a record of what the
    HIP trainer is given (the complete state BEFORE the iteration) and what it is compared with (the iteration's loss, terms,
    outputs, float32 / float64 gradients, parameters after the update).  Plain tensors only: crosses a process boundary."""
    import torch
    if threads:
        torch.set_num_threads(threads)
    oracle = load_oracle()
    gt, noisy, smooth, _ = meshes if meshes is not None else case(which)
    torch.manual_seed(11)
    sd_pos, sd_norm = oracle.PosNetRef().state_dict(), oracle.NormalNetRef().state_dict()
    rp, rn = oracle_nets(oracle, sd_pos, sd_norm)
    odata, omesh = oracle_inputs(oracle, noisy, smooth, torch.float32)
    args = oracle.StepArgs(bnfloop=bnfloop, k1=k[0], k2=k[1], k3=k[2], k4=k[3], k5=k[4])
    op = torch.optim.Adam(rp.parameters(), lr=args.pos_lr)
    on = torch.optim.Adam(rn.parameters(), lr=args.norm_lr)
    recs = {}
    for it in range(1, iters + 1):
        rec = None
        if it in check_at:
            rec = {"state": [snapshot(rp, op), snapshot(rn, on)],
                   "g32": oracle_grads(oracle, rp, rn, noisy, smooth, args, ep0 + it, torch.float32),
                   "g64": oracle_grads(oracle, rp, rn, noisy, smooth, args, ep0 + it, torch.float64)}
        ref_loss, ref_pos, ref_norm, parts = oracle.train_step(rp, rn, op, on, odata, omesh, args, ep0 + it)
        if rec is not None:
            rec.update(loss=ref_loss, pos=ref_pos, norm=ref_norm, parts=parts,
                       after=[{n: p.detach().clone() for n, p in net.named_parameters()} for net in (rp, rn)])
            recs[it] = rec
    return recs


# the two cases worth a worker process (tests/test_gpu_path.py: *_at_144k), keyed by the arguments the tests pass
DEFAULT_K = (3.0, 4.0, 4.0, 4.0, 1.0)                    # main.py:22-26
BIG_JOBS = {"torus144k": ("torus144k", DEFAULT_K, 1, 0, 2, (1, 2)), "flip144k": ("flip144k", DEFAULT_K, 1, 0, 1, (1,))}
_FUTURES = {}
_POOL = None


def start_big_jobs(threads=12):
    """Submit BIG_JOBS to two spawned worker processes (idempotent).  The parent may already hold a GPU context: spawn, not fork."""
    global _POOL
    if _POOL is not None:
        return
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    _POOL = ProcessPoolExecutor(max_workers=len(BIG_JOBS), mp_context=mp.get_context("spawn"))
    for name, a in BIG_JOBS.items():
        _FUTURES[name] = _POOL.submit(teacher_forced_oracle, *a, threads)


def big_job(name):
    """The record of a BIG_JOBS case: from its worker when start_big_jobs() ran, else None (the caller computes it inline)."""
    f = _FUTURES.get(name)
    if f is None:
        return None
    try:
        return f.result(timeout=3000)
    except Exception as e:      # noqa: BLE001  (a worker that could not start must not fail the comparison: compute inline)
        print("oracle_jobs: worker for %s failed (%s: %s); computing inline" % (name, type(e).__name__, e), file=sys.stderr)
        return None


def shutdown():
    global _POOL
    if _POOL is not None:
        _POOL.shutdown(wait=False, cancel_futures=True)
        _POOL = None
