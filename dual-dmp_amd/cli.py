"""Command lines with the flags and defaults of the reference's ``main.py:15-37`` and ``main4real.py:12-32``.

    python main.py -i datasets/<name> [--pos_lr .01 --norm_lr .01 --iter 1000 --k1 3 --k2 4 --k3 4 --k4 4 --k5 1
                                       --grad_crip .8 --bnfloop 1 --gpu 0]
    python main4real.py -i datasets/<name> [... defaults k=(3,0,3,4,2), bnfloop 5]

Same loop (``main.py:86-149``): iteration = :class:`trainer.FusedTrainer.step`; every 10 iterations the output
mesh is evaluated (MAD vs ``*_gt.obj`` on the device), every 100 (``main.py``) / every 10 (``main4real.py``) an OBJ
is written to ``datasets/<mesh_name>/output/`` with the reference's file names.  ``--viewer/--port`` are accepted
for command-line compatibility; the viser web viewer is outside the hot path and is not started.
``--norm_optim`` is parsed and unused, as in the reference.

Meshes that exceed one GPU (round 5): the same two command lines under a launcher,

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 main.py -i datasets/<name> ...
    python main.py -i datasets/<name> --gpus N ...        (starts that launcher itself, before this process touches a GPU)

run one process per GPU: rank r takes ``cuda:LOCAL_RANK`` (``--gpu`` is the single-process flag), the mesh is face / vertex
partitioned with 1-hop halos (:mod:`dist`), the step is :class:`dist.DistributedTrainer`; every rank reads the same dataset
directory, rank 0 alone prints, evaluates and writes the OBJ files (the evaluation's all-gather is entered by every rank).
The reference has no counterpart (``main.py:51``: one device).
"""
from __future__ import annotations

import argparse
import os
import socket
import subprocess
import sys

import numpy as np
import torch


def get_parser(real: bool):
    p = argparse.ArgumentParser(description="Dual Deep Mesh Prior (MI355X HIP path)")
    p.add_argument("-i", "--input", type=str, required=True)
    p.add_argument("--pos_lr", type=float, default=0.01)
    p.add_argument("--norm_lr", type=float, default=0.01)
    if not real:
        p.add_argument("--norm_optim", type=str, default="Adam")
    p.add_argument("--iter", type=int, default=1000)
    p.add_argument("--k1", type=float, default=3.0)
    p.add_argument("--k2", type=float, default=0.0 if real else 4.0)
    p.add_argument("--k3", type=float, default=3.0 if real else 4.0)
    p.add_argument("--k4", type=float, default=4.0)
    p.add_argument("--k5", type=float, default=2.0 if real else 1.0)
    p.add_argument("--grad_crip", type=float, default=0.8)
    p.add_argument("--bnfloop", type=int, default=5 if real else 1)
    p.add_argument("--gpu", type=int, default=0)
    if not real:
        p.add_argument("--viewer", action="store_true", default=True)
        p.add_argument("--port", type=int, default=8080)
    p.add_argument("--no_graph", action="store_true", help="launch every kernel eagerly on one stream (default: one hipGraph per iteration, PosNet and NormalNet on two streams)")
    p.add_argument("--seed", type=int, default=None, help="torch seed for the weight init (the reference is unseeded)")
    p.add_argument("--gpus", type=int, default=1,
                   help="N > 1 outside a torchrun environment: start N ranks (one per GPU) through torch.distributed.run")
    p.add_argument("--dtype", choices=["fp32", "bf16"], default="fp32",
                   help="feature dtype in HBM: fp32 (default, float32-class arithmetic) or bf16 features "
                        "(bfloat16 activations / activation gradients, float32 parameters and accumulation)")
    return p


def _self_launch(argv, real, n):
    """``--gpus N`` without a launcher: become the parent of ``torch.distributed.run`` (this process has not initialised HIP;
    the children are started, never exec'd into)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "main4real.py" if real else "main.py")
    # the launcher picks its own free port on the loopback address (--standalone: a c10d rendezvous at localhost:0 -- no
    # bind-then-close race with another process; --local-addr: the container's hostname may not resolve)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(n), script] + list(sys.argv[1:] if argv is None else argv)
    return subprocess.run(cmd, env=env).returncode


def train_loop(tr, args, mesh_dic, real, rank=0, world=1, evaluator=None, out_dir=None, log=print):
    """The loop of ``main.py:86-149`` / ``main4real.py:52-87`` around ``tr.step()`` for one process or for rank ``rank`` of
    ``world``: with peers the positions are all-gathered by EVERY rank at the evaluation cadence (a collective), rank 0 alone
    logs, evaluates (``evaluator.mad(pos)``) and saves.  -> last MAD (rank 0) or None."""
    from . import loss as Loss
    from .mesh import Mesh
    gt_mesh, n_mesh, o1_mesh = mesh_dic["gt_mesh"], mesh_dic["n_mesh"], mesh_dic["o1_mesh"]
    get_pos = tr.gather_pos if world > 1 else (lambda: tr.pos)
    mad_value = None
    if gt_mesh is not None and not real:
        mad_value = Loss.mad(n_mesh.fn, gt_mesh.fn)
        if rank == 0:
            log("initial_mad: {:.3f}".format(mad_value))
    for epoch in range(1, args.iter + 1):
        loss = tr.step().item()
        if rank == 0 and (epoch % 10 == 0 or epoch == args.iter):
            log("Epoch {}: loss={:.6f}".format(epoch, loss) + ("" if mad_value is None else " mad={:.3f}".format(mad_value)))
        if epoch % 10 == 0:
            healed = tr.check_scales()                           # OverflowError (non-finite operands) is fatal: let it out
            if healed and rank == 0:
                log("[INFO] %d f16x3 GEMM operand(s) outgrew their scale and were recomputed with the measured one" % healed)
            save = real or epoch % 100 == 0
            if evaluator is not None or save:
                pos = get_pos()                                  # (with peers: every rank enters the all-gather)
                if rank == 0:
                    if evaluator is not None:
                        mad_value = evaluator.mad(pos)
                    if save and out_dir is not None:
                        o1_mesh.vs = pos.to("cpu").detach().numpy().copy()
                        name = "_ddmp.obj" if real or evaluator is None else "_ddmp={:.3f}.obj".format(mad_value)
                        Mesh.save(o1_mesh, out_dir + "/" + str(epoch) + name)
    if mad_value is not None and rank == 0:
        log("final_mad: {:.3f}".format(mad_value))
    return mad_value


def run(argv=None, real: bool = False):
    args = get_parser(real).parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1 and "RANK" not in os.environ:
        sys.exit(_self_launch(argv, real, args.gpus))            # (before anything touches a GPU)
    if args.gpus > 1 and world != args.gpus:
        sys.exit("--gpus %d but the launcher started %d ranks" % (args.gpus, world))
    rank, local_rank = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if rank == 0:
        for k, v in vars(args).items():
            print("{:12s}: {}".format(k, v))
    from . import datamaker
    from .evaluate import Evaluator
    from .networks import PosNet, NormalNet
    from .trainer import FusedTrainer

    mesh_dic, dataset = datamaker.create_dataset(args.input)
    mesh_name = mesh_dic["mesh_name"]
    gt_mesh, n_mesh, o1_mesh = mesh_dic["gt_mesh"], mesh_dic["n_mesh"], mesh_dic["o1_mesh"]
    if not torch.cuda.is_available():
        raise RuntimeError("the HIP path needs a GPU: there is no CPU fallback")
    device = torch.device("cuda:" + str(local_rank if world > 1 else args.gpu))
    if world > 1 and rank == 0 and (args.gpu != 0 or args.no_graph):
        print("[WARN]: --gpu / --no_graph have no effect with %d ranks: rank r runs on cuda:LOCAL_RANK, the partitioned trainer "
              "launches eagerly (dual-dmp_amd/dist.py)" % world)
    torch.cuda.set_device(device)
    seed = args.seed
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        box = [seed if seed is not None else int.from_bytes(os.urandom(4), "little")]
        dist.broadcast_object_list(box, src=0)                   # the reference is unseeded: every rank takes rank 0's draw
        seed = box[0]
    if seed is not None:
        torch.manual_seed(seed)
    fdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    posnet, normnet = PosNet(device, dtype=fdt).to(device), NormalNet(device, dtype=fdt).to(device)
    k = (args.k1, args.k2, args.k3, args.k4, args.k5)
    if world > 1:
        from .dist import make_distributed_trainer
        tr = make_distributed_trainer(n_mesh, o1_mesh, dataset, device, rank, world, bnfloop=args.bnfloop, nets=(posnet, normnet),
                                      pos_lr=args.pos_lr, norm_lr=args.norm_lr, k=k, grad_crip=args.grad_crip)
    else:
        dataset.to(device)
        tr = FusedTrainer(posnet, normnet, dataset, n_mesh, pos_lr=args.pos_lr, norm_lr=args.norm_lr, k=k,
                          grad_crip=args.grad_crip, bnfloop=args.bnfloop, use_graph=not args.no_graph, overlap=not args.no_graph)
    out_dir = "datasets/" + mesh_name + "/output"
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)
    ev = Evaluator(n_mesh, gt_mesh.fn, device) if (gt_mesh is not None and not real and rank == 0) else None
    # (ranks > 0 pass a stand-in so that they enter the evaluation's all-gather at the same epochs)
    evaluator = ev if rank == 0 else (object() if (gt_mesh is not None and not real) else None)
    train_loop(tr, args, mesh_dic, real, rank, world, evaluator, out_dir)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return tr


def main():
    run(real=False)


def main4real():
    run(real=True)
