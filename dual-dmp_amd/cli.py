"""Command lines with the flags and defaults of the reference's ``main.py:15-37`` and ``main4real.py:12-32``.

    python main.py -i datasets/<name> [--pos_lr .01 --norm_lr .01 --iter 1000 --k1 3 --k2 4 --k3 4 --k4 4 --k5 1
                                       --grad_crip .8 --bnfloop 1 --gpu 0]
    python main4real.py -i datasets/<name> [... defaults k=(3,0,3,4,2), bnfloop 5]

Same loop (``main.py:86-149``): iteration = :class:`trainer.FusedTrainer.step`; every 10 iterations the output
mesh is evaluated (MAD vs ``*_gt.obj`` on the device), every 100 (``main.py``) / every 10 (``main4real.py``) an OBJ
is written to ``datasets/<mesh_name>/output/`` with the reference's file names.  ``--viewer/--port`` are accepted
for command-line compatibility; the viser web viewer is outside the hot path and is not started.
``--norm_optim`` is parsed and unused, as in the reference.
"""
from __future__ import annotations

import argparse
import os

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
    p.add_argument("--dtype", choices=["fp32", "bf16"], default="fp32",
                   help="feature dtype in HBM: fp32 (default, float32-class arithmetic) or bf16 features "
                        "(bfloat16 activations / activation gradients, float32 parameters and accumulation)")
    return p


def run(argv=None, real: bool = False):
    args = get_parser(real).parse_args(argv)
    for k, v in vars(args).items():
        print("{:12s}: {}".format(k, v))
    from . import datamaker, loss as Loss
    from .evaluate import Evaluator
    from .mesh import Mesh
    from .networks import PosNet, NormalNet
    from .trainer import FusedTrainer

    mesh_dic, dataset = datamaker.create_dataset(args.input)
    mesh_name = mesh_dic["mesh_name"]
    gt_mesh, n_mesh, o1_mesh = mesh_dic["gt_mesh"], mesh_dic["n_mesh"], mesh_dic["o1_mesh"]
    if not torch.cuda.is_available():
        raise RuntimeError("the HIP path needs a GPU: there is no CPU fallback")
    device = torch.device("cuda:" + str(args.gpu))
    torch.cuda.set_device(device)
    if args.seed is not None:
        torch.manual_seed(args.seed)
    fdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    posnet, normnet = PosNet(device, dtype=fdt).to(device), NormalNet(device, dtype=fdt).to(device)
    dataset.to(device)
    tr = FusedTrainer(posnet, normnet, dataset, n_mesh, pos_lr=args.pos_lr, norm_lr=args.norm_lr,
                      k=(args.k1, args.k2, args.k3, args.k4, args.k5), grad_crip=args.grad_crip, bnfloop=args.bnfloop,
                      use_graph=not args.no_graph, overlap=not args.no_graph)
    out_dir = "datasets/" + mesh_name + "/output"
    os.makedirs(out_dir, exist_ok=True)
    ev = None
    mad_value = None
    if gt_mesh is not None and not real:
        init_mad = mad_value = Loss.mad(n_mesh.fn, gt_mesh.fn)
        print("initial_mad: {:.3f}".format(init_mad))
        ev = Evaluator(n_mesh, gt_mesh.fn, device)
    for epoch in range(1, args.iter + 1):
        loss = tr.step().item()
        if epoch % 10 == 0 or epoch == args.iter:
            print("Epoch {}: loss={:.6f}".format(epoch, loss) + ("" if mad_value is None else " mad={:.3f}".format(mad_value)))
        if epoch % 10 == 0:
            healed = tr.check_scales()                           # OverflowError (non-finite operands) is fatal: let it out
            if healed:
                print("[INFO] %d f16x3 GEMM operand(s) outgrew their scale and were recomputed with the measured one" % healed)
            if ev is not None:
                mad_value = ev.mad(tr.pos)
            if real or epoch % 100 == 0:
                o1_mesh.vs = tr.pos.to("cpu").detach().numpy().copy()
                name = "_ddmp.obj" if real or ev is None else "_ddmp={:.3f}.obj".format(mad_value)
                Mesh.save(o1_mesh, out_dir + "/" + str(epoch) + name)
    if mad_value is not None:
        print("final_mad: {:.3f}".format(mad_value))
    return tr


def main():
    run(real=False)


def main4real():
    run(real=True)
