"""Evaluation block of the training loop on the device (``main.py:117-127`` of the reference).

Every 10 iterations the reference copies ``pos`` to the host, recomputes face AND vertex normals in numpy
(with a Python list comprehension over faces, ``util/mesh.py:102``) and evaluates MAD; at 1M faces that stalls
the loop for seconds.  Here face normals and the MAD reduction are two small kernels; only the scalar comes
back.  ``Evaluator.mad(pos)`` == ``Loss.mad(o1_mesh.fn, gt_mesh.fn)`` after ``o1_mesh.vs = pos`` (float32).
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import check
from .loss import tables_for, _target
from .ops import Workspace, _p, _stream


class Evaluator:
    def __init__(self, mesh, gt_fn, device):
        self.tb = tables_for(mesh, device)
        self.gt = _target(gt_fn, device)
        self.fn = torch.empty((self.tb.F, 3), dtype=torch.float32, device=device)
        self.out = torch.zeros(1, dtype=torch.float64, device=device)

    def face_normals(self, pos: torch.Tensor) -> torch.Tensor:
        pos = pos.detach().to(torch.float32).contiguous()
        check(_lib.lib().ddmp_face_normals_f32(self.tb.F, _p(pos), _p(self.tb.faces), _p(self.fn), None, _stream()),
              "ddmp_face_normals_f32")
        return self.fn

    def mad(self, pos: torch.Tensor) -> float:
        self.face_normals(pos)
        L = _lib.lib()
        ws = Workspace.get(L.ddmp_mad_workspace_bytes(), pos.device)
        check(L.ddmp_mad_f64(self.tb.F, _p(self.fn), _p(self.gt), _p(self.out), _p(ws), ws.numel(), _stream()),
              "ddmp_mad_f64")
        return float(self.out.item())
