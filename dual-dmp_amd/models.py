"""Tensor geometry helpers on the device (host-side mirror of ``util/models.py`` of the reference).

    compute_fn(vs, faces)                         util/models.py:5-10
    vertex_updating(pos, norm, mesh, loop=10)     util/models.py:31-44

``vertex_updating`` is the post-process of the paper's pipeline (move the vertices so that the faces agree with
the predicted normals).  In the reference it is an O(V * loop) Python loop that is only reachable through
``vs_update = False`` (``main.py:115,139``), i.e. dead; here it is two small kernels per sweep.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from ._lib import check
from .loss import tables_for
from .ops import _p, _stream


def compute_fn(vs: torch.Tensor, faces, mesh=None) -> torch.Tensor:
    """unit face normals of ``vs`` [V,3] (float32 on the device).  ``faces``: int array [F,3] (numpy or tensor)."""
    dev = vs.device
    if isinstance(faces, np.ndarray):
        faces = torch.from_numpy(np.ascontiguousarray(faces, dtype=np.int32))
    faces = faces.to(device=dev, dtype=torch.int32).contiguous()
    F = faces.shape[0]
    out = torch.empty((F, 3), dtype=torch.float32, device=dev)
    pos = vs.detach().to(torch.float32).contiguous()
    check(_lib.lib().ddmp_face_normals_f32(F, _p(pos), _p(faces), _p(out), None, _stream()), "ddmp_face_normals_f32")
    return out


def vertex_updating(pos: torch.Tensor, norm: torch.Tensor, mesh, loop=10) -> torch.Tensor:
    tb = tables_for(mesh, pos.device)
    new_pos = pos.detach().to(torch.float32).clone().contiguous()
    nrm = norm.detach().to(torch.float32).contiguous()
    fc = torch.empty((tb.F, 3), dtype=torch.float32, device=pos.device)
    check(_lib.lib().ddmp_vertex_update_f32(tb.V, tb.F, _p(new_pos), _p(nrm), _p(tb.faces), _p(tb.vf_ptr),
                                            _p(tb.vf_corner), _p(fc), int(loop), _stream()), "ddmp_vertex_update_f32")
    return new_pos
