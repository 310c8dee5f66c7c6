"""Dataset assembly (host-side mirror of ``util/datamaker.py`` of the reference).

``create_dataset(file_path)`` reads ``*_noise.obj``, ``*_smooth.obj`` and the optional ``*_gt.obj`` from a
directory (``util/datamaker.py:23-40``) and returns ``(mesh_dic, dataset)`` with the same keys / fields:

    z1         [V,16] f32   np.random.seed(314); normal(size=(V,16))            (:43-49 "rand16")
    z2         [F,7]  f32   [fc, fn, fa] of the noisy mesh                        (:63,80-81 "pos_norm_area")
    x_pos      [V,3]  f32   smoothed vertex positions                             (:87)
    x_norm     [F,3]  f32   noisy face normals                                    (:88)
    edge_index [2,2E] i64   edges.T followed by the reversed pairs                (:90-91)
    face_index [2,S]  i64   f_edges                                               (:92)

The reference wraps these in a ``torch_geometric.data.Data`` only to read a few bookkeeping attributes
back (``:10-15``); those are computed directly here (no PyG dependency).
"""
from __future__ import annotations

import glob
from typing import Tuple

import numpy as np
import torch

from .mesh import Mesh


class Dataset:
    def __init__(self, z1, z2, x_pos, x_norm, edge_index, face_index):
        self.keys = ["x", "z1", "z2", "x_pos", "x_norm", "edge_index", "face_index"]
        self.z1, self.z2 = z1, z2
        self.x_pos, self.x_norm = x_pos, x_norm
        self.edge_index, self.face_index = edge_index, face_index
        self.num_nodes = int(z1.shape[0])
        self.num_edges = int(edge_index.shape[1])
        self.num_node_features = int(z1.shape[1])
        deg = torch.bincount(edge_index.reshape(-1), minlength=self.num_nodes)
        self.contains_isolated_nodes = bool((deg == 0).any())
        self.contains_self_loops = bool((edge_index[0] == edge_index[1]).any())

    def to(self, device):
        """Move every tensor once (the reference re-uploads them on every forward, util/networks.py:49,110)."""
        for k in ("z1", "z2", "x_pos", "x_norm", "edge_index", "face_index"):
            setattr(self, k, getattr(self, k).detach().to(device))
        return self


def dataset_from_meshes(n_mesh: Mesh, s_mesh: Mesh) -> Dataset:
    np.random.seed(314)
    z1 = np.random.normal(size=(n_mesh.vs.shape[0], 16))
    z2 = np.concatenate([n_mesh.fc, n_mesh.fn, n_mesh.fa.reshape(-1, 1)], axis=1)
    z1 = torch.tensor(z1, dtype=torch.float)
    z2 = torch.tensor(z2, dtype=torch.float)
    x_pos = torch.tensor(s_mesh.vs, dtype=torch.float)
    x_norm = torch.tensor(n_mesh.fn, dtype=torch.float)
    edge_index = torch.tensor(n_mesh.edges.T, dtype=torch.long)
    edge_index = torch.cat([edge_index, edge_index[[1, 0], :]], dim=1)
    face_index = torch.from_numpy(np.ascontiguousarray(n_mesh.f_edges))
    return Dataset(z1, z2, x_pos, x_norm, edge_index, face_index)


def create_dataset(file_path: str) -> Tuple[dict, Dataset]:
    n_file = glob.glob(file_path + "/*_noise.obj")[0]
    s_file = glob.glob(file_path + "/*_smooth.obj")[0]
    mesh_name = n_file.split("/")[-2]
    gt_file = glob.glob(file_path + "/*_gt.obj")
    if len(gt_file) != 0:
        gt_file = gt_file[0]
        gt_mesh = Mesh(gt_file)
    else:
        gt_mesh = None
    n_mesh = Mesh(n_file)
    o1_mesh = Mesh(n_file)
    s_mesh = Mesh(s_file)
    dataset = dataset_from_meshes(n_mesh, s_mesh)
    mesh_dic = {"gt_file": gt_file, "n_file": n_file, "s_file": s_file, "mesh_name": mesh_name,
                "gt_mesh": gt_mesh, "n_mesh": n_mesh, "o1_mesh": o1_mesh, "s_mesh": s_mesh}
    return mesh_dic, dataset
