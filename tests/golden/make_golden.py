#!/usr/bin/env python3
"""Generate golden vectors by RUNNING the reference's importable modules.

Runs only in the build container (needs /root/reference); the GPU box and the test
suite read the committed ``tests/golden/*.npz`` and never this script's imports.

What is captured:
  bnf_<name>.npz    ``loss.bnf`` (util/loss.py:195-259): filtered normals and the updated mesh, iter in {1, 3}
  mesh_<name>.npz   reference ``Mesh(path)`` arrays for small synthetic meshes
                    (util/mesh.py:8-21) + the OBJ text they were parsed from
                    + ``Mesh.save`` output text (util/mesh.py:267-285)
  loss_<name>.npz   values AND autograd gradients of the five training losses
                    (util/loss.py:16,37,55,86,140) for seeded inputs, bnf loop in {1,5},
                    plus mad / angular_difference (util/loss.py:261-277),
                    Mesh.compute_face_normals on a displaced mesh (util/mesh.py:87-92),
                    models.compute_fn (util/models.py:5-10) and models.vertex_updating (:31-44)

  noise_<name>.npz  the numpy halves of the two preprocess scripts, which need no MeshLab: preprocess/noisemaker.py:60-73
                    (``edge_based_scaling`` :32-36 + ``gausian_noise`` :38-42 between ``Mesh()`` / ``Mesh.save`` round trips)
                    and preprocess/preprocess.py:56-78 (rescale of the saved triplet) -- input OBJ texts and the OBJ texts
                    the reference's functions leave behind (``python tests/golden/make_golden.py noise`` writes only these)

``pymeshlab`` is stubbed at import (util/loss.py:4; only used at :279-284).
``util/networks.py`` / ``util/datamaker.py`` cannot be imported (torch_geometric absent):
the GCN stack has no golden vectors -> "parity unpinned" for it (see oracle/README.md).

    python tests/golden/make_golden.py
"""
import io
import os
import sys
import tempfile
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")

REF = "/root/reference"
sys.path.insert(0, REF)
sys.modules["pymeshlab"] = types.SimpleNamespace(MeshSet=object)
import util.loss as RefLoss          # noqa: E402
import util.models as RefModels      # noqa: E402
from util.mesh import Mesh as RefMesh  # noqa: E402

from dual_dmp_amd import synth       # noqa: E402  (generators only: inputs, not outputs)


def write_obj(path, vs, faces):
    with open(path, "w") as fp:
        for x, y, z in vs:
            fp.write("v {0:.8f} {1:.8f} {2:.8f}\n".format(x, y, z))
        for a, b, c in faces:
            fp.write("f {0} {1} {2}\n".format(a + 1, b + 1, c + 1))


def mesh_arrays(m):
    v2v = m.v2v_mat
    return dict(
        vs=m.vs, faces=m.faces, edges=m.edges, f2f=m.f2f, f_edges=m.f_edges,
        fn=m.fn, fa=m.fa, fc=m.fc, vn=m.vn, v_dims=m.v_dims.numpy(),
        v2v_indices=v2v._indices().numpy(), v2v_values=v2v._values().numpy(),
        vf_ptr=np.cumsum([0] + [len(s) for s in m.vf]),
        vf_idx=np.concatenate([np.sort(list(s)) for s in m.vf]),
        v2f_indices=m.v2f_mat._indices().numpy(),
    )


def grads(fn, *tensors):
    leaves = [t.clone().requires_grad_(True) for t in tensors]
    out = fn(*leaves)
    if isinstance(out, tuple):
        loss, extra = out
    else:
        loss, extra = out, None
    g = torch.autograd.grad(loss, leaves, allow_unused=True)
    g = [torch.zeros_like(l) if x is None else x for x, l in zip(g, leaves)]
    return loss.detach(), extra, g


def main():
    meshes = {
        "ico2": synth.icosphere(2),
        "grid4": synth.open_grid(4, 4),
        "cube3": synth.cube_cad(3),
        "grid7x5": synth.open_grid(7, 5),
    }
    tmp = tempfile.mkdtemp()
    for name, (vs, faces) in meshes.items():
        # unit mean edge + noise so the geometry looks like a training input
        gt, noisy, _ = synth.make_triplet(vs, faces)
        path = os.path.join(tmp, name + ".obj")
        write_obj(path, noisy.vs, faces)
        obj_text = open(path).read()
        m = RefMesh(path)
        arrs = mesh_arrays(m)
        spath = os.path.join(tmp, name + "_saved.obj")
        m.save(spath)
        np.savez_compressed(os.path.join(HERE, "mesh_%s.npz" % name),
                            obj_text=np.frombuffer(obj_text.encode(), dtype=np.uint8),
                            save_text=np.frombuffer(open(spath).read().encode(), dtype=np.uint8),
                            **arrs)

        # ---------------- losses
        rng = np.random.default_rng(2718)
        V, F = len(m.vs), len(m.faces)
        pos = torch.tensor(m.vs + 0.05 * rng.standard_normal((V, 3)), dtype=torch.float32)
        nrm = m.fn + 0.3 * rng.standard_normal((F, 3))
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        nrm = torch.tensor(nrm, dtype=torch.float32)
        out = dict(pos=pos.numpy(), norm=nrm.numpy(), real_pos=m.vs, real_norm=m.fn)

        l, _, g = grads(lambda p: RefLoss.pos_rec_loss(p, m.vs), pos)
        out.update(pos_rec=l.numpy(), pos_rec_dpos=g[0].numpy())
        l, _, g = grads(lambda p: RefLoss.mesh_laplacian_loss(p, m), pos)
        out.update(lap=l.numpy(), lap_dpos=g[0].numpy())
        l, _, g = grads(lambda n: RefLoss.norm_rec_loss(n, m.fn), nrm)
        out.update(norm_rec=l.numpy(), norm_rec_dnorm=g[0].numpy())
        for loop in (1, 5):
            l, new_fn, g = grads(lambda p, n: RefLoss.fn_bnf_loss(p, n, m, loop=loop), pos, nrm)
            out.update({"bnf%d" % loop: l.numpy(), "bnf%d_newfn" % loop: new_fn.detach().numpy(),
                        "bnf%d_dnorm" % loop: g[1].numpy(), "bnf%d_dpos" % loop: g[0].numpy()})
        l, _, g = grads(lambda p, n: RefLoss.pos_norm_loss(p, n, m), pos, nrm)
        out.update(pos_norm=l.numpy(), pos_norm_dpos=g[0].numpy(), pos_norm_dnorm=g[1].numpy())

        # the weighted sum exactly as main.py:106 forms it (k = 3,4,4,4,1), f64 by promotion
        def total(p, n):
            a = RefLoss.pos_rec_loss(p, m.vs)
            b = RefLoss.mesh_laplacian_loss(p, m)
            c = RefLoss.norm_rec_loss(n, m.fn)
            d, _ = RefLoss.fn_bnf_loss(p, n, m, loop=1)
            e = RefLoss.pos_norm_loss(p, n, m)
            return 3.0 * a + 4.0 * b + 4.0 * c + 4.0 * d + 1.0 * e
        l, _, g = grads(total, pos, nrm)
        out.update(total=l.numpy(), total_dpos=g[0].numpy(), total_dnorm=g[1].numpy())

        # ---------------- metric
        out["mad"] = np.float64(RefLoss.mad(nrm, m.fn))
        out["angdiff"] = RefLoss.angular_difference(nrm.numpy().astype(np.float64), m.fn)
        m2 = RefMesh(path)
        m2.vs = pos.numpy().astype(np.float64)
        RefMesh.compute_face_normals(m2)
        out.update(cfn_fn=m2.fn, cfn_fa=m2.fa, mad_pos=np.float64(RefLoss.mad(m2.fn, m.fn)))
        out["models_compute_fn"] = RefModels.compute_fn(pos, m.faces).numpy()
        # post-process of the paper's pipeline (util/models.py:31-44; dead code in main.py:115,139): move vertices so
        # that the faces agree with the predicted normals
        for loop in (1, 3):
            out["vertex_updating_%d" % loop] = RefModels.vertex_updating(pos, nrm, m, loop=loop).numpy()
        np.savez_compressed(os.path.join(HERE, "loss_%s.npz" % name), **out)

        # ---------------- the non-default ltype variants (util/loss.py:22,43,62-77,119-130,153): dead in both drivers, part
        # of the functions' signatures -> ltype_<name>.npz (same inputs as above)
        lv = {}
        l, _, g = grads(lambda p: RefLoss.pos_rec_loss(p, m.vs, ltype="l1mae"), pos)
        lv.update(pos_rec_l1mae=l.numpy(), pos_rec_l1mae_dpos=g[0].numpy())
        l, _, g = grads(lambda p: RefLoss.mesh_laplacian_loss(p, m, ltype="mae"), pos)
        lv.update(lap_mae=l.numpy(), lap_mae_dpos=g[0].numpy())
        for lt in ("l2mae", "l2rmse", "l1rmse", "cos"):
            l, _, g = grads(lambda n: RefLoss.norm_rec_loss(n, m.fn, ltype=lt), nrm)
            lv.update({"norm_rec_%s" % lt: l.numpy(), "norm_rec_%s_dnorm" % lt: g[0].numpy()})
        for lt in ("mae", "rmse", "l1rmse"):
            for loop in (1, 5):
                l, new_fn, g = grads(lambda p, n: RefLoss.fn_bnf_loss(p, n, m, ltype=lt, loop=loop), pos, nrm)
                lv.update({"bnf%d_%s" % (loop, lt): l.numpy(), "bnf%d_%s_dnorm" % (loop, lt): g[1].numpy()})
        l, _, g = grads(lambda p, n: RefLoss.pos_norm_loss(p, n, m, ltype="rmse"), pos, nrm)
        lv.update(pos_norm_rmse=l.numpy(), pos_norm_rmse_dpos=g[0].numpy(), pos_norm_rmse_dnorm=g[1].numpy())
        np.savez_compressed(os.path.join(HERE, "ltype_%s.npz" % name), **lv)

        # ---------------- classical bilateral normal filter + vertex update (util/loss.py:195-259; dead in both drivers) ->
        # bnf_<name>.npz: float64 input normals, iter in {1, 3}; and the float32-tensor input the signature also accepts
        bv = {}
        for it in (1, 3):
            nf, nm = RefLoss.bnf(nrm.numpy().astype(np.float64), m, iter=it)
            bv.update({"newfn_%d" % it: nf, "vs_%d" % it: nm.vs, "fc_%d" % it: nm.fc, "fa_%d" % it: nm.fa, "fn_%d" % it: nm.fn})
        nf, nm = RefLoss.bnf(nrm, m, sigma_s=0.5, sigma_c=0.3, iter=2)
        bv.update(newfn_f32in=nf, vs_f32in=nm.vs)
        np.savez_compressed(os.path.join(HERE, "bnf_%s.npz" % name), **bv)
        print(name, "V", V, "F", F, "pos_rec", out["pos_rec"], out["pos_rec"].dtype,
              "lap", out["lap"], "norm_rec", out["norm_rec"], "bnf1", out["bnf1"],
              "bnf5", out["bnf5"], "pos_norm", out["pos_norm"], "mad", out["mad"])


def _import_ref_script(name):
    """preprocess/<name>.py of the reference as a module (its ``import pymeshlab`` hits the stub above; nothing of it is
    called: the MeshLab filters are the part that cannot run here)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, "preprocess", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def noise_golden():
    nm = _import_ref_script("noisemaker")
    meshes = {
        "ico2": synth.icosphere(2),
        "grid7x5": synth.open_grid(7, 5),
        "cube3": synth.cube_cad(3),
    }
    tmp = tempfile.mkdtemp()
    rd = lambda p: np.frombuffer(open(p).read().encode(), dtype=np.uint8)
    for k, (name, (vs, faces)) in enumerate(meshes.items()):
        out = {}
        level = (0.2, 0.35, 0.1)[k]
        # -------- noisemaker.py:60-73.  Input: what MeshLab's normalize + save would leave (any mesh in a unit box will do)
        lo, hi = vs.min(0), vs.max(0)
        pre = (vs - 0.5 * (lo + hi)) / float((hi - lo).max()) * (1.0, 0.731, 2.4)[k]
        g_file, n_file = os.path.join(tmp, name + "_gt.obj"), os.path.join(tmp, name + "_noise.obj")
        write_obj(g_file, pre, faces)
        out["pre_text"] = rd(g_file)
        g_mesh = RefMesh(g_file)
        g_mesh = nm.edge_based_scaling(g_mesh)
        g_mesh.compute_face_normals()
        g_mesh.save(g_file)
        n_mesh = RefMesh(g_file)
        n_mesh = nm.gausian_noise(n_mesh, level)
        n_mesh.compute_face_normals()
        n_mesh.save(n_file)
        out.update(level=np.float64(level), gt_text=rd(g_file), noise_text=rd(n_file), noise_vs=n_mesh.vs, gt_vs=g_mesh.vs,
                   mad=np.float64(RefLoss.mad(n_mesh.fn, g_mesh.fn)))
        # -------- preprocess.py:56-78 (its body past the MeshLab calls, statement by statement).  Input: three saved layers
        rng = np.random.default_rng(99 + k)
        s_file = os.path.join(tmp, name + "_smooth.obj")
        write_obj(n_file, pre * 3.7 + 0.02 * rng.standard_normal(pre.shape), faces)
        write_obj(s_file, pre * 3.7, faces)
        write_obj(g_file, pre * 3.7 + 0.001, faces)
        out.update(p_noise_in=rd(n_file), p_smooth_in=rd(s_file), p_gt_in=rd(g_file))
        n2, s2, g2 = RefMesh(n_file), RefMesh(s_file), RefMesh(g_file)
        edge_vec = n2.vs[n2.edges][:, 0, :] - n2.vs[n2.edges][:, 1, :]
        ave_len = np.sum(np.linalg.norm(edge_vec, axis=1)) / n2.edges.shape[0]
        n2.vs /= ave_len
        s2.vs /= ave_len
        n2.save(n_file)
        s2.save(s_file)
        g2.vs /= ave_len
        g2.save(g_file)
        out.update(p_noise_out=rd(n_file), p_smooth_out=rd(s_file), p_gt_out=rd(g_file))
        np.savez_compressed(os.path.join(HERE, "noise_%s.npz" % name), **out)
        print("noise", name, "level", level, "mad", out["mad"])


if __name__ == "__main__":
    if sys.argv[1:] == ["noise"]:
        noise_golden()
    else:
        main()
        noise_golden()
