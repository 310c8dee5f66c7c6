"""Dataset preparation without MeshLab: the two command lines of the reference's ``preprocess/``.

    python -m dual_dmp_amd.preprocess -i datasets/<name>/<clean>.obj [--noise gaussian] [--level 0.2] [--step 30]
        = preprocess/noisemaker.py:12-80: one clean mesh -> <name>_gt.obj, <name>_noise.obj, <name>_smooth.obj in the
          directory of the input (``<name>`` = that directory's name, noisemaker.py:47-52); the input file is moved to
          ``original/`` (noisemaker.py:55-56).  The directory is then what ``main.py -i datasets/<name>`` reads
          (util/datamaker.py:26-35).
    python -m dual_dmp_amd.preprocess -i datasets/<name> [--step 30]
        = preprocess/preprocess.py:12-79: a directory that holds ``*_noise.obj`` (and optionally ``*_gt.obj``), e.g. a
          real scan -> writes ``<name>_smooth.obj`` and rescales all of them to unit mean edge length of the noisy mesh.

Same flags, same file names, same order of operations, same seed (314).  The scripts' numpy halves -- everything between
the MeshLab calls: :func:`rescale_and_noise` (noisemaker.py:60-73) and :func:`rescale_saved` (preprocess.py:56-78) -- are PINNED:
they reproduce byte for byte the OBJ files the reference's own functions write (tests/golden/noise_*.npz,
tests/test_mesh.py::test_preprocess_numpy_halves_match_reference_golden).  What differs, by necessity: the reference calls
pymeshlab for three steps; here they are numpy, with these definitions --

* normalize (``transform_scale_normalize`` unit box + ``transform_translate_center_set_origin`` on the bbox centre,
  noisemaker.py:28-30): divide by the longest bounding-box side, move the bounding-box centre to the origin.  The
  edge-based rescale that follows makes the scale factor immaterial; only the centring survives.
* ``laplacian_smooth(stepsmoothnum=step, cotangentweight=False)`` (noisemaker.py:25-26): MeshLab's source is not in the
  reference tree and pymeshlab (==2021.10, requirements.txt:6) cannot be installed here; :func:`synth.laplacian_smooth` restates
  the published algorithm of that filter -- vcglib's ``Smooth::VertexCoordLaplacian``: interior vertices
  p <- (p + 2 sum_nbr p_j) / (2 deg + 1), border vertices averaged with their border neighbours only,
  p <- (2 p + p_a + p_b) / 4 -- in float64 (MeshLab: float32).  No MeshLab output exists to pin it against: said so in DESIGN.md.
"""
from __future__ import annotations

import argparse
import glob
import os
import shutil

import numpy as np

from . import synth
from .mesh import Mesh


def get_parser(argv=None):
    parser = argparse.ArgumentParser(description="create datasets (noisy mesh & smoothed mesh) from a single clean mesh, "
                                                 "or smooth + rescale a directory that holds *_noise.obj")
    parser.add_argument("-i", "--input", type=str, required=True)
    parser.add_argument("--noise", type=str, default="gaussian")
    parser.add_argument("--level", type=float, default=0.2)
    parser.add_argument("--step", type=int, default=30)
    args = parser.parse_args(argv)
    for k, v in vars(args).items():
        print("{:12s}: {}".format(k, v))
    return args


def normalize(vs: np.ndarray) -> np.ndarray:
    lo, hi = vs.min(0), vs.max(0)
    side = float((hi - lo).max())
    return (vs - 0.5 * (lo + hi)) / (side if side > 0 else 1.0)


def smooth_mesh(mesh: Mesh, step: int) -> Mesh:
    return Mesh(vs=synth.laplacian_smooth(mesh.vs, mesh.vv_ptr, mesh.vv_idx, steps=step, faces=mesh.faces), faces=mesh.faces)


def rescale_and_noise(g_file: str, n_file: str, level: float = 0.2):
    """The numpy half of noisemaker.py (:60-73), which needs no MeshLab and is PINNED to the reference's own output
    (tests/golden/noise_*.npz): read the pre-saved ground truth back, rescale it to unit mean edge length
    (``edge_based_scaling`` :32-36) and save it; read THAT file back (coordinates at the OBJ writer's precision), displace
    every vertex along its normal by N(0, level) drawn with seed 314 (``gausian_noise`` :38-42) and save.
    Returns (gt, noisy)."""
    g_mesh = Mesh(g_file)
    g_mesh = Mesh(vs=g_mesh.vs / synth.mean_edge_length(g_mesh.vs, g_mesh.edges), faces=g_mesh.faces)    # re-scaling
    g_mesh.save(g_file)
    base = Mesh(g_file)
    n_mesh = Mesh(vs=synth.gaussian_noise(base.vs, base.vn, level=level), faces=base.faces)
    n_mesh.save(n_file)
    return g_mesh, n_mesh


def rescale_saved(n_file: str, s_file: str, g_file=None):
    """The numpy half of preprocess.py (:56-78), pinned like :func:`rescale_and_noise`: read the normalised files back, divide
    all of them by the mean edge length of the NOISY mesh, save.  Returns (gt or None, noisy, smooth)."""
    n_mesh, s_mesh = Mesh(n_file), Mesh(s_file)
    g_mesh = Mesh(g_file) if g_file is not None and os.path.exists(g_file) else None
    ave_len = synth.mean_edge_length(n_mesh.vs, n_mesh.edges)
    out = []
    for m, f in ((g_mesh, g_file), (n_mesh, n_file), (s_mesh, s_file)):
        if m is None:
            out.append(None)
            continue
        mm = Mesh(vs=m.vs / ave_len, faces=m.faces)
        mm.save(f)
        out.append(mm)
    return out[0], out[1], out[2]


def from_clean_obj(path: str, level: float = 0.2, step: int = 30, move_original: bool = True):
    """noisemaker.py:44-80.  Returns (gt, noisy, smooth, directory)."""
    root_dir = os.path.dirname(os.path.abspath(path))
    mesh_name = os.path.basename(root_dir)
    n_file = os.path.join(root_dir, mesh_name + "_noise.obj")
    s_file = os.path.join(root_dir, mesh_name + "_smooth.obj")
    g_file = os.path.join(root_dir, mesh_name + "_gt.obj")
    src = Mesh(path)
    if move_original:
        os.makedirs(os.path.join(root_dir, "original"), exist_ok=True)
        shutil.move(path, os.path.join(root_dir, "original", os.path.basename(path)))
    Mesh(vs=normalize(src.vs), faces=src.faces).save(g_file)                   # pre-scaling & transformation, pre-saving
    g_mesh, n_mesh = rescale_and_noise(g_file, n_file, level)
    s_mesh = smooth_mesh(Mesh(n_file), step)
    s_mesh.save(s_file)
    return g_mesh, n_mesh, s_mesh, root_dir


def from_noisy_dir(directory: str, step: int = 30):
    """preprocess.py:42-79.  Returns (gt or None, noisy, smooth, directory)."""
    directory = os.path.abspath(directory)
    found = glob.glob(os.path.join(directory, "*_noise.obj"))
    if not found:
        raise FileNotFoundError("no *_noise.obj in %s" % directory)
    n_file = found[0]
    mesh_name = os.path.basename(os.path.dirname(n_file))
    s_file = os.path.join(directory, mesh_name + "_smooth.obj")
    g_file = os.path.join(directory, mesh_name + "_gt.obj")
    n_mesh = Mesh(n_file)
    s_mesh = smooth_mesh(n_mesh, step)
    g_mesh = Mesh(g_file) if os.path.exists(g_file) else None
    # MeshLab normalises the layers together (alllayers=True): one box over all of them
    allv = np.concatenate([m.vs for m in (s_mesh, g_mesh, n_mesh) if m is not None])
    lo, hi = allv.min(0), allv.max(0)
    side = float((hi - lo).max()) or 1.0
    centre = 0.5 * (lo + hi)
    # MeshLab saves the normalised layers (preprocess.py:29-40); the script then reads them back and rescales (:56-78)
    for m, f in ((s_mesh, s_file), (g_mesh, g_file), (n_mesh, n_file)):
        if m is not None:
            Mesh(vs=(m.vs - centre) / side, faces=m.faces).save(f)
    out = rescale_saved(n_file, s_file, g_file)
    return out[0], out[1], out[2], directory


def main(argv=None):
    args = get_parser(argv)
    if os.path.isdir(args.input):
        g_mesh, n_mesh, s_mesh, d = from_noisy_dir(args.input, step=args.step)
    else:
        if args.noise != "gaussian":
            print("[WARN]: noise type %r: the reference applies gaussian noise for every type (noisemaker.py:68-75)" % args.noise)
        g_mesh, n_mesh, s_mesh, d = from_clean_obj(args.input, level=args.level, step=args.step)
    if g_mesh is not None:
        from .loss import mad
        print("[Finished] Vertices: {}, faces: {}, mad: {:.4f}".format(n_mesh.vs.shape[0], n_mesh.faces.shape[0],
                                                                         mad(n_mesh.fn, g_mesh.fn)))
    else:
        print("[Finished] Vertices: {}, faces: {}".format(n_mesh.vs.shape[0], n_mesh.faces.shape[0]))
    return d


if __name__ == "__main__":
    main()
