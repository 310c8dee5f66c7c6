"""Host-side pieces that need no GPU: C ABI surface, CLI flags, dataset assembly."""
import ctypes
import os

import numpy as np
import pytest
import torch


def test_c_abi_exports_every_declared_symbol():
    from dual_dmp_amd import _lib
    protos = _lib.parse_header()
    assert len(protos) >= 40
    lib = _lib.lib()                       # resolves every prototype or raises
    for name in protos:
        assert hasattr(lib, name), name
    assert lib.ddmp_abi_version() == 3
    assert lib.ddmp_build_ablation_flags() == 0            # the in-tree library is never a timing-only ablation build
    assert _lib.status_string(0) == "ok" and _lib.status_string(-4) == "workspace too small"
    # argument validation happens before any device work
    assert lib.ddmp_spmm_f32(None, None, 0, None, 0, 0, None, None, None, 0.01, None) == -1
    assert lib.ddmp_gemm_nt_f32(None, 0, None, 0, None, 0, 0, 0, 0, None, None, None, 0.01, None, 0, None) == -1
    assert lib.ddmp_gemm_tn_workspace_bytes(1000000, 512, 512) > 0
    # round 3 entry points: argument checks need no device either
    assert lib.ddmp_gemm_prepare_weights(25, None, None, None, None, None, None, None, None, 1000, None, None) == -1   # > 24 matrices
    assert lib.ddmp_spmm_stats_supported(256) == 1 and lib.ddmp_spmm_stats_supported(40) == 0


def test_per_call_options_reach_their_own_call_only():
    """ABI 3: the *_o entry points take what ABI 2 armed "for the next call" as an explicit ddmp_opts argument (the arming calls
    themselves are gone from the ABI since round 6).  The options apply to that call alone -- whatever it returns -- and a
    malformed block is an argument error.  No device work involved (every call fails its argument checks)."""
    from dual_dmp_amd import _lib, ops
    lib = _lib.lib()
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.ddmp_next_pending() == 0
    for gone in ("ddmp_bn_next_prepare", "ddmp_bn_next_bwd_prepare", "ddmp_gemm_next_scales", "ddmp_gemm_next_prepared", "ddmp_next_cancel"):
        assert not hasattr(lib, gone), gone                      # (hidden visibility: not exported any more)

    class T:                                                     # a stand-in with .data_ptr() / .numel()
        def data_ptr(self):
            return p.value

        def numel(self):
            return 32
    t = T()
    o, keep = ops._mk_opts(ops.BnFwd(100.0, t, t, [t, t, t, t]), (t, None, True), True)
    assert keep[0].flags == ops.OPT_BN_FWD | ops.OPT_SCALES | ops.OPT_PREPARED and keep[0].struct_size == ctypes.sizeof(ops._Opts)
    # the wrapped call fails its argument checks: the options are gone afterwards
    assert lib.ddmp_gemm_nt_stats_f32_o(None, 0, None, 0, None, 0, 0, 0, 0, None, None, None, 0.01, None, None, 0, None, 0, None, o) == -1
    assert lib.ddmp_next_pending() == 0
    assert lib.ddmp_gemm_nn_o(None, 0, None, 0, None, 0, 0, 0, 0, 0, None, 0, None, o) == -1
    assert lib.ddmp_next_pending() == 0
    # a block of another size / with unknown flags / with both BatchNorm directions: argument error before the call
    keep[0].struct_size = 8
    assert lib.ddmp_bn_stats_o(None, 0, 0, 32, 0, None, None, 0, None, o) == -1
    keep[0].struct_size = ctypes.sizeof(ops._Opts)
    keep[0].flags = 64
    assert lib.ddmp_bn_stats_o(None, 0, 0, 32, 0, None, None, 0, None, o) == -1
    keep[0].flags = ops.OPT_BN_FWD | ops.OPT_BN_BWD
    assert lib.ddmp_bn_stats_o(None, 0, 0, 32, 0, None, None, 0, None, o) == -1
    assert lib.ddmp_next_pending() == 0


def test_host_csr_matches_gcn_norm(oracle):
    from dual_dmp_amd import ops, synth
    from dual_dmp_amd.mesh import Mesh
    v, f = synth.open_grid(6, 5)
    m = Mesh(vs=v, faces=f)
    e = torch.tensor(m.edges.T, dtype=torch.long)
    ei = torch.cat([e, e[[1, 0]], torch.tensor([[0, 3], [0, 3]])], 1)          # + two explicit self loops
    n = len(v)
    rowptr, col, dinv = ops.csr_build_host(ei.numpy(), n)
    keep = ei[:, ei[0] != ei[1]]
    row, colo, w = oracle.gcn_norm(keep, n, torch.float64)
    deg = torch.zeros(n, dtype=torch.float64).scatter_add_(0, colo, torch.ones_like(w))
    np.testing.assert_allclose(dinv, deg.pow(-0.5).numpy(), rtol=1e-7)
    for i in range(n):
        want = sorted(keep[0][keep[1] == i].tolist() + [i])
        assert col[rowptr[i]:rowptr[i + 1]].tolist() == want
    order = ops.bfs_order_host(rowptr, col)
    assert sorted(order.tolist()) == list(range(n))
    with pytest.raises(Exception):
        ops.csr_build_host(np.array([[0, 99], [1, 2]]), n)


def test_cli_flags_match_reference_defaults():
    from dual_dmp_amd.cli import get_parser
    a = get_parser(False).parse_args(["-i", "x"])
    assert (a.pos_lr, a.norm_lr, a.iter, a.k1, a.k2, a.k3, a.k4, a.k5, a.grad_crip, a.bnfloop, a.gpu, a.port) == \
           (0.01, 0.01, 1000, 3.0, 4.0, 4.0, 4.0, 1.0, 0.8, 1, 0, 8080)
    assert a.norm_optim == "Adam" and a.viewer is True
    r = get_parser(True).parse_args(["-i", "x"])
    assert (r.k1, r.k2, r.k3, r.k4, r.k5, r.bnfloop) == (3.0, 0.0, 3.0, 4.0, 2.0, 5)
    with pytest.raises(SystemExit):
        get_parser(False).parse_args([])


def test_create_dataset_from_directory(tmp_path, oracle):
    from dual_dmp_amd import synth, datamaker
    gt, noisy, smooth = synth.make_triplet(*synth.icosphere(1))
    d = synth.write_dataset_dir(str(tmp_path), "ball", gt, noisy, smooth)
    mesh_dic, ds = datamaker.create_dataset(d)
    assert mesh_dic["mesh_name"] == "ball" and mesh_dic["gt_mesh"] is not None
    n_mesh, s_mesh = mesh_dic["n_mesh"], mesh_dic["s_mesh"]
    V, F, E = len(n_mesh.vs), len(n_mesh.faces), n_mesh.edges_count
    assert ds.z1.shape == (V, 16) and ds.z1.dtype == torch.float32 and ds.z2.shape == (F, 7)
    assert ds.edge_index.shape == (2, 2 * E) and ds.edge_index.dtype == torch.int64
    assert ds.face_index.shape == (2, 3 * F)
    assert (ds.num_nodes, ds.num_edges, ds.num_node_features) == (V, 2 * E, 16)
    assert not ds.contains_isolated_nodes and not ds.contains_self_loops
    od = oracle.OracleDataset(n_mesh, s_mesh)
    for k in ("z1", "z2", "x_pos", "x_norm", "edge_index", "face_index"):
        assert torch.equal(getattr(ds, k), getattr(od, k)), k
    np.random.seed(314)
    assert np.allclose(ds.z1.numpy(), np.random.normal(size=(V, 16)).astype(np.float32))
    # without ground truth (main4real.py)
    os.remove(os.path.join(d, "ball_gt.obj"))
    assert datamaker.create_dataset(d)[0]["gt_mesh"] is None


def _clean_obj(tmp_path, name="bunny"):
    """An arbitrary clean OBJ a user would hold: not unit-edge, not centred."""
    from dual_dmp_amd import synth
    from dual_dmp_amd.mesh import Mesh
    v, f = synth.icosphere(2)
    d = tmp_path / "datasets" / name
    d.mkdir(parents=True)
    path = str(d / "scan_clean.obj")
    Mesh(vs=v * 37.0 + np.array([5.0, -2.0, 11.0]), faces=f).save(path)
    return str(d), path


def test_preprocess_from_one_clean_obj(tmp_path, capsys):
    """`python -m dual_dmp_amd.preprocess -i <clean.obj> --level 0.2` = preprocess/noisemaker.py:12-80: the directory it
    leaves behind is what main.py -i reads (util/datamaker.py:26-35)."""
    from dual_dmp_amd import preprocess, datamaker, synth
    from dual_dmp_amd.mesh import Mesh
    d, path = _clean_obj(tmp_path)
    out = preprocess.main(["-i", path, "--level", "0.2", "--step", "30"])
    txt = capsys.readouterr().out
    assert "level       : 0.2" in txt and "[Finished] Vertices: 162, faces: 320, mad:" in txt
    assert out == d and not os.path.exists(path) and os.path.exists(os.path.join(d, "original", "scan_clean.obj"))
    g, n, s = (Mesh(os.path.join(d, "bunny_%s.obj" % k)) for k in ("gt", "noise", "smooth"))
    assert abs(synth.mean_edge_length(g.vs, g.edges) - 1.0) < 1e-6                   # noisemaker.py:32-36
    lo, hi = g.vs.min(0), g.vs.max(0)
    assert np.abs(lo + hi).max() < 1e-5                                               # bbox centre at the origin
    np.random.seed(314)                                                               # noisemaker.py:38-42
    noise = np.random.normal(loc=0, scale=0.2, size=(len(g.vs), 1))
    assert np.allclose(n.vs, g.vs + g.vn * noise, atol=1e-6)
    assert np.allclose(s.vs, synth.laplacian_smooth(n.vs, n.vv_ptr, n.vv_idx, steps=30), atol=1e-6)
    mesh_dic, ds = datamaker.create_dataset(d)
    assert mesh_dic["mesh_name"] == "bunny" and mesh_dic["gt_mesh"] is not None
    assert ds.z1.shape == (162, 16) and ds.z2.shape == (320, 7)


def test_preprocess_directory_with_a_noisy_scan(tmp_path):
    """`-i <dir>` = preprocess/preprocess.py:42-79: *_noise.obj (no ground truth) -> *_smooth.obj, unit mean edge."""
    from dual_dmp_amd import preprocess, datamaker, synth
    from dual_dmp_amd.mesh import Mesh
    v, f = synth.open_grid(9, 7)
    d = tmp_path / "datasets" / "scan"
    d.mkdir(parents=True)
    rng = np.random.default_rng(0)
    Mesh(vs=v * 12.0 + rng.normal(scale=0.05, size=v.shape), faces=f).save(str(d / "scan_noise.obj"))
    preprocess.main(["-i", str(d), "--step", "10"])
    n, s = Mesh(str(d / "scan_noise.obj")), Mesh(str(d / "scan_smooth.obj"))
    assert abs(synth.mean_edge_length(n.vs, n.edges) - 1.0) < 1e-6
    assert np.array_equal(n.faces, s.faces) and not np.allclose(n.vs, s.vs)
    mesh_dic, ds = datamaker.create_dataset(str(d))
    assert mesh_dic["gt_mesh"] is None and ds.x_pos.shape == (len(v), 3)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under dual-dmp_amd/ may reference it."""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dual-dmp_amd")
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "ddmp_oracle" not in txt and "oracle/" not in txt.replace("oracle/README", ""), f


def test_row_register_gemm_isa_audit():
    """The row-register GEMMs count the in-order VMEM counter by hand; that is sound only while their main loops hold no
    VMEM instruction the counts do not know about (no register-destination load, no scratch reload) and exactly the counted
    copies and waits per k-step.  scripts/check_rr_asm.py compiles csrc/gemm.hip for gfx950 (device only, no GPU needed)
    and audits every instantiation -- it is what found the spilled main loop of the reduction epilogue in round 3."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not installed")
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "check_rr_asm.py")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "kernels audited: 6, problems: 0" in r.stdout


def test_wide_wgrad_isa_audit():
    """The wide f16x3 wgrad of round 4 (csrc/gemm_tn_rm.hip): what made its predecessor slow was invisible in the source --
    loads in exec-masked side blocks, vmcnt(0) before every use.  scripts/check_tn_asm.py compiles the file for gfx950 (device
    only) and checks, for all 18 instantiations (three panel shapes), that each main loop is ONE basic block holding its 48 (24) MFMAs and all of its
    buffer loads, has no scratch traffic and no `s_waitcnt vmcnt(0)`."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not installed")
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "check_tn_asm.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert "kernels audited: 18, problems: 0" in r.stdout


def test_bench_parity_object_bounds():
    """bench.py's parity_1m object (HIP iteration 1 vs the oracle's, SURVEY.md 8d bounds): within the bounds -> ok; one face off
    by 5e-3 on a small mesh -> not ok (the quantile clause only forgives <= 1e-5 of the faces); a loss off by 1e-4 -> not ok."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("ddmp_bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from dual_dmp_amd import synth
    from dual_dmp_amd.mesh import Mesh
    v, f = synth.torus(24, 12)
    m = Mesh.__new__(Mesh)
    m.vs, m.faces = np.asarray(v, dtype=np.float64), np.asarray(f)
    Mesh.compute_face_normals(m)
    g = torch.Generator().manual_seed(3)
    pos = torch.tensor(m.vs, dtype=torch.float32)
    norm = torch.tensor(m.fn, dtype=torch.float32)
    hip = {"loss": [1.0, 0.9], "pos": pos, "norm": norm, "faces": len(f), "mesh_faces": m.faces, "gt_fn": m.fn}
    ref = {"loss": [1.0 + 2e-6, 0.91], "pos": pos + 1e-6 * torch.randn(pos.shape, generator=g),
           "norm": norm + 1e-6 * torch.randn(norm.shape, generator=g)}
    out = bench.parity_object(hip, ref)
    assert out["ok"] and out["rel"] < 1e-5 and out["dnorm_rows_above_1e-3"] == 0 and out["dnorm_p9999"] < 1e-4
    bad = dict(ref, norm=ref["norm"].clone())
    bad["norm"][7, 1] += 5e-3
    out = bench.parity_object(hip, bad)
    assert not out["ok"] and out["dnorm_rows_above_1e-3"] == 1 and out["max_abs_dnorm"] > 4e-3
    out = bench.parity_object(hip, dict(ref, loss=[1.0 + 1e-4, 0.9]))
    assert not out["ok"]
    # with the float64 forward: both float32 paths are measured against it; a HIP path no further from the truth than twice the
    # float32 reference passes the normals clause even when a few rows exceed the absolute bound
    truth_p, truth_n = pos.double(), norm.double()
    noisy_ref = dict(ref, pos64=truth_p, norm64=truth_n, f64_forward_s=1.0, norm=norm + 2e-3 * torch.randn(norm.shape, generator=g))
    noisy_hip = dict(hip, norm=norm + 2e-3 * torch.randn(norm.shape, generator=g))
    out = bench.parity_object(noisy_hip, noisy_ref)
    v = out["vs_float64"]
    assert out["max_abs_dnorm"] > 1e-3 and v["hip_within_2x_of_the_float32_reference"] and out["ok"]
    assert abs(v["hip"]["dnorm_rms"] - v["oracle_float32"]["dnorm_rms"]) < 0.5 * v["oracle_float32"]["dnorm_rms"]
    out = bench.parity_object(dict(hip, norm=norm + 2e-2 * torch.randn(norm.shape, generator=g)), noisy_ref)
    assert not out["vs_float64"]["hip_within_2x_of_the_float32_reference"] and not out["ok"]


def test_self_launch_builds_the_launcher_command(monkeypatch):
    """`main.py --gpus N` outside a launcher starts torch.distributed.run as a CHILD (never an exec: this process may not have
    touched a GPU, the children do) with a standalone loopback rendezvous -- the launcher picks its own port -- and hands the
    command line through."""
    from dual_dmp_amd import cli
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env

        class R:
            returncode = 7
        return R()
    monkeypatch.setattr(cli.subprocess, "run", fake_run)
    monkeypatch.setenv("RANK", "3")
    monkeypatch.setenv("WORLD_SIZE", "5")
    rc = cli._self_launch(["-i", "datasets/x", "--gpus", "4", "--iter", "20"], False, 4)
    cmd = seen["cmd"]
    assert rc == 7 and cmd[1:3] == ["-m", "torch.distributed.run"] and "--standalone" in cmd
    assert cmd[cmd.index("--local-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert "--master-port" not in cmd and cmd[-6:] == ["-i", "datasets/x", "--gpus", "4", "--iter", "20"]
    assert cmd[-7].endswith("main.py") and os.path.exists(cmd[-7])
    assert "RANK" not in seen["env"] and "WORLD_SIZE" not in seen["env"] and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert cli._self_launch(["-i", "x"], True, 2) == 7 and seen["cmd"][-3].endswith("main4real.py")
