"""The library's A/B switches are product configurations too: each one listed here runs ONE training
iteration at 144,400 faces in its own process (the switches are read once per process) and must agree with the default build's
iteration from the same weights to float32 rounding -- loss, outputs, gradients.  Same mathematics, other kernels / orders."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

SWITCHES = [
    {"DDMP_SPMM_PATCH": "0"},                  # lean gather everywhere
    {"DDMP_SPMM_PATCH": "1"},                  # LDS-patch gather wherever it applies (incl. its fused-reduction form)
    {"DDMP_SPMM_PATCH_NE": "0"},               # ... with the entries read from LDS per slab
    {"DDMP_SPMM_LEAN": "0"},                   # round-2 slab gather (the fallback of the lean kernel)
    {"DDMP_GEMM_RR": "0"},                     # row-panel instead of row-register GEMMs
    {"DDMP_GEMM_MODE": "6"},                   # bf16x6 everywhere
    {"DDMP_GEMM_PANEL": "0"},                  # tiled kernels instead of the row panels
    # DDMP_UNFUSE=<names>: fused routes composed from their parts instead (round 6: one switch for what were eleven)
    {"DDMP_UNFUSE": "dgrad_red"},              # dgrad without the reductions epilogue
    {"DDMP_UNFUSE": "dgrad_red_narrow"},
    {"DDMP_UNFUSE": "bnbwd_narrow,bnbwd_l0"},  # narrow layers: bn_bwd_apply + plain GEMMs
    {"DDMP_UNFUSE": "gather_bwd"},             # transform-first layers: bn_bwd_apply + plain gather
    {"DDMP_UNFUSE": "tail,wprep"},
    {"DDMP_UNFUSE": "stats"},
    {"DDMP_UNFUSE": "equal_width"},            # equal-width layers transform first (the reference's own order)
    {"DDMP_UNFUSE": "stats,gather_bwd,bnbwd_l0,bnbwd_narrow,dgrad_red,dgrad_red_narrow,tail,wprep"},   # everything composed
]
BF16_SWITCHES = [
    {"DDMP_UNFUSE": "bf16_spmm_red"},          # backward reductions as separate passes
    {"DDMP_UNFUSE": "bf16_gemm"},              # no BatchNorm backward on the GEMM operand loads
    {"DDMP_GEMM_RR": "0"},
    {"DDMP_SPMM_LEAN": "0"},
    {"DDMP_SPMM_PATCH": "0"},
]


def _run(tmp, name, env_extra, dtype="f32"):
    out = os.path.join(tmp, name + ".npz")
    env = dict(os.environ)
    for k in list(env):
        if k.startswith("DDMP_") and k not in ("DDMP_LIB",):
            del env[k]
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(HERE, "switch_worker.py"), out, dtype], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, (env_extra, r.stdout[-2000:], r.stderr[-3000:])
    return dict(np.load(out))


@pytest.fixture(scope="module")
def baseline(tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp("switches"))
    return tmp, _run(tmp, "default", {})


@pytest.mark.parametrize("env", SWITCHES, ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_switch_agrees_with_the_default(baseline, env):
    tmp, ref = baseline
    got = _run(tmp, "_".join("%s%s" % kv for kv in env.items()), env)
    assert np.isfinite(got["loss"]) and abs(got["loss"] - ref["loss"]) <= 3e-6 * abs(ref["loss"]), (got["loss"], ref["loss"])
    assert np.abs(got["pos"] - ref["pos"]).max() <= 2e-5
    assert np.abs(got["norm"] - ref["norm"]).max() <= 5e-4
    for k in ("m0", "m1"):
        assert np.linalg.norm(got[k] - ref[k]) <= 3e-4 * np.linalg.norm(ref[k]), k


@pytest.fixture(scope="module")
def baseline_bf16(baseline):
    tmp, _ = baseline
    return tmp, _run(tmp, "bf16_default", {}, "bf16")


@pytest.mark.parametrize("env", BF16_SWITCHES, ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_bf16_switch_agrees_with_the_default(baseline_bf16, env):
    """bf16 features: the same iteration through the other kernels / the unfused passes -- bf16 rounding of intermediate tensors
    differs between the routes, so the agreement is to bf16 accuracy."""
    tmp, ref = baseline_bf16
    got = _run(tmp, "bf16_" + "_".join("%s%s" % kv for kv in env.items()), env, "bf16")
    assert np.isfinite(got["loss"]) and abs(got["loss"] - ref["loss"]) <= 5e-3 * abs(ref["loss"]), (got["loss"], ref["loss"])
    for k in ("m0", "m1"):
        assert np.linalg.norm(got[k] - ref[k]) <= 5e-2 * np.linalg.norm(ref[k]), k
