"""bench.py end to end at a small size: the ONE JSON line the driver parses carries every field of the contract (and the fields the
round-5 review asked for) -- a crash or a missing key here would void the round's measurement."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_fields_at_20k_faces():
    env = dict(os.environ)
    for k in list(env):
        if k.startswith("DDMP_") and k != "DDMP_LIB":
            del env[k]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--faces", "20000", "--steps", "3", "--warmup", "2",
                        "--cpu-iters", "1", "--cpu-sample-faces", "20000", "--profile-steps", "1"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.strip().splitlines() if l.strip()]
    assert len(lines) == 1, lines                                  # stdout carries exactly one line
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["unit"] == "iters/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 2 and d["data"] == "synthetic" and d["scaling"] in ("strong", "weak")
    assert d["value"] > 0 and abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]
    assert d["vs_baseline"] is None and "datasets.zip" in d["vs_baseline_note"]
    assert d["dtype"].startswith("f32 (f16x3 split-MFMA")          # the arithmetic type, labelled as what it is
    assert "workload" in d["config"] and "model" not in d["config"] and 19000 <= d["config"]["faces"] <= 20000
    for name in ("roofline", "roofline_gather"):
        ro = d[name]
        assert ro["bound"] in ("hbm", "mfma") and ro["unit"] in ("GB/s", "TFLOP/s")
        assert ro["achieved"] > 0 and ro["peak"] > 0 and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-3
        assert "traffic" in ro and "traffic_measured_on" in ro
        assert 0 < ro["frac_survey_8d"] <= ro["frac"] + 1e-6 or ro["bound"] == "mfma"      # SURVEY 8d bytes <= the fused forms' count
    cb = d["cpu_baseline"]
    assert cb["value"] > 0 and cb["unit"] == "iters/s" and cb["cores"] >= 1 and cb["kind"] == "port" and "faces" in cb["sample"]
    assert d["value"] > 10 * cb["value"]                            # north_star: >= 10x the CPU reference on the same host
    # gate-open headline with the gate-closed figure and the reference run's blend beside it
    assert d["gate_open_ms_per_step"] == d["ms_per_step"] and d["gate_closed_ms_per_step"] > 0
    assert abs(d["reference_run_blend_ms_per_step"] - (0.1 * d["gate_closed_ms_per_step"] + 0.9 * d["gate_open_ms_per_step"])) < 2e-3
    p = d["parity_1m"]
    assert p["ok"] and "UNPINNED" in p["what"] and p["faces"] == d["config"]["faces"]
    t2 = p["iter2_teacher_forced"]                                  # the oracle's state after iteration 1 -> one HIP iteration
    assert "error" not in t2 and t2["ok"] and t2["rel"] <= 1e-5, t2
    assert d["bf16"]["ms_per_step"] > 0 and d["gemm_scale_overflow"] is None
