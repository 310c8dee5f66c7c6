#!/usr/bin/env python3
"""Fill the {PLACEHOLDER}s of DESIGN.md from profiles/r05_bench_line.json (+ the bf16 line).  usage: fill_design_numbers.py [STEP_RANGE]"""
import json
import os
import re
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.loads(open(os.path.join(R, "profiles", "r05_bench_line.json")).read().strip().split("\n")[-1])
b = json.loads(open(os.path.join(R, "profiles", "r05_bf16_bench_line.json")).read().strip().split("\n")[-1])
rg, ro, irr, cpu, par, ab, cp = d["roofline_gather"], d["roofline"], d["irregular"], d["cpu_baseline"], d["parity_1m"], d["gemm_mode_ab"], d["device_copy"]
km = d["kernel_ms_per_step"]
gemm_ms = sum(v for k, v in km.items() if k.startswith("gemm"))
rest = sum(v for k, v in km.items() if not k.startswith("gemm") and k != "spmm")
bf = d.get("bf16") or {}
bl = b                      # the bf16 evidence line (its own run)
v64 = par.get("vs_float64", {})
f = {
    "STEP_RANGE": sys.argv[1] if len(sys.argv) > 1 else "43.4–44.3",
    "STEP_MS": "%.2f" % d["ms_per_step"], "STEP_ITS": "%.2f" % d["value"],
    "IRR_MS": "%.2f" % irr["ms_per_step"], "IRR_RATIO": "%.3f" % irr["vs_regular"], "IRR_FRAC": "%.3f" % irr["roofline_gather"]["frac"],
    "GEMM_MS": "%.1f" % ro["ms_per_step"], "GEMM_FRAC": "%.3f" % ro["frac"], "GEMM_GBS": "%.0f" % ro["achieved"],
    "GEMM_TRAFFIC": "%.2f" % ((ro.get("traffic") or 0) / 1e9), "GEMM_TR_RATIO": "%.2f" % ((ro.get("traffic") or 0) / ro["alg_bytes_per_launch"]),
    "GEMM_MFMA": "%.2f" % ro["other_roofline"]["frac"],
    "SPMM_MS": "%.1f" % rg["ms_per_step"], "GFRAC": "%.3f" % rg["frac"], "GFACE": "%.3f" % rg["by_fan_in"]["4"]["frac"],
    "GVERT": "%.3f" % rg["by_fan_in"]["7"]["frac"], "GCOPY": "%.2f" % rg["frac_of_device_copy"], "GGUIDE": "%.2f" % rg["frac_of_guide_copy_6290"],
    "G_TRAFFIC": "%.2f" % ((rg.get("traffic") or 0) / 1e9), "G_TR_RATIO": "%.2f" % ((rg.get("traffic") or 0) / rg["alg_bytes_per_launch"]),
    "BF16_MS": "%.2f" % bl["ms_per_step"], "BF16_ITS": "%.1f" % bl["value"], "BF16_G_MS": "%.1f" % bl["roofline_gather"]["ms_per_step"],
    "BF16_GFRAC": "%.3f" % bl["roofline_gather"]["frac"],
    "P_REL": "%.1e" % par["rel"], "P_DPOS": "%.1e" % par["max_abs_dpos"], "P_DNORM": "%.1e" % par["max_abs_dnorm"], "P_BY": str(par["normals_ok_by"]),
    "P_MAD": "%.1e" % par["mad_delta_deg"], "P_HIP_RMS": "%.1e" % v64["hip"]["dnorm_rms"], "P_HIP_MAX": "%.1e" % v64["hip"]["max_abs_dnorm"],
    "P_ORA_RMS": "%.1e" % v64["oracle_float32"]["dnorm_rms"], "P_ORA_MAX": "%.1e" % v64["oracle_float32"]["max_abs_dnorm"],
    "AB_F32": "%.1f" % ab["f32_mfma_ms_per_step"], "AB_BF": "%.1f" % ab["bf16x6_ms_per_step"], "AB_F16": "%.1f" % ab["f16x3_ms_per_step"],
    "AB_REL": "%.1e" % ab["f16x3_forward_vs_f32_mfma_max_layer_rel_l2"],
    "CP_PLAIN": "%.2f" % (cp["ddmp_copy_probe_GBs"]["plain"] / 1e3), "CP_NT": "%.2f" % (cp["ddmp_copy_probe_GBs"]["nontemporal"] / 1e3),
    "CP_SLAB": "%.2f" % (cp["slab_pattern_copy_GBs_by_row_width"]["512"] / 1e3), "CP_TORCH": "%.2f" % (cp["torch_copy__GBs"] / 1e3),
    "CPU_S": "%.1f" % (1.0 / cpu["value"]), "CPU_T": str(cpu["threads"]), "CPU_H": str(cpu["host_cores"]),
    "CPU_EACH": "/".join("%.0f" % x for x in cpu["s_per_iter_each"]) + " s", "CPU_X": "%.0f" % (d["value"] / cpu["value"]),
    "REST_MS": "%.1f" % rest, "HIDE_MS": "%.1f" % (gemm_ms + km["spmm"] + rest - d["ms_per_step"]),
}
p = os.path.join(R, "DESIGN.md")
s = open(p).read()
missing = set(re.findall(r"\{([A-Z0-9_]+)\}", s)) - set(f)
assert not missing, missing
for k, v in f.items():
    s = s.replace("{" + k + "}", v)
open(p, "w").write(s)
print("filled", len(f), "placeholders")
