#!/usr/bin/env python3
"""Static check of the row-panel GEMM kernels' ISA: panel_barrier(NL) waits with s_waitcnt vmcnt(NL), which is only
correct if the NL youngest VMEM instructions before it are the register loads of A -- every global_load_lds (the W
copies, which the barrier must cover) has to be OLDER.  The scheduler is free to reorder independent VMEM instructions,
so this is checked on the generated code.  usage: check_vmem_order.py [gemm.s]  (compiles csrc/gemm.hip if omitted)"""
import os, re, subprocess, sys, tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    asm = sys.argv[1]
else:
    asm = os.path.join(tempfile.gettempdir(), "ddmp_gemm_dev.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--cuda-device-only",
                           "-S", os.path.join(root, "dual-dmp_amd", "csrc", "gemm.hip"), "-o", asm], stderr=subprocess.DEVNULL)
lines = open(asm).read().split("\n")
bad = n_checked = 0
i = 0
while i < len(lines):
    m = re.match(r"^(_ZN\S*gemm_panel_kernel\S*):", lines[i])
    if not m:
        i += 1
        continue
    name = m.group(1)
    recent = []                  # VMEM instructions since the last wait / barrier, oldest first
    i += 1
    while i < len(lines) and not lines[i].strip().startswith("s_endpgm"):
        t = lines[i].strip()
        if t.startswith("global_load_lds"):
            recent.append("lds")
        elif t.startswith("global_load") or t.startswith("buffer_load"):
            recent.append("reg")
        elif t.startswith("global_store") or t.startswith("global_atomic"):
            recent.append("st")
        w = re.match(r"s_waitcnt vmcnt\((\d+)\)", t)
        if w:
            n = int(w.group(1))
            n_checked += 1
            young = recent[len(recent) - n:] if n else []
            if "lds" in young:
                bad += 1
                print("HAZARD %s: vmcnt(%d) leaves a global_load_lds in flight (stream tail: %s)" % (name[:90], n, recent[-8:]))
            recent = recent[len(recent) - n:] if n else []
        i += 1
print("checked %d vmcnt waits in gemm_panel_kernel instances: %d hazards" % (n_checked, bad))
sys.exit(1 if bad else 0)
