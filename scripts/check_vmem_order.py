#!/usr/bin/env python3
"""Static check of the row-panel GEMM kernels' ISA: panel_barrier(n) waits with s_waitcnt vmcnt(n), which is only
correct if the n youngest VMEM instructions before it are the ones that may still be in flight: with two stage buffers
the register loads of A -- every global_load_lds (the W copies the barrier must cover) has to be OLDER; with three
(f16 modes) everything issued since the previous barrier, and nothing older.  The scheduler is free to reorder
independent VMEM instructions, so this is checked on the generated code (linear scan: loop bodies are unrolled).  usage: check_vmem_order.py [gemm.s]  (compiles csrc/gemm.hip if omitted)"""
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
    three = re.search(r"gemm_panel_kernelILi\d+ELi\d+ELi1[34]E", name) is not None      # f16 modes: three stage buffers
    recent = []                  # VMEM instructions not yet known complete, oldest first: (kind, barrier segment)
    seg = 0
    i += 1
    while i < len(lines) and not lines[i].strip().startswith("s_endpgm"):
        t = lines[i].strip()
        if t.startswith("global_load_lds"):
            recent.append(("lds", seg))
        elif t.startswith("global_load") or t.startswith("buffer_load"):
            recent.append(("reg", seg))
        elif t.startswith("global_store") or t.startswith("global_atomic"):
            recent.append(("st", seg))
        w = re.match(r"s_waitcnt vmcnt\((\d+)\)", t)
        if w:
            n = int(w.group(1))
            recent = recent[len(recent) - n:] if n else []
            at_barrier = any(lines[j].strip().startswith("s_barrier") for j in range(i + 1, min(i + 3, len(lines))))
            if at_barrier:
                n_checked += 1
                # two buffers: no W copy may be in flight at the barrier.  three: only copies issued since the
                # previous barrier (they have one more stage to land).
                late = [k for k, sg in recent if k == "lds" and (not three or sg < seg)]
                if late:
                    bad += 1
                    print("HAZARD %s: vmcnt(%d) before s_barrier leaves %d stale global_load_lds in flight (tail: %s)"
                          % (name[:90], n, len(late), recent[-8:]))
        if t.startswith("s_barrier"):
            seg += 1
        i += 1
print("checked %d barrier waits in gemm_panel_kernel instances: %d hazards" % (n_checked, bad))
sys.exit(1 if bad else 0)
