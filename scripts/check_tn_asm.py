#!/usr/bin/env python3
"""Static audit of gemm_tn_rm_kernel's ISA (csrc/gemm_tn_rm.hip, the wide f16x3 wgrad of round 4).  What broke the round-3
kernel was invisible in the source: loads in exec-masked side blocks (vmcnt(0) before every use, no interleaving with the
MFMAs).  For every instantiation this checks the main loops (the basic blocks that branch back to themselves and hold MFMAs):
  (1) ONE basic block per loop: 48 MFMAs (24 in the 256 x 128 / 128 x 256 panels), the stage's buffer loads, fragment reads and LDS writes of two iterations in it --
      no load sits in a side block;
  (2) no scratch (spill) traffic in it;
  (3) no `s_waitcnt vmcnt(0)`: every wait leaves the younger stage's loads in flight;
  (4) (round 5) the f16 split is the v_fma_mix form of ddmp_common.h: no f16 -> f32 conversion (v_cvt_f32_f16) in the loop -- what
      hipcc emitted for half of the elements when the split was written as casts (3.7 instead of 2.5 VALU per element).
    usage: check_tn_asm.py [gemm_tn_rm_dev.s]     (compiles csrc/gemm_tn_rm.hip --cuda-device-only -S if omitted)"""
import os, re, subprocess, sys, tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    asm = sys.argv[1]
else:
    asm = os.path.join(tempfile.gettempdir(), "ddmp_gemm_tn_rm_dev.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--cuda-device-only",
                           "-S", os.path.join(root, "dual-dmp_amd", "csrc", "gemm_tn_rm.hip"), "-o", asm], stderr=subprocess.DEVNULL)
lines = open(asm).read().split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN.*gemm_tn_rm_kernelILb[01]ELb[01]ELi[01]ELi\d+ELi\d+E.*:", l)]
n_kernels = problems = 0
for st in starts:
    end = next(i for i in range(st, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[st:end]
    m = re.search(r"gemm_tn_rm_kernelILb([01])ELb([01])ELi([01])ELi(\d+)ELi(\d+)E", body[0])
    pro, gdual, pp, tm, tk = (int(x) for x in m.groups())
    n_kernels += 1
    # basic blocks
    blocks, cur = [], ["entry", []]
    for l in body[1:]:
        mm = re.match(r"^(\.LBB\d+_\d+):", l)
        if mm:
            blocks.append(cur)
            cur = [mm.group(1), []]
        else:
            t = l.strip()
            if t and not t.startswith(";") and not t.startswith("."):
                cur[1].append(t)
    blocks.append(cur)
    loops = [(name, ins) for name, ins in blocks
             if any(re.match(r"s_cbranch_\w+ " + re.escape(name) + r"$", t) for t in ins) and sum("v_mfma" in t for t in ins) > 0]
    want_loops = 2 if pp else 1
    tag = "gemm_tn_rm_kernel<PRO=%d, GDUAL=%d, PP=%d, %dx%d>" % (pro, gdual, pp, tm, tk)
    if len(loops) != want_loops:
        print("%s: %d self-looping MFMA blocks, expected %d" % (tag, len(loops), want_loops))
        problems += 1
    for name, ins in loops:
        n_mfma = sum("v_mfma" in t for t in ins)
        n_ld = sum(t.startswith("buffer_load") or t.startswith("global_load") for t in ins)
        n_scratch = sum(t.startswith("scratch_") for t in ins)
        waits = [int(x) for t in ins if t.startswith("s_waitcnt") for x in re.findall(r"vmcnt\((\d+)\)", t)]
        gp, zp = tm // 128, tk // 128                   # loads per thread and stage of the two operands
        want_ld = 2 * (gp * (2 if gdual else 1) + zp)
        want_mfma = 48 if (tm, tk) == (256, 256) else 24
        bad = []
        if n_mfma != want_mfma:
            bad.append("%d MFMAs (%d)" % (n_mfma, want_mfma))
        if n_ld != want_ld:
            bad.append("%d loads (%d)" % (n_ld, want_ld))
        if n_scratch:
            bad.append("%d scratch operations" % n_scratch)
        if any(w == 0 for w in waits):
            bad.append("vmcnt(0) in the loop: waits %s" % waits)
        n_back = sum(t.startswith("v_cvt_f32_f16") for t in ins)
        n_mix = sum(t.startswith("v_fma_mix") for t in ins)
        if n_back or not n_mix:
            bad.append("%d v_cvt_f32_f16, %d v_fma_mix: the split is not the v_fma_mix form" % (n_back, n_mix))
        print("%s %s: %d instructions, %d MFMA, %d loads, vmcnt waits %s%s" % (
            tag, name, len(ins), n_mfma, n_ld, waits, ("   <-- " + "; ".join(bad)) if bad else ""))
        problems += bool(bad)
print("kernels audited: %d, problems: %d" % (n_kernels, problems))
sys.exit(1 if problems or n_kernels != 18 else 0)
