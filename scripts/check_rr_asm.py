#!/usr/bin/env python3
"""Static audit of gemm_rr_kernel's ISA (csrc/gemm_rr.inc).  The kernel counts the in-order VMEM counter by hand, which is
only sound if (1) its main loop holds no VMEM instruction the hand count does not know about -- no load with a register
destination (every load is a global_load_lds from inline asm), no scratch traffic (spills) --, (2) every wait of the loop
is one of the counted ones and none is a compiler-inserted vmcnt(0) drain, (3) per k-step the loop issues exactly the
copies the counts assume.  A first version of the kernel kept A in a register ring behind asm loads; this script found the
register allocator copying ring registers while their loads were in flight, which is why the ring now lives in LDS.
    usage: check_rr_asm.py [gemm_dev.s]     (compiles csrc/gemm.hip --cuda-device-only -S if omitted)"""
import os, re, subprocess, sys, tempfile

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    asm = sys.argv[1]
else:
    asm = os.path.join(tempfile.gettempdir(), "ddmp_gemm_dev.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--cuda-device-only",
                           "-S", os.path.join(root, "dual-dmp_amd", "csrc", "gemm.hip"), "-o", asm], stderr=subprocess.DEVNULL)
lines = open(asm).read().split("\n")


def audit(name, body):
    m = re.search(r"gemm_rr_kernelILi(\d)EL[bi](\d)ELi(\d)E", name)
    pm, stats, nb = int(m.group(1)), int(m.group(2)), int(m.group(3))
    na = 4 if pm == 2 else 2
    nv = 4 + na
    labels = {mm.group(1): i for i, l in enumerate(body) for mm in [re.match(r"^(\.LBB\d+_\d+):", l)] if mm}
    best = None
    for i, l in enumerate(body):
        mm = re.match(r"\s+s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            n = sum("v_mfma" in x for x in body[labels[mm.group(1)]:i])
            if best is None or n > best[2]:
                best = (labels[mm.group(1)], i, n)
    lo, hi, nm = best
    # the k-step proper = from the loop header to the counted barrier wait (what follows is the tile epilogue)
    loop = [l.strip() for l in body[lo:hi + 1]]
    end = max(i for i, t in enumerate(loop) if t.startswith("s_barrier"))
    first_bar = min(i for i, t in enumerate(loop) if t.startswith("s_barrier"))
    kstep = loop[:first_bar + 1]
    problems = []
    dma = sum(t.startswith("global_load_lds_dwordx4") for t in kstep)
    if dma != nv:
        problems.append("k-step issues %d copies, the counts assume %d" % (dma, nv))
    for t in kstep:
        if (t.startswith("global_load") and "lds" not in t) or t.startswith("buffer_load") or t.startswith("scratch_") \
                or t.startswith("global_store") or t.startswith("global_atomic"):
            problems.append("uncounted VMEM in the k-step: " + t)
    waits = [int(w.group(1)) for t in kstep for w in [re.search(r"vmcnt\((\d+)\)", t)] if t.startswith("s_waitcnt") and w]
    want = [(nb - 1) * nv, na + (nb - 2) * nv]
    if waits != want:
        problems.append("vmcnt waits of the k-step %s, expected %s" % (waits, want))
    scratch = sum(l.strip().startswith("scratch_") for l in body)
    mf = sum("v_mfma" in t for t in kstep)
    print("%-70s PM %d STATS %d NB %d: %d MFMAs, %d copies, waits %s per k-step; scratch ops in the kernel %d%s" % (
        name[:70], pm, stats, nb, mf, dma, waits, scratch, "" if not problems else "   <-- PROBLEMS"))
    for p in problems:
        print("   !!", p)
    return len(problems) + (1 if scratch else 0)


total = found = 0
i = 0
while i < len(lines):
    m = re.match(r"^(_ZN\S*gemm_rr_kernel\S*):", lines[i])
    if not m:
        i += 1
        continue
    j = i + 1
    while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
        j += 1
    total += audit(m.group(1), lines[i:j])
    found += 1
    i = j
print("kernels audited: %d, problems: %d" % (found, total))
sys.exit(1 if total or not found else 0)
