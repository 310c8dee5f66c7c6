#!/usr/bin/env python3
"""Turn the raw output of scripts/spmm_pmc.sh (pmc_dump.py lines) into the per-output-row table committed as
profiles/rNN_gather_tcp_tcc_counters.txt.  usage: gather_counter_summary.py <raw.txt> [C]"""
import re
import sys

raw = [l for l in open(sys.argv[1]) if "launches" in l and "per launch" in l]
C = int(sys.argv[2]) if len(sys.argv) > 2 else 512
rows = {}
for l in raw:
    m = re.match(r"void (\S.*?)\s+grid (\d+)\s+(\S+)\s+launches\s+(\d+)\s+sum (\S+)\s+per launch (\S+)", l)
    if m:
        rows.setdefault((m.group(1).strip(), int(m.group(2))), {})[m.group(3)] = float(m.group(6))
print("# scripts/r06_gather_counters.sh %d (= scripts/archive/spmm_pmc.sh) on one MI355X: TCP / TCC / SQ counters of the plain gather at C = %d on the 1M-face torus in RCB" % (C, C))
print("# order (the engines' numbering): 3 launches each on the FACE graph (1,000,000 rows x 4 entries) and the VERTEX graph (500,000 rows x 7")
print("# entries).  The wide launches run spmm_patch2_kernel (LDS-patch gather: NE = 4 | 8 register entries); small spmm_lean_kernel")
print("# launches beside them, if any, are heavy-chunk lists (round 5; since round 6 oversized chunks of a regular mesh are split inside the launch).")
print("# Seven rocprofv3 --pmc passes (kernel trace only), sums over the device, per launch and OUTPUT ROW (%d bytes of output).\n" % (4 * C))
for (k, grid), c in sorted(rows.items(), key=lambda kv: -kv[0][1]):
    n_rows = {4001792: 1000000, 2000896: 500000}.get(grid)
    if n_rows is None:
        print("%-52s grid %-8d  (%d workgroups: a heavy-chunk list)  L2 -> fabric reads %.0f, writes %.0f per launch" % (
            k, grid, grid // 256, c.get("TCC_EA0_RDREQ_sum", 0), c.get("TCC_EA0_WRREQ_sum", 0)))
        continue
    g = lambda name: c.get(name, float("nan")) / n_rows
    print("%s   grid %d = %d rows" % (k, grid, n_rows))
    print("   L1 accesses (TCP_TOTAL_CACHE_ACCESSES)      %8.1f      L1 -> L2 read requests (TCP_TCC_READ_REQ)  %8.1f" % (g("TCP_TOTAL_CACHE_ACCESSES_sum"), g("TCP_TCC_READ_REQ_sum")))
    print("   L2 hits / misses                            %8.1f / %.1f" % (g("TCC_HIT_sum"), g("TCC_MISS_sum")))
    print("   L2 -> fabric reads (TCC_EA0_RDREQ)          %8.1f      (%d bytes per row in ~%.0f-byte requests)   writes (64 B) %.1f" % (
        g("TCC_EA0_RDREQ_sum"), 4 * C, 4.0 * C / max(g("TCC_EA0_RDREQ_sum"), 1e-9), g("TCC_EA0_WRREQ_sum")))
    gate = c.get("TCP_GATE_EN1_sum", float("nan"))
    print("   TCP active cycles (TCP_GATE_EN1) %.3g   stalled on TA data return %.0f %%   with requests pending %.0f %%" % (
        gate, 100.0 * c.get("TCP_TCP_TA_DATA_STALL_CYCLES_sum", 0) / gate, 100.0 * c.get("TCP_PENDING_STALL_CYCLES_sum", 0) / gate))
    if "SQ_WAVE_CYCLES" in c:
        print("   wave cycles waiting on any instruction %.0f %%   VALU instructions per row %.1f   LDS instructions per row %.1f   VMEM reads per row %.2f" % (
            100.0 * c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"], g("SQ_INSTS_VALU"), g("SQ_INSTS_LDS"), g("SQ_INSTS_VMEM_RD")))
    print()
print("# raw counter lines\n#")
for l in raw:
    print(l.rstrip()[:200])
