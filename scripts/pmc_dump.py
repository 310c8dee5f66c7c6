#!/usr/bin/env python3
"""Counter sums per (kernel, grid size) from rocprofv3 --pmc passes (rocpd sqlite).  usage: pmc_dump.py <results.db> [name filter]"""
import re
import sqlite3
import sys


def table(db, frag):
    names = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    hit = [t for t in names if frag in t]
    if not hit:
        raise SystemExit("no table matching %s in %s" % (frag, names))
    return hit[0]


db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
pmc, info, kd, ks = table(db, "pmc_event"), table(db, "info_pmc"), table(db, "kernel_dispatch"), table(db, "info_kernel_symbol")
cols = [r[1] for r in db.execute("pragma table_info(%s)" % kd)]
gx = "k.grid_size_x" if "grid_size_x" in cols else ("k.grid_x" if "grid_x" in cols else "0")
q = ("select s.display_name, {gx}, i.name, count(distinct k.id), sum(p.value), sum(distinct (k.end - k.start) * 1000003 + k.id) "
     "from {pmc} p join {info} i on p.pmc_id = i.id join {kd} k on p.event_id = k.event_id join {ks} s on k.kernel_id = s.id "
     "group by 1, 2, 3 order by 1, 2, 3").format(gx=gx, pmc=pmc, info=info, kd=kd, ks=ks)
for name, grid, cname, n, val, _ in db.execute(q):
    short = re.sub(r"\(.*", "", re.sub(r"\(anonymous namespace\)::", "", name))
    if flt and flt not in short:
        continue
    print("%-60s grid %-10s %-34s launches %3d  sum %.6g  per launch %.6g" % (short[:60], grid, cname, n, val, val / max(n, 1)))
