#!/usr/bin/env python3
"""HBM traffic per kernel from two rocprofv3 PMC passes (rocpd sqlite) of the same command:
      rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_F -o r -- python3 bench.py ...
      rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_W -o r -- python3 bench.py ...
   usage: pmc_traffic.py <fetch.db> <write.db> [note] [dtype] [iterations] > profiles/rNN_pmc_hbm_traffic[_bf16].json
The output records the content hash of the kernel sources it was measured on (csrc_sha16) and the feature dtype:
bench.py quotes a traffic figure only for the tree it belongs to.
Units (MI355X_MICROARCH.md): both counters are in KB (x1024); FETCH_SIZE is doubled on gfx950, which reports half of the
bytes of wide coalesced reads.  Kernel times come from the kernel trace of the FETCH pass."""
import json
import re
import sqlite3
import sys


def table(db, frag):
    names = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    hit = [t for t in names if frag in t]
    if not hit:
        raise SystemExit("no table matching %s in %s" % (frag, names))
    return hit[0]


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    pmc, info, kd, ks = table(db, "pmc_event"), table(db, "info_pmc"), table(db, "kernel_dispatch"), table(db, "info_kernel_symbol")
    # only the dispatches between bench.py's two trace markers (its timed region), when the trace has them
    marks = db.execute("select k.start, k.end from {kd} k join {ks} s on k.kernel_id = s.id where s.display_name like "
                       "'%ddmp_trace_marker_kernel%' order by 1".format(kd=kd, ks=ks)).fetchall()
    win = " and k.start > %d and k.end < %d" % (marks[0][1], marks[-1][0]) if len(marks) >= 2 else ""
    q = ("select s.display_name, count(distinct k.id), sum(p.value), sum(distinct (k.end - k.start) * 1000003 + k.id) "
         "from {pmc} p join {info} i on p.pmc_id = i.id join {kd} k on p.event_id = k.event_id "
         "join {ks} s on k.kernel_id = s.id where i.name = ?{win} group by 1").format(pmc=pmc, info=info, kd=kd, ks=ks, win=win)
    out = {}
    for name, n, v, _ in db.execute(q, (counter,)):
        out[name] = [n, v]
    dur = {}
    for name, ns in db.execute("select s.display_name, sum(k.end - k.start) from {kd} k join {ks} s on k.kernel_id = s.id "
                               "where 1{win} group by 1".format(kd=kd, ks=ks, win=win)):
        dur[name] = ns
    return out, dur


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name)


def family(k):
    if "gemm" in k or "split_w" in k or "reduce_splits" in k or "w_planes" in k or "f16s_" in k:
        return "gemm"
    if "spmm" in k:
        return "spmm"
    if "colreduce" in k or "reduce_partials" in k or "bn_" in k:
        return "bn/colreduce"
    return "other"


def main():
    f, fdur = per_kernel(sys.argv[1], "FETCH_SIZE")
    w, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    dtype = sys.argv[4] if len(sys.argv) > 4 else "f32"
    iterations = int(sys.argv[5]) if len(sys.argv) > 5 else 4             # steps + warmup of the profiled command
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    rows = []
    for k in sorted(f, key=lambda k: -fdur.get(k, 0)):
        rd = f[k][1] * 1024 * 2 / 1e9
        wr = w.get(k, [0, 0])[1] * 1024 / 1e9
        rows.append({"kernel": short(k), "launches": f[k][0], "hbm_read_GB": round(rd, 3), "hbm_write_GB": round(wr, 3),
                     "ms": round(fdur.get(k, 0) / 1e6, 3)})
    fam = {}
    for r in rows:
        a = fam.setdefault(family(r["kernel"]), {"launches": 0, "hbm_read_GB": 0.0, "hbm_write_GB": 0.0, "kernel_ms": 0.0})
        a["launches"] += r["launches"]
        a["hbm_read_GB"] = round(a["hbm_read_GB"] + r["hbm_read_GB"], 3)
        a["hbm_write_GB"] = round(a["hbm_write_GB"] + r["hbm_write_GB"], 3)
        a["kernel_ms"] = round(a["kernel_ms"] + r["ms"], 2)
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); bytes = KB*1024, FETCH_SIZE doubled per "
                       "MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads). " + note,
               "dtype": dtype, "iterations": iterations, "csrc_sha16": bench.csrc_sha16(), "per_family": fam, "per_kernel": rows[:40]}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
