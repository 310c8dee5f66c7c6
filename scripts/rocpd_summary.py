#!/usr/bin/env python3
"""Turn a rocprofv3 (rocpd sqlite) kernel trace into the text summary committed under profiles/.
usage: rocpd_summary.py <results.db> [--top N]"""
import re
import sqlite3
import sys


def short(name):
    if name.startswith("_Z"):                       # rocprofv3 leaves names with __bf16 parameters mangled
        m = re.match(r"_ZN12_GLOBAL__N_1\d+([A-Za-z_0-9]+?)I((?:L[ib]\d+E)+)E", name)
        if m:
            args = re.findall(r"L([ib])(\d+)E", m.group(2))
            name = "%s<%s>(...)" % (m.group(1), ", ".join(a if t == "i" else ("true" if a == "1" else "false") for t, a in args))
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name[:110]


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                      "max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by name "
                      "order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    print("# rocprofv3 --kernel-trace --stats summary (durations in us; %d dispatches, %.3f ms total GPU kernel time)"
          % (sum(r[1] for r in rows), total / 1e6))
    print("%-112s %7s %12s %10s %10s %10s %6s %5s %5s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us",
                                                              "pct", "vgpr", "agpr", "lds"))
    for r in rows[:top]:
        print("%-112s %7d %12.1f %10.1f %10.1f %10.1f %6.2f %5d %5d %7d" % (
            short(r[0]), r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total, r[6] or 0, r[7] or 0, r[8] or 0))


if __name__ == "__main__":
    main()
