#!/usr/bin/env python3
"""Turn a rocprofv3 (rocpd sqlite) kernel trace into the text summary committed under profiles/.
usage: rocpd_summary.py <results.db> [--top N] [--between-markers [STEPS]]
--between-markers: only the dispatches between the first and the last ddmp_trace_marker_kernel (bench.py launches one
right before and one right after its timed region): the statistics of exactly the timed iterations -- no priming
iteration, no set-up.  With STEPS the table also shows microseconds per step."""
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
    where, note, steps = "", "", 0
    if "--between-markers" in sys.argv:
        i = sys.argv.index("--between-markers")
        if i + 1 < len(sys.argv) and sys.argv[i + 1].isdigit():
            steps = int(sys.argv[i + 1])
        cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
        t0c, t1c = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
        marks = db.execute("select %s, %s from kernels where name like '%%ddmp_trace_marker_kernel%%' order by 1" % (t0c, t1c)).fetchall()
        if len(marks) < 2:
            raise SystemExit("fewer than two ddmp_trace_marker_kernel dispatches in the trace")
        where = " where %s > %d and %s < %d" % (t0c, marks[0][1], t1c, marks[-1][0])
        note = " between the two trace markers of bench.py = its timed region (%.3f ms wall%s)" % (
            (marks[-1][0] - marks[0][1]) / 1e6, ", %d steps" % steps if steps else "")
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                      "max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels" + where + " group by name "
                      "order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    print("# rocprofv3 --kernel-trace --stats summary (durations in us; %d dispatches, %.3f ms total GPU kernel time)%s"
          % (sum(r[1] for r in rows), total / 1e6, note))
    print("%-112s %7s %12s %10s %10s %10s %6s %5s %5s %7s%s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us",
                                                                "pct", "vgpr", "agpr", "lds", "  us/step" if steps else ""))
    for r in rows[:top]:
        print("%-112s %7d %12.1f %10.1f %10.1f %10.1f %6.2f %5d %5d %7d%s" % (
            short(r[0]), r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total, r[6] or 0, r[7] or 0, r[8] or 0,
            "  %8.1f" % (r[2] / 1e3 / steps) if steps else ""))


if __name__ == "__main__":
    main()
