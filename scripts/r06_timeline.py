#!/usr/bin/env python3
"""Where does a launch-bound iteration go?  From a rocprofv3 kernel trace (rocpd sqlite) of N identical iterations: per stream
(queue) the busy time (sum of kernel durations), the idle time between consecutive kernels, and the longest kernels; for the
last `--iters` iterations delimited by the adam_kernel dispatches.
usage: r06_timeline.py <results.db> [--tail-frac 0.5]"""
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    t0c, t1c = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
    qc = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    frac = float(sys.argv[sys.argv.index("--tail-frac") + 1]) if "--tail-frac" in sys.argv else 0.5
    rows = db.execute("select name, %s, %s, %s from kernels order by %s" % (t0c, t1c, qc or "0", t0c)).fetchall()
    if not rows:
        raise SystemExit("no kernels")
    a, b = rows[0][1], rows[-1][2]
    cut = a + (b - a) * (1.0 - frac)
    rows = [r for r in rows if r[1] >= cut]
    n_adam = sum(1 for r in rows if "adam_kernel" in r[0]) / 2.0            # two Adam launches per iteration
    span = (rows[-1][2] - rows[0][1]) / 1e3
    busy = sum(r[2] - r[1] for r in rows) / 1e3
    print("# last %.0f %% of the trace: %d dispatches, %.1f iterations (by adam_kernel count), span %.1f us = %.1f us/iteration, "
          "sum of kernel durations %.1f us = %.1f us/iteration" % (100 * frac, len(rows), n_adam, span, span / max(n_adam, 1), busy,
                                                                   busy / max(n_adam, 1)))
    # union of busy intervals (any queue): time with at least one kernel running
    ev = sorted([(r[1], 1) for r in rows] + [(r[2], -1) for r in rows])
    depth, last, covered, two = 0, ev[0][0], 0, 0
    for t, d in ev:
        if depth >= 1:
            covered += t - last
        if depth >= 2:
            two += t - last
        depth += d
        last = t
    print("# >= 1 kernel running %.1f us/iteration, >= 2 running %.1f us/iteration, nothing running %.1f us/iteration"
          % (covered / 1e3 / max(n_adam, 1), two / 1e3 / max(n_adam, 1), (span - covered / 1e3) / max(n_adam, 1)))
    by_q = defaultdict(list)
    for r in rows:
        by_q[r[3]].append(r)
    for q, rs in sorted(by_q.items(), key=lambda kv: -len(kv[1])):
        gaps = [rs[i + 1][1] - rs[i][2] for i in range(len(rs) - 1)]
        gaps_us = sorted(g / 1e3 for g in gaps)
        if not gaps_us:
            continue
        med = gaps_us[len(gaps_us) // 2]
        print("queue %s: %d kernels, busy %.1f us/it, gaps: median %.2f us, mean %.2f us, p90 %.2f us, sum %.1f us/it"
              % (q, len(rs), sum(r[2] - r[1] for r in rs) / 1e3 / max(n_adam, 1), med, sum(gaps_us) / len(gaps_us),
                 gaps_us[int(0.9 * len(gaps_us))], sum(gaps_us) / max(n_adam, 1)))
    hist = defaultdict(lambda: [0, 0.0])
    for r in rows:
        d = (r[2] - r[1]) / 1e3
        k = "<3us" if d < 3 else "3-6us" if d < 6 else "6-12us" if d < 12 else "12-25us" if d < 25 else "25-50us" if d < 50 else ">=50us"
        hist[k][0] += 1
        hist[k][1] += d
    for k in ("<3us", "3-6us", "6-12us", "12-25us", "25-50us", ">=50us"):
        print("  kernels of %-8s: %6.1f per iteration, %8.1f us per iteration" % (k, hist[k][0] / max(n_adam, 1), hist[k][1] / max(n_adam, 1)))


if __name__ == "__main__":
    main()
