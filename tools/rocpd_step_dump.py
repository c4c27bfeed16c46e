#!/usr/bin/env python3
"""One step of a rocprofv3 kernel trace (rocpd database), launch by launch: for the stream that carries the network's
passes, every kernel of ONE step in start order with its grid, duration and the gap in front of it -- the dependent
chain a step is made of -- followed by per-kernel sums for that step and for the launches at or below `--small-us`.
usage: rocpd_step_dump.py results.db [--marker KERNEL] [--step K] [--small-us U] [--all-streams]
The marker is a kernel launched exactly once per step (default k_pack_weights_jobs); step K counts from the end (1 =
last complete step)."""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    name = name.replace("aabr::", "")
    return name[:64]


def arg(flag, default):
    return sys.argv[sys.argv.index(flag) + 1] if flag in sys.argv else default


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    rows = list(cur.execute("select %s, start, end, stream_id, grid_x, grid_y, grid_z, workgroup_x from kernels order by start"
                            % namecol))
    marker = arg("--marker", "k_pack_weights_jobs")
    k = int(arg("--step", "2"))
    small = float(arg("--small-us", "25")) * 1e3
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(marks) < k + 1:
        sys.exit("marker %s seen %d times" % (marker, len(marks)))
    lo, hi = marks[-k - 1], marks[-k]
    t0, t1 = rows[lo][1], rows[hi][1]
    step = [r for r in rows if t0 <= r[1] < t1]
    print("step of %.3f ms (marker to marker), %d launches on all streams" % ((t1 - t0) / 1e6, len(step)))
    per = defaultdict(list)
    for r in step:
        per[r[3]].append(r)
    main_q = rows[lo][3]
    for q, v in sorted(per.items(), key=lambda kv: -sum(r[2] - r[1] for r in kv[1])):
        busy = sum(r[2] - r[1] for r in v)
        print("stream %s: %4d launches, busy %8.3f ms%s" % (q, len(v), busy / 1e6, "  <- marker's stream" if q == main_q else ""))
    for q in (sorted(per) if "--all-streams" in sys.argv else [main_q]):
        v = per[q]
        print("\n== stream %s, in start order: t(us from step start)  dur(us)  gap(us)  workgroups  kernel" % q)
        prev = None
        for n, s, e, _, gx, gy, gz, wx in v:
            wg = (max(gx, 1) * max(gy, 1) * max(gz, 1)) // max(wx, 1)
            gap = (s - prev) / 1e3 if prev is not None else 0.0
            print("%9.1f %8.1f %7.1f %7d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, wg, short(n)))
            prev = e
        by = defaultdict(lambda: [0, 0, 0, 0])
        gaps = 0
        prev = None
        for n, s, e, *_ in v:
            a = by[short(n)]
            a[0] += 1
            a[1] += e - s
            if e - s <= small:
                a[2] += 1
                a[3] += e - s
            if prev is not None and s > prev:
                gaps += s - prev
            prev = max(prev or 0, e)
        print("\n-- stream %s per kernel: launches, ms | launches <= %.0f us, ms" % (q, small / 1e3))
        for n, a in sorted(by.items(), key=lambda kv: -kv[1][1]):
            print("%5d %8.3f | %5d %8.3f  %s" % (a[0], a[1] / 1e6, a[2], a[3] / 1e6, n))
        print("total %d launches %.3f ms busy, %.3f ms of gaps; small launches: %d, %.3f ms" %
              (len(v), sum(a[1] for a in by.values()) / 1e6, gaps / 1e6, sum(a[2] for a in by.values()),
               sum(a[3] for a in by.values()) / 1e6))


if __name__ == "__main__":
    main()
