#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace --stats` on ROCm 7.2)
as a per-kernel CSV (name, calls, total / average / min / max duration in ns, share) -- the same columns as
rocprofv3's kernel_stats.csv.  usage: rocpd_stats.py results.db [out.csv] [--after-frac F] [--last N] [--by-grid]
--last N: only the last N kernel launches of the trace (a tool that ends with N launches of ONE instance -- tools/
tools_dominant_instance.py -- gives that instance a row of its own, whatever else shares its kernel name and grid).
--by-grid: one row per (kernel, grid size in threads) -- `name|grid=N`, the key bench.py's roofline object and the PMC
summaries use -- so that the average duration of ONE instance (e.g. the dominant convolution launch) can be read from the
committed profile instead of from the kernel's average over all its grids."""
import csv
import re
import sqlite3
import sys


def short(name):
    tail = ""
    if "|grid=" in name:
        name, tail = name.rsplit("|", 1)
        tail = "|" + tail
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    return name[:150] + tail


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    by_grid = "--by-grid" in sys.argv
    gcols = [c for c in ("grid_size_x", "grid_size_y", "grid_size_z") if c in cols] or \
            [c for c in ("grid_x", "grid_y", "grid_z") if c in cols]
    if by_grid and not gcols:
        sys.stderr.write("no grid-size columns in the kernels view (columns: %s)\n" % ", ".join(cols))
        by_grid = False
    if by_grid:
        rows = []
        for r in cur.execute("select %s, start, end, %s from kernels order by start" % (namecol, ", ".join(gcols))):
            g = 1
            for v in r[3:]:
                g *= max(int(v or 1), 1)
            rows.append((r[0] + "|grid=%d" % g, r[1], r[2]))
    else:
        rows = list(cur.execute("select %s, start, end from kernels order by start" % namecol))
    frac = 0.0
    if "--after-frac" in sys.argv:
        frac = float(sys.argv[sys.argv.index("--after-frac") + 1])
    rows = rows[int(len(rows) * frac):]
    if "--last" in sys.argv:
        rows = rows[-int(sys.argv[sys.argv.index("--last") + 1]):]
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(short(n), [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    out = sorted(agg.items(), key=lambda kv: -kv[1][1])
    w = csv.writer(open(sys.argv[2], "w") if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else sys.stdout)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for n, a in out:
        w.writerow([n, a[0], a[1], round(a[1] / a[0], 1), round(100.0 * a[1] / tot, 3), a[2], a[3]])
    if rows:
        span = rows[-1][2] - rows[0][1]
        sys.stderr.write("kernels: %d, busy %.3f ms of %.3f ms span (%.1f %%)\n" % (len(rows), tot / 1e6, span / 1e6,
                                                                                 100.0 * tot / span))


if __name__ == "__main__":
    main()
