#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace --stats` on ROCm 7.2)
as a per-kernel CSV (name, calls, total / average / min / max duration in ns, share) -- the same columns as
rocprofv3's kernel_stats.csv.  usage: rocpd_stats.py results.db [out.csv] [--after-frac F]"""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    return name[:150]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    rows = list(cur.execute("select %s, start, end from kernels order by start" % namecol))
    frac = 0.0
    if "--after-frac" in sys.argv:
        frac = float(sys.argv[sys.argv.index("--after-frac") + 1])
    rows = rows[int(len(rows) * frac):]
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
