#!/usr/bin/env python3
"""Timeline view of a rocprofv3 rocpd database: per queue/stream busy time, the union of busy intervals, and
where the device sat idle on the busiest queue (gap histogram + the kernels around the largest gaps) over the
last `--after-frac` part of the trace (the timed steps).
usage: rocpd_timeline.py results.db [--after-frac F] [--to-frac F] [--steps K | --step-marker KERNEL]"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*\)$", "", name)
    return name[:70]


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    print("columns:", cols)
    namecol = "name" if "name" in cols else "kernel_name"
    qcol = next((c for c in ("stream_id", "stream", "queue_id", "queue") if c in cols), None)
    rows = list(cur.execute("select %s, start, end, %s from kernels order by start" % (namecol, qcol or "0")))
    frac = float(sys.argv[sys.argv.index("--after-frac") + 1]) if "--after-frac" in sys.argv else 0.0
    to = float(sys.argv[sys.argv.index("--to-frac") + 1]) if "--to-frac" in sys.argv else 1.0
    rows = rows[int(len(rows) * frac):int(len(rows) * to)]
    marker = sys.argv[sys.argv.index("--step-marker") + 1] if "--step-marker" in sys.argv else None
    steps = float(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 1.0
    if marker:      # a kernel launched exactly once per step
        steps = max(1, sum(1 for r in rows if marker in r[0]))
    span = rows[-1][2] - rows[0][1]
    per_q = defaultdict(list)
    for n, s, e, q in rows:
        per_q[q].append((s, e, n))
    print("span %.3f ms, %d kernels, %.3f ms/step over %.1f steps" % (span / 1e6, len(rows), span / 1e6 / steps, steps))
    for q, v in sorted(per_q.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
        busy = sum(e - s for s, e, _ in v)
        print("  %s=%s: %6d kernels busy %9.3f ms (%.1f %% of span), %.3f ms/step" %
              (qcol, q, len(v), busy / 1e6, 100.0 * busy / span, busy / 1e6 / steps))
    if "--per-stream" in sys.argv:      # what each stream's busy time is made of
        for q, v in sorted(per_q.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
            by = defaultdict(lambda: [0, 0])
            for s_, e_, n_ in v:
                a = by[short(n_)]
                a[0] += 1
                a[1] += e_ - s_
            print("  kernels of %s=%s (ms/step, launches/step):" % (qcol, q))
            for n_, a in sorted(by.items(), key=lambda kv: -kv[1][1])[:18]:
                print("    %8.3f %7.1f  %s" % (a[1] / 1e6 / steps, a[0] / steps, n_))
    # union of busy intervals
    iv = sorted((s, e) for _, s, e, _ in rows)
    u, cs, ce = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce:
            u += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    u += ce - cs
    print("union busy %.3f ms (%.1f %% of span): device idle %.3f ms/step" %
          (u / 1e6, 100.0 * u / span, (span - u) / 1e6 / steps))
    # gaps on the busiest queue
    q, v = max(per_q.items(), key=lambda kv: sum(e - s for s, e, _ in kv[1]))
    v.sort()
    gaps = [(v[i + 1][0] - v[i][1], i) for i in range(len(v) - 1)]
    hist = defaultdict(lambda: [0, 0])
    for g, _ in gaps:
        if g <= 0:
            continue
        b = "<2us" if g < 2000 else "2-5us" if g < 5000 else "5-10us" if g < 10000 else "10-20us" if g < 20000 else \
            "20-50us" if g < 50000 else "50-200us" if g < 200000 else ">=200us"
        hist[b][0] += 1
        hist[b][1] += g
    print("gaps on %s=%s (between consecutive kernels):" % (qcol, q))
    for b in ("<2us", "2-5us", "5-10us", "10-20us", "20-50us", "50-200us", ">=200us"):
        if b in hist:
            print("  %-9s %6d gaps %9.3f ms/step" % (b, hist[b][0], hist[b][1] / 1e6 / steps))
    # gap time attributed to the kernel that FOLLOWS (what the queue was waiting to launch)
    by_next = defaultdict(lambda: [0, 0])
    for g, i in gaps:
        if g > 0:
            a = by_next[short(v[i + 1][2])]
            a[0] += 1
            a[1] += g
    print("gap time by following kernel (top 25):")
    for n, a in sorted(by_next.items(), key=lambda kv: -kv[1][1])[:25]:
        print("  %8.3f ms/step %6d  %s" % (a[1] / 1e6 / steps, a[0], n))
    print("largest gaps:")
    for g, i in sorted(gaps, reverse=True)[:12]:
        print("  %8.1f us after %-50s before %s" % (g / 1e3, short(v[i][2])[:50], short(v[i + 1][2])[:50]))


if __name__ == "__main__":
    main()
