#!/usr/bin/env python3
"""Turns a rocprofv3 rocpd database (…_results.db) into the per-kernel stats table committed under
profiles/.   usage: python tools/rocprof_summary.py gpurun_out/prof/r01_results.db > profiles/<name>.txt
Columns: calls, total, mean, MEDIAN and minimum duration per launch (us).  The mean of a kernel whose first launch queued behind a
one-off table build carries that wait; the median does not."""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
per = collections.defaultdict(list)
for name, dur in db.execute("select name, end - start from kernels"):
    per[name.split("(")[0].replace("void ", "")].append(dur / 1e3)
total_all = sum(sum(v) for v in per.values())
print("%-58s %7s %13s %11s %11s %11s %7s" % ("kernel", "calls", "total_us", "mean_us", "median_us", "min_us", "pct"))
for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print("%-58s %7d %13.1f %11.2f %11.2f %11.2f %6.2f%%" % (name, len(v), sum(v), sum(v) / len(v), v[len(v) // 2], v[0], 100.0 * sum(v) / total_all))
