#!/usr/bin/env python3
"""Turns a rocprofv3 rocpd database (…_results.db) into the per-kernel stats table committed under
profiles/.   usage: python tools/rocprof_summary.py gpurun_out/prof/r01_results.db > profiles/<name>.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
print("%-58s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for name, calls, total, avg, pct in rows:
    short = name.split("(")[0]
    print("%-58s %8d %14.1f %12.2f %6.2f%%" % (short, calls, total, avg, pct))
