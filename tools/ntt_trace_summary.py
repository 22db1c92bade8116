#!/usr/bin/env python3
"""Per-pass kernel durations of the lone transforms of tools/ntt_only.py from a rocprofv3 --kernel-trace run
(the `ntt_trace` step of tools/gpu_session.sh) -> the table committed as profiles/rNN_ntt_kernel_stats.txt.
   usage: python tools/ntt_trace_summary.py <rocprof output dir> <log of ntt_only.py (its last line is the plan)>"""
import csv
import glob
import json
import os
import sys

src, log = sys.argv[1], sys.argv[2]
plan = [json.loads(l) for l in open(log) if l.startswith("{")][-1]
paths = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
rows = list(csv.DictReader(open(paths[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k = [(r["Kernel_Name"].split("(")[0].replace("void ", ""), int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0)),
      int(r["Workgroup_Size_X"]) if "Workgroup_Size_X" in r else int(r.get("Workgroup_Size", 0)),
      int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if plan["kernel_substring"] in r["Kernel_Name"]]
print("# rocprofv3 --kernel-trace of tools/ntt_only.py: lone forward transforms, BN254 Fr, no warm-up launches (rep 0 of a size is cold)")
print("# HBM-roofline fraction = 64 N bytes / (sum of the passes' durations) / 8 TB/s; gap = idle time between the two dependent launches")
print("%-6s %-4s %-44s %10s %6s %10s %10s" % ("size", "pass", "kernel", "grid", "wg", "min_us", "median_us"))
pos = 0
for log_n, reps, launches in plan["plan"]:
    per_pass = [[] for _ in range(launches)]
    gaps, spans, first = [], [], None
    for r in range(reps):
        ks = k[pos:pos + launches]
        pos += launches
        if len(ks) < launches:
            break
        first = first or ks
        for i, (_, _, _, a, b) in enumerate(ks):
            per_pass[i].append((b - a) / 1e3)
        if launches > 1:
            gaps.append((ks[1][3] - ks[0][4]) / 1e3)
        spans.append((ks[-1][4] - ks[0][3]) / 1e3)
    tot_min = 0.0
    for i in range(launches):
        ts = sorted(per_pass[i])
        if not ts:
            continue
        name, grid, wg = first[i][0], first[i][1], first[i][2]
        tot_min += ts[0]
        print("2^%-4d %-4d %-44s %10d %6d %10.2f %10.2f" % (log_n, i, name[:44], grid, wg, ts[0], ts[len(ts) // 2]))
    n = 1 << log_n
    if spans:
        print("2^%-4d sum of passes (min) %.2f us -> %.1f GB/s = %.2f %% of 8 TB/s; first launch start -> last end (min) %.2f us; gap between the launches (median) %.2f us"
              % (log_n, tot_min, 64.0 * n / (tot_min * 1e-6) / 1e9, 64.0 * n / (tot_min * 1e-6) / 1e9 / 80.0, min(spans),
                 sorted(gaps)[len(gaps) // 2] if gaps else 0.0))
