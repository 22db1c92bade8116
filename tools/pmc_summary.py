#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_collect.sh) into profiles/<name>.json.

HBM traffic per launch = FETCH_SIZE * 2 * 1024 + WRITE_SIZE * 1024 bytes: on gfx950 FETCH_SIZE counts 128-byte
requests as 64 B (MI355X_MICROARCH.md §HBM); the factor is calibrated here on our own kernels — a 2^20-point NTT
pass must read its 32 MiB tile set exactly once, and reports 16.3 MiB.  WRITE_SIZE needs no correction (the same
pass writes 32 MiB and reports 32.0).  usage: python tools/pmc_summary.py gpurun_out/pmc profiles/r02_pmc_summary.json"""
import collections
import csv
import json
import os
import sys

src, out = sys.argv[1], sys.argv[2]


def load(name):
    agg = collections.defaultdict(lambda: [0, 0.0])
    path = os.path.join(src, name, "p_counter_collection.csv")
    for r in csv.DictReader(open(path)):
        k = (r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]))
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return {k: (n, v / n) for k, (n, v) in agg.items()}


res = {"units": "bytes per launch; traffic = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (see tools/pmc_summary.py)", "kernels": []}
for tag, algo in (("ntt", None), ("bench", None)):
    f, w = load(tag + "_FETCH_SIZE"), load(tag + "_WRITE_SIZE")
    for k in sorted(set(f) | set(w), key=lambda k: -(f.get(k, (0, 0))[1] + w.get(k, (0, 0))[1])):
        name, grid = k
        if name.startswith("__amd") or (f.get(k, (0, 0))[1] + w.get(k, (0, 0))[1]) < 1000:
            continue
        fk, wk = f.get(k, (0, 0.0)), w.get(k, (0, 0.0))
        res["kernels"].append({
            "run": tag, "kernel": name, "grid_threads": grid, "launches": max(fk[0], wk[0]),
            "fetch_size_kb_raw": round(fk[1], 1), "write_size_kb": round(wk[1], 1),
            "traffic_bytes": int(2 * fk[1] * 1024 + wk[1] * 1024),
        })
# the standalone 2^20 transform as a whole: tools/ntt_only.py runs NTT_REPS of them and no other NTT
NTT_REPS = 4
ntt_total = sum(k["traffic_bytes"] * k["launches"] for k in res["kernels"] if k["run"] == "ntt" and k["kernel"].find("ntt_") >= 0)
if ntt_total:
    res["kernels"].append({"run": "ntt", "kernel": "ntt_2^20", "grid_threads": 0, "launches": NTT_REPS,
                           "fetch_size_kb_raw": 0, "write_size_kb": 0, "traffic_bytes": int(ntt_total / NTT_REPS),
                           "note": "all pass kernels of one 2^20-point transform (algorithmic 64 MiB)"})
json.dump(res, open(out, "w"), indent=1)
for k in res["kernels"]:
    print("%-6s %-30s grid=%-9d n=%-3d fetch_raw=%10.0f KB write=%10.0f KB traffic=%8.1f MiB" % (
        k["run"], k["kernel"][:30], k["grid_threads"], k["launches"], k["fetch_size_kb_raw"], k["write_size_kb"], k["traffic_bytes"] / 2**20))
