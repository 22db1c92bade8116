#!/usr/bin/env python3
"""NTT microbench (SURVEY.md 8(d) / BASELINE configs[3]): one JSON line per (size, batch, split, kernel kind, direction,
in-place) with the best of `--reps` timed launches (HIP events on the library's stream), G elem/s and the HBM-roofline
fraction on the algorithmic 64 N bytes.

  python tools/ntt_sweep.py                         # the default sweep: batched 2^8..2^13, lone 2^14..2^24, LDS-kernel baselines
  python tools/ntt_sweep.py --shapes 20:1:10,20:1:11  # log_n:batch[:log_r1 of the two-pass split]
  PLONK_HIP_LIB=plonkathon_amd/libplonk_hip_alt.so python tools/ntt_sweep.py --tag alt   # an alternative build"""
import argparse
import json
import os
import random
import sys

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)
from plonkathon_amd import Context, set_context  # noqa: E402
from plonkathon_amd._lib import check  # noqa: E402

HBM_PEAK = 8.0e12

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="")
ap.add_argument("--kinds", default="0")
ap.add_argument("--reps", type=int, default=7)
ap.add_argument("--tag", default="")
ap.add_argument("--directions", default="fwd")      # fwd,inv
ap.add_argument("--placements", default="out")      # out,in
ap.add_argument("--table-gb", type=float, default=-1.0)  # budget of the full inter-pass twiddle tables (0 = two small tables; default: the library's 4 GiB)
ap.add_argument("--field", default="bn254")         # bn254 | bls12_381 (plonk_bls_fr_ntt: 2^8 .. 2^26)
args = ap.parse_args()

ctx = Context(0)
set_context(ctx)
L, H = ctx.L, ctx.handle
NTT = L.plonk_bls_fr_ntt if args.field == "bls12_381" else L.plonk_fr_ntt
if args.table_gb >= 0:
    check(L.plonk_ntt_set_table_budget(H, int(args.table_gb * (1 << 30))))
rng = random.Random(1)
src = ctx.upload_ints([rng.randrange(1 << 253) for _ in range(4096)])


def fill(n):
    buf = ctx.alloc(n)
    per = 4096
    for off in range(0, n, per):
        check(L.plonk_mem_d2d(H, buf.at(off), src.ptr, 32 * min(per, n - off)))
    return buf


if args.shapes:
    shapes = []
    for s in args.shapes.split(","):
        f = [int(x) for x in s.split(":")]
        shapes.append((f[0], f[1], f[2] if len(f) > 2 else 0))
else:
    shapes = [(k, (1 << 22) >> k, 0) for k in range(8, 14)] + [(11, 512, 0), (13, 512, 0), (10, 512, 0), (12, 512, 0)]
    shapes += [(k, 1, 0) for k in range(14, 25)]
    shapes += [(16, 256, 0), (18, 64, 0), (20, 16, 0), (22, 4, 0)]  # constant work: 2^24 elements

for kind in [int(k) for k in args.kinds.split(",")]:
    check(L.plonk_ntt_select_kernel(H, kind))
    for log_n, batch, split in shapes:
        n = 1 << log_n
        if split:
            check(L.plonk_ntt_set_split(H, log_n, split))
        buf = fill(n * batch)
        dst = ctx.alloc(n * batch)
        for direction in args.directions.split(","):
            inv = 1 if direction == "inv" else 0
            for place in args.placements.split(","):
                o = buf if place == "in" else dst
                for _ in range(2):
                    check(NTT(H, buf.ptr, o.ptr, log_n, inv, batch))
                ctx.sync()
                times = []
                for _ in range(args.reps):
                    ctx.timer_start()
                    check(NTT(H, buf.ptr, o.ptr, log_n, inv, batch))
                    times.append(ctx.timer_stop_ms())
                best = min(times)
                print(json.dumps({"what": "ntt", "field": args.field, "tag": args.tag, "kind": kind, "log_n": log_n, "batch": batch, "split": split,
                                  "dir": direction, "place": place, "ms": round(best, 5), "ms_median": round(sorted(times)[len(times) // 2], 5),
                                  "Gelem_s": round(n * batch / best / 1e6, 3),
                                  "hbm_frac": round(64.0 * n * batch / (best * 1e-3) / HBM_PEAK, 4)}), flush=True)
        if split:
            check(L.plonk_ntt_set_split(H, log_n, 0))
        del buf, dst
