#!/usr/bin/env python3
"""Randomised cross-check of the MSM schedules on the GPU: for random (number of scalars n <= 2^11, MSMs per call M, scalar stride,
workgroups per MSM, table layout — combs with and without top tables, window tables — and size) the table MSM must return the bytes the bucket method returns — scalars include 0, 1,
r - 1, repeated values and runs of equal scalars (equal pieces in a column's tree).  One JSON line per round; exit status 1 on a
mismatch.        python tools/msm_fuzz.py [rounds] [seed]"""
import ctypes
import json
import os
import random
import sys

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)
from plonkathon_amd import Context, Setup, set_context  # noqa: E402
from plonkathon_amd._lib import check  # noqa: E402

R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 2025)
ctx = Context(0)
set_context(ctx)
L, H = ctx.L, ctx.handle
PTAU = os.path.join(REPO, "tests", "golden", "srs_2048.ptau")
bad = 0
for rd in range(rounds):
    n = rng.choice([1, 2, 3, 17, 19, 20, 63, 64, 65, 255, 256, 257, 300, 1000, 2047, 2048, rng.randrange(1, 2049)])
    M = rng.choice([1, 2, 3, 9, 64, 65, 300])
    stride = n + rng.choice([0, 1, 7])
    kind = rng.choice(["comb", "comb", "comb_top", "comb_top", "windows"])  # comb_top: the comb with top tables (csrc/msm_comb.h)
    bits = (rng.choice([2, 3, 5, 8, 11, 13, 16, 17, 18]) if kind == "comb" else
            rng.choice([4, 6, 7, 9, 11, 12, 14, 14, 18]) if kind == "comb_top" else rng.choice([3, 6, 10, 13]))
    groups = rng.choice([0, 0, 1, 2, 3, 8, 64])
    special = [0, 1, 2, R_MOD - 1, R_MOD - 2, (R_MOD - 1) // 2, 1 << 253]
    vals = []
    for m in range(M):
        mode = rng.randrange(4)
        row = [rng.randrange(R_MOD) for _ in range(stride)]
        if mode == 1:
            row = [rng.choice(special) if rng.random() < 0.3 else v for v in row]
        elif mode == 2:
            row = [row[0]] * stride  # every scalar equal
        elif mode == 3:
            row = [0] * stride
        vals += row
    sc = ctx.upload_ints(vals)
    out = []
    for method in ("table", "bucket"):
        ctx.msm_lookup(2 if method == "table" else 1, bits if method == "table" else 0, 0, windows=(kind == "windows"),
                       top=(kind == "comb_top" and method == "table"))
        ctx.msm_configure(0, groups)
        setup = Setup.from_file(PTAU)
        bases = setup.device_bases()
        xy, fl = ctypes.create_string_buffer(64 * M), ctypes.create_string_buffer(M)
        check(L.plonk_g1_msm(H, bases.handle, sc.ptr, n, M, stride, xy, fl))
        out.append((xy.raw, fl.raw))
        del bases, setup
    ok = out[0] == out[1]
    bad += 0 if ok else 1
    print(json.dumps({"round": rd, "n": n, "msms": M, "stride": stride, "layout": kind, "bits": bits, "groups": groups, "equal": ok}), flush=True)
ctx.msm_lookup(0)
ctx.msm_configure(0, 0)
sys.exit(1 if bad else 0)
