#!/usr/bin/env python3
"""Batched MSM rate against the method and its window size: 1152 MSMs of 2^11 scalars per call — the mean
launch of a lock-step batch of 512 proofs — over the SRS slice, for
  bucket   the bucket method (Pippenger; no lookup table: what arbitrary bases get) at c = 9 .. 13 signed-digit windows,
  windows  the window tables at c = 8 .. 14 (0.4 .. 20 GB), and c = 16, 17 (68.7, 128.8 GB) with `big`,
  comb     the comb tables at h = 10 .. 20 teeth (67 MB .. 68.7 GB; csrc/msm_comb.h).
One JSON line per row: ms per call (best of 3, HIP events), MSMs/s, the proofs/s nine such MSMs per proof would allow, and the
kernels' own times from the library's events."""
import ctypes
import json
import os
import random
import sys

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)
from plonkathon_amd import Context, Setup, set_context  # noqa: E402
from plonkathon_amd._lib import check  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1152
WHAT = sys.argv[2].split(",") if len(sys.argv) > 2 else ["bucket", "windows", "comb"]
BIG = "big" in sys.argv[3:]
n = 2048
ctx = Context(0)
set_context(ctx)
L, H = ctx.L, ctx.handle
rng = random.Random(1)
src = ctx.upload_ints([rng.randrange(1 << 253) for _ in range(4096)])
total = ((n + 1) * M + 4095) // 4096 * 4096
sc = ctx.alloc(total)
for off in range(0, total, 4096):
    check(L.plonk_mem_d2d(H, sc.at(off), src.ptr, 32 * 4096))
xy, fl = ctypes.create_string_buffer(64 * M), ctypes.create_string_buffer(M)
PTAU = os.path.join(REPO, "tests", "golden", "srs_2048.ptau")


def run(tag, mode, c, groups=0):
    ctx.msm_lookup(mode, c if mode == 2 else 0, 0, windows=(tag == "windows"))
    check(L.plonk_msm_configure(H, c if mode == 1 else 0, groups))
    setup = Setup.from_file(PTAU)
    bases = setup.device_bases()
    call = lambda: check(L.plonk_g1_msm(H, bases.handle, sc.ptr, n, M, n + 1, xy, fl))
    call()
    ctx.sync()
    ctx.profile_reset()
    ctx.profile(True)
    best = None
    for _ in range(3):
        ctx.timer_start()
        call()
        ms = ctx.timer_stop_ms()
        best = ms if best is None or ms < best else best
    ctx.profile(False)
    info = bases.lookup_info()
    row = {"method": tag, "c": c, "additions_per_base": info["additions_per_base"], "groups": groups, "msms": M, "ms": best, "msms_per_s": M / (best * 1e-3), "proofs_per_s_at_9_msms": M / 9.0 / (best * 1e-3),
           "table_bytes": info["bytes"], "table_build_s": info["build_s"], "xy0": xy.raw[:8].hex()}
    for k in ("msm_digits", "msm_comb", "msm_lookup", "msm_sort", "msm_accumulate", "msm_bucket_reduce"):
        t, cnt, _ = ctx.profile_read(k)
        if cnt:
            row[k + "_ms"] = t / cnt
    print(json.dumps(row), flush=True)
    del bases, setup


if "bucket" in WHAT:
    for c in (9, 10, 11, 12, 13):
        for g in ((0,) if c != 10 else (0, 1, 2)):
            run("bucket", 1, c, g)
if "windows" in WHAT:
    for c in (8, 10, 11, 12, 13, 14) + ((16, 17) if BIG else ()):
        run("windows", 2, c)
if "comb20" in WHAT:
    run("comb", 2, 20)
    run("windows", 2, 17)
if "comb" in WHAT:
    for h in (10, 12, 13, 14, 15, 16, 17) + ((19, 20) if BIG else ()):
        run("comb", 2, h)
ctx.msm_lookup(0)
