#!/usr/bin/env python3
"""Runs REPS lone forward transforms of each size in SHAPES and nothing else that launches an NTT kernel — the subject of
the rocprofv3 --pmc passes of tools/pmc_collect.sh.  No warm-up launches: every wave-kernel dispatch of this process
belongs to one of the listed transforms, in order (tools/pmc_summary.py splits the counter rows by this plan, which is
also printed as one JSON line)."""
import json
import os
import random
import sys

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)
from plonkathon_amd import Context, set_context  # noqa: E402
from plonkathon_amd._lib import check  # noqa: E402

SHAPES = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "16,18,20,22,24").split(",")]
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = Context(0)
set_context(ctx)
L, H = ctx.L, ctx.handle
rng = random.Random(1)
src = ctx.upload_ints([rng.randrange(1 << 253) for _ in range(4096)])
for log_n in SHAPES:
    n = 1 << log_n
    buf, out = ctx.alloc(n), ctx.alloc(n)
    for off in range(0, n, 4096):
        check(L.plonk_mem_d2d(H, buf.at(off), src.ptr, 32 * 4096))
    for _ in range(REPS):
        check(L.plonk_fr_ntt(H, buf.ptr, out.ptr, log_n, 0, 1))
    ctx.sync()
    del buf, out
print(json.dumps({"plan": [[log_n, REPS, 2] for log_n in SHAPES], "kernel_substring": "ntt_wavel_"}))
