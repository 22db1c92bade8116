#!/usr/bin/env python3
"""Runs NTT_REPS (4) standalone forward NTTs of 2^20 points and nothing else (for the rocprofv3 --pmc passes:
tools/pmc_summary.py divides the traffic of every ntt kernel of this run by NTT_REPS)."""
import os, random, sys
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)
from plonkathon_amd import Context, set_context
from plonkathon_amd._lib import check
ctx = Context(0); set_context(ctx); L, H = ctx.L, ctx.handle
rng = random.Random(1)
def fill(n):
    per = min(n, 4096)
    src = ctx.upload_ints([rng.randrange(1 << 253) for _ in range(per)])
    buf = ctx.alloc(n)
    for off in range(0, n, per):
        check(L.plonk_mem_d2d(H, buf.at(off), src.ptr, 32 * min(per, n - off)))
    return buf
for log_n, batch, reps in ((20, 1, 4),):
    n = 1 << log_n
    buf, out = fill(n * batch), ctx.alloc(n * batch)
    # (no warm-up launch: every ntt kernel launch of this process belongs to one of the `reps` transforms)
    for _ in range(reps):
        check(L.plonk_fr_ntt(H, buf.ptr, out.ptr, log_n, 0, batch))
    ctx.sync()
print("done")
