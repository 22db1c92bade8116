#!/usr/bin/env python3
"""Standalone NTT launches at the prover's own shapes (for rocprofv3 --pmc utilisation passes)."""
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
kind = int(sys.argv[1]) if len(sys.argv) > 1 else 0
check(L.plonk_ntt_select_kernel(H, kind))
for log_n, batch, reps in ((8, 16384, 3), (9, 8192, 3), (10, 4096, 3), (11, 2048, 3), (12, 1024, 3), (13, 512, 3), (20, 16, 2)):
    n = 1 << log_n
    buf, out = fill(n * batch), ctx.alloc(n * batch)
    for _ in range(reps):
        check(L.plonk_fr_ntt(H, buf.ptr, out.ptr, log_n, 0, batch))
    ctx.sync()
print("done")
