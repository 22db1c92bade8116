#!/usr/bin/env python3
"""NTT microbench per kernel kind (0 auto, 1 radix-2 stages, 2 Stockham, 3 wave on packed residues, 5 wave on signed
limbs): ms and G elem/s at several shapes.  NTT_KINDS=3,5 selects the kinds."""
import json, os, random, sys
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
shapes = [(9, 4096), (11, 512), (11, 2048), (13, 512), (13, 2048), (18, 1), (18, 16), (20, 1), (20, 8), (22, 1)]
out = {}
for kind in [int(k) for k in os.environ.get('NTT_KINDS', '3,5').split(',')]:
    check(L.plonk_ntt_select_kernel(H, kind))
    for log_n, batch in shapes:
        n = 1 << log_n
        buf, dst = fill(n * batch), ctx.alloc(n * batch)
        check(L.plonk_fr_ntt(H, buf.ptr, dst.ptr, log_n, 0, batch)); ctx.sync()
        best = 1e9
        for _ in range(5):
            ctx.timer_start(); check(L.plonk_fr_ntt(H, buf.ptr, dst.ptr, log_n, 0, batch)); best = min(best, ctx.timer_stop_ms())
        out["kind%d_2^%d_x%d" % (kind, log_n, batch)] = {"ms": round(best, 4), "Gelem_s": round(n * batch / best / 1e6, 2)}
        del buf, dst
print(json.dumps(out))
