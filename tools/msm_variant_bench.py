#!/usr/bin/env python3
"""Times 768 MSMs of 2^11 points (c = 10, auto segments) with whichever library PLONK_HIP_LIB points to."""
import ctypes, json, os, random, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from plonkathon_amd import Context, Setup, set_context
from plonkathon_amd._lib import check
ctx = Context(0); set_context(ctx)
L, H = ctx.L, ctx.handle
setup = Setup.from_file(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "srs_2048.ptau"))
bases = setup.device_bases()
rng = random.Random(1)
n, M = 2048, 768
src = ctx.upload_ints([rng.randrange(1 << 253) for _ in range(4096)])
sc = ctx.alloc(n * M)
for off in range(0, n * M, 4096):
    check(L.plonk_mem_d2d(H, sc.at(off), src.ptr, 32 * 4096))
xy, fl = ctypes.create_string_buffer(64 * M), ctypes.create_string_buffer(M)
call = lambda: check(L.plonk_g1_msm(H, bases.handle, sc.ptr, n, M, n, xy, fl))
call(); ctx.sync()
ctx.profile_reset(); ctx.profile(True)
for _ in range(5):
    call()
acc, launches, _ = ctx.profile_read("msm_accumulate")
red, _, _ = ctx.profile_read("msm_bucket_reduce")
print(json.dumps({"lib": os.path.basename(os.environ.get("PLONK_HIP_LIB", "default")), "accumulate_ms": acc / launches, "bucket_reduce_ms": red / launches,
                  "xy0": xy.raw[:8].hex()}))
