#!/usr/bin/env python3
"""an MSM whose result is the identity (all scalars zero) must not cost more than any other MSM on the comb tables.
Times 512 MSMs of 2^11 zero scalars against 512 of random scalars (one plonk_g1_msm call each, best of 5) -> one JSON line."""
import ctypes
import json
import os
import random
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
from plonkathon_amd import Setup, get_context  # noqa: E402
from plonkathon_amd._lib import check  # noqa: E402
from plonkathon_amd.field import R_MOD  # noqa: E402

ctx = get_context()
n, M = 2048, 512
setup = Setup.from_file(os.path.join(REPO, "tests", "golden", "srs_2048.ptau"))
bases = setup.device_bases(ctx)
rng = random.Random(1)
rnd = ctx.upload_ints([rng.randrange(R_MOD) for _ in range(n)])
bufs = {"zero": ctx.alloc(n * M), "random": ctx.alloc(n * M)}
check(ctx.L.plonk_mem_zero(ctx.handle, bufs["zero"].ptr, 32 * n * M))
for m in range(M):
    check(ctx.L.plonk_mem_d2d(ctx.handle, bufs["random"].at(m * n), rnd.ptr, 32 * n))
out = {}
xy, fl = ctypes.create_string_buffer(64 * M), ctypes.create_string_buffer(M)
for name, buf in bufs.items():
    best = 1e9
    for _ in range(5):
        t = time.perf_counter()
        check(ctx.L.plonk_g1_msm(ctx.handle, bases.handle, buf.ptr, n, M, n, xy, fl))
        best = min(best, time.perf_counter() - t)
    out[name + "_ms"] = round(1e3 * best, 3)
    out[name + "_identity_flags"] = sum(fl.raw)
out["table"] = bases.lookup_info()
assert out["zero_identity_flags"] == M and out["random_identity_flags"] == 0
print(json.dumps(out))
