#!/usr/bin/env python3
"""GPU tuning sweep (writes JSON lines to stdout): batched MSM time vs window bits / groups, batched NTT
times at the prover's sizes, standalone NTT sizes 2^16..2^24, BatchProver time vs batch size."""
import ctypes
import json
import os
import random
import sys
import time

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)
from plonkathon_amd import BatchProver, Context, Program, Setup, set_context  # noqa: E402
from plonkathon_amd._lib import check  # noqa: E402

ctx = Context(0)
set_context(ctx)
L, H = ctx.L, ctx.handle
PTAU = os.path.join(REPO, "tests", "golden", "srs_2048.ptau")
setup = Setup.from_file(PTAU)
bases = setup.device_bases()
rng = random.Random(1)


def fill(n_elems):
    per = min(n_elems, 4096)
    src = ctx.upload_ints([rng.randrange(1 << 253) for _ in range(per)])
    buf = ctx.alloc(n_elems)
    for off in range(0, n_elems, per):
        check(L.plonk_mem_d2d(H, buf.at(off), src.ptr, 32 * min(per, n_elems - off)))
    return buf


def timed(fn, reps=3):
    fn()
    ctx.sync()
    best = 1e30
    for _ in range(reps):
        ctx.timer_start()
        fn()
        best = min(best, ctx.timer_stop_ms())
    return best


which = sys.argv[1:] or ["msm", "ntt", "prover"]
if "msm" in which:
    n = 2048
    for M in (1, 64, 768):
        sc = fill(n * M)
        xy = ctypes.create_string_buffer(64 * M)
        fl = ctypes.create_string_buffer(M)
        for c in (9, 10, 11, 12, 13):
            for G in ((0, 1, 2, 4) if M == 768 else (0,) if M > 1 else (4, 8, 16)):
                check(L.plonk_msm_configure(H, c, G))
                t0 = time.perf_counter()
                check(L.plonk_g1_msm(H, bases.handle, sc.ptr, n, M, n, xy, fl))  # includes table (re)build
                first = time.perf_counter() - t0
                ctx.profile_reset(); ctx.profile(True)
                ms = timed(lambda: check(L.plonk_g1_msm(H, bases.handle, sc.ptr, n, M, n, xy, fl)))
                acc_ms, launches, _ = ctx.profile_read("msm_accumulate")
                sort_ms, _, _ = ctx.profile_read("msm_sort")
                red_ms, _, _ = ctx.profile_read("msm_bucket_reduce")
                ctx.profile(False)
                L_ = max(launches, 1)
                print(json.dumps({"what": "msm", "n": n, "M": M, "c": c, "G": G, "ms": ms, "ms_per_msm": ms / M,
                                  "sort_ms": sort_ms / L_, "accumulate_ms": acc_ms / L_, "bucket_reduce_ms": red_ms / L_,
                                  "first_call_s": first}), flush=True)
    check(L.plonk_msm_configure(H, 0, 0))
if "ntt" in which:
    for log_n, batch in ((11, 1), (11, 1024), (13, 1), (13, 1280), (16, 1), (16, 64), (18, 1), (20, 1), (22, 1), (24, 1)):
        n = 1 << log_n
        buf, out = fill(n * batch), ctx.alloc(n * batch)
        for inv in (0, 1):
            ms = timed(lambda: check(L.plonk_fr_ntt(H, buf.ptr, out.ptr, log_n, inv, batch)), reps=5)
            print(json.dumps({"what": "ntt", "log_n": log_n, "batch": batch, "inverse": inv, "ms": ms,
                              "gf_elems_per_s": n * batch / (ms * 1e-3), "algo_GBps": 64.0 * n * batch / (ms * 1e-3) / 1e9}), flush=True)
        del buf, out
    # tile / radix variants at 2^20
    n = 1 << 20
    buf, out = fill(n), ctx.alloc(n)
    for tile, single, radix in ((12, 11, 10), (11, 11, 10), (12, 11, 7), (11, 10, 7), (10, 10, 7), (12, 11, 8), (9, 9, 7)):
        check(L.plonk_ntt_configure(H, tile, single, radix))
        ms = timed(lambda: check(L.plonk_fr_ntt(H, buf.ptr, out.ptr, 20, 0, 1)), reps=5)
        print(json.dumps({"what": "ntt_cfg", "log_n": 20, "tile": tile, "single": single, "radix": radix, "ms": ms}), flush=True)
    check(L.plonk_ntt_configure(H, 0, 0, 0))
if "nttkind" in which:
    for kind in (0, 1, 6, 7):
        check(L.plonk_ntt_select_kernel(H, kind))
        for log_n, batch in ((11, 1024), (13, 1280), (16, 64), (20, 1), (24, 1)):
            n = 1 << log_n
            buf, out = fill(n * batch), ctx.alloc(n * batch)
            ms = timed(lambda: check(L.plonk_fr_ntt(H, buf.ptr, out.ptr, log_n, 0, batch)), reps=5)
            print(json.dumps({"what": "nttkind", "kind": kind, "log_n": log_n, "batch": batch, "ms": ms,
                              "gf_elems_per_s": n * batch / (ms * 1e-3)}), flush=True)
            del buf, out
    check(L.plonk_ntt_select_kernel(H, 0))
if "nttcfg" in which:
    for log_n in (20, 24):
        n = 1 << log_n
        buf, out = fill(n), ctx.alloc(n)
        for kind in (1,):
            check(L.plonk_ntt_select_kernel(H, kind))
            for tile, single, radix in ((12, 11, 10), (11, 11, 10), (11, 11, 7), (10, 10, 7), (12, 11, 8), (11, 11, 8), (12, 11, 7)):
                check(L.plonk_ntt_configure(H, tile, single, radix))
                ms = timed(lambda: check(L.plonk_fr_ntt(H, buf.ptr, out.ptr, log_n, 0, 1)), reps=5)
                print(json.dumps({"what": "nttcfg", "log_n": log_n, "kind": kind, "tile": tile, "radix": radix, "ms": ms,
                                  "gf_elems_per_s": n / (ms * 1e-3)}), flush=True)
        del buf, out
    check(L.plonk_ntt_configure(H, 0, 0, 0)); check(L.plonk_ntt_select_kernel(H, 0))
if "prover" in which:
    from bench import chain_program_lines, witness_for

    program = Program(chain_program_lines(2048), 2048)
    wits = [witness_for(program, i) for i in range(4)]
    ctxs = [ctx] + [Context(0) for _ in range(3)]
    provers = [BatchProver(setup, program, c) for c in ctxs]
    for B, S in ((1, 1), (64, 1), (256, 1), (256, 2), (256, 4), (512, 1), (512, 2), (512, 4), (1024, 2), (1024, 4)):
        prs = provers[:S]
        for k, pr in enumerate(prs):
            pr.upload([wits[i % 4] for i in range(B // S)])
        for pr in prs:
            pr.run()
        for pr in prs:
            pr.download_raw()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            for pr in prs:
                pr.run()
            ok = True
            for pr in prs:
                raw, st = pr.download_raw()
                ok = ok and not any(st)
        dt = (time.perf_counter() - t0) / reps
        print(json.dumps({"what": "prover", "B": B, "streams": S, "s_per_batch": dt, "proofs_per_s": B / dt,
                          "ms_per_proof": 1e3 * dt / B, "status_ok": ok}), flush=True)
if "proverc" in which:
    from bench import chain_program_lines, witness_for

    program = Program(chain_program_lines(2048), 2048)
    wits = [witness_for(program, i) for i in range(4)]
    ctxs = [ctx, Context(0)]
    provers = [BatchProver(setup, program, c) for c in ctxs]
    for c in (10, 11, 12):
        for G in (0, 1, 2):
            for B, S in ((512, 1), (512, 2)):
                prs = provers[:S]
                for pr in prs:
                    check(L.plonk_msm_configure(pr.ctx.handle, c, G))
                    pr.upload([wits[i % 4] for i in range(B // S)])
                for _ in range(2):
                    for pr in prs:
                        pr.run()
                    for pr in prs:
                        pr.download_raw()
                t0 = time.perf_counter()
                reps = 4
                for _ in range(reps):
                    for pr in prs:
                        pr.run()
                    for pr in prs:
                        pr.download_raw()
                dt = (time.perf_counter() - t0) / reps
                print(json.dumps({"what": "proverc", "c": c, "G": G, "B": B, "streams": S, "proofs_per_s": B / dt,
                                  "ms_per_step": 1e3 * dt}), flush=True)
    for pr in provers:
        check(L.plonk_msm_configure(pr.ctx.handle, 0, 0))
