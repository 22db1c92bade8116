#!/usr/bin/env python3
"""Single-proof latency at group_order 2^k (BASELINE configs[1] circuit): the reference-shaped `Prover.prove` (with and
without its sanity asserts) and the lock-step `BatchProver.prove` with a batch of one.  One JSON line."""
import json
import os
import sys
import time

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)
import bench  # noqa: E402  (chain circuit + seeded witnesses)
from plonkathon_amd import BatchProver, Program, Prover, Setup, get_context  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 11
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = 1 << log_n
setup = Setup.from_file(os.path.join(REPO, "tests", "golden", "srs_2048.ptau"))
program = Program(bench.chain_program_lines(n), n)
ctx = get_context()
out = {"what": "latency", "group_order": n, "reps": reps}


def timed(fn):
    fn()  # warm: tables, Lagrange SRS, kernel code
    ts = []
    for i in range(reps):
        t0 = time.perf_counter()
        fn()
        ctx.sync()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return {"best_ms": round(1e3 * ts[0], 3), "median_ms": round(1e3 * ts[len(ts) // 2], 3)}


assert n == bench.GROUP_ORDER
wits = [bench.witness_for(i) for i in range(3)]
if os.environ.get("LATENCY_ONLY") == "b1":  # for a kernel trace: nothing but batches of one through the lock-step prover
    bp = BatchProver(setup, program)
    print(json.dumps(dict(out, batch_prover_b1=timed(lambda: bp.prove(dict(wits[0]))))))
    sys.exit(0)
api = Prover(setup, program)
k = [0]


def run_api():
    k[0] += 1
    return api.prove(dict(wits[k[0] % 3]))


out["api_prover_with_asserts"] = timed(run_api)
api.check = False
out["api_prover"] = timed(run_api)
bp = BatchProver(setup, program)
out["batch_prover_b1"] = timed(lambda: bp.prove(dict(wits[0])))
a, b = api.prove(dict(wits[0])).flatten(), bp.prove(dict(wits[0])).flatten()
out["api_equals_batch"] = all((a[key] == b[key]) for key in a)
print(json.dumps(out))
if os.environ.get("LATENCY_PROFILE"):
    import cProfile, pstats
    api.check = False
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        run_api()
    ctx.sync()
    pr.disable()
    pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(35)
