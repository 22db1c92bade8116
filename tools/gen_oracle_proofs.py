#!/usr/bin/env python3
"""Generates tests/golden/oracle_proofs.json: full proofs (9 G1 + 6 Fr + 6 challenges) produced by the
ORACLE prover (oracle/plonk_prover.py, pinned to the reference by the K6 golden proof) for circuits whose
reference-shaped CPU proof takes too long to recompute inside the GPU test-suite (group_order 2^10, 2^11).
These are regression vectors of the restatement, not reference outputs (the reference ships no
prover); small circuits are compared against the live oracle instead.     ~2 min on one core."""
import json
import os
import sys
import time

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)
from oracle.circuit import Program  # noqa: E402
from oracle.plonk_prover import Prover  # noqa: E402
from oracle.poseidon import poseidon_hash, poseidon_program_lines  # noqa: E402
from oracle.srs import Setup  # noqa: E402


def chain_lines(n):
    return ["x0 public"] + ["x%d <== x%d * x%d" % (i + 1, i, i) for i in range(n - 1)]


def main():
    setup = Setup.from_file(os.path.join(REPO, "tests", "golden", "srs_2048.ptau"))
    cases = [
        ("chain_2048_x0_3", chain_lines(2048), 2048, {"x0": 3}),
        ("chain_2048_x0_4", chain_lines(2048), 2048, {"x0": 4}),
        ("poseidon_1024", poseidon_program_lines(), 1024, {"L0": 1, "M0": 2}),  # test.py:242-259
        ("poseidon_2048", poseidon_program_lines(), 2048, {"L0": 1, "M0": 2}),  # BASELINE configs[2]
    ]
    out = {"source": "oracle/plonk_prover.py (CPU restatement, pinned by K6)", "cases": []}
    for name, lines, n, start in cases:
        prog = Program(lines, n)
        wit = prog.fill_variable_assignments(start)
        t0 = time.time()
        prover = Prover(setup, prog)
        proof = prover.prove(dict(wit)).flatten()
        dt = time.time() - t0
        enc = {k: ([str(v[0]), str(v[1])] if isinstance(v, tuple) else str(v)) for k, v in proof.items()}
        case = {"name": name, "group_order": n, "start": start, "proof": enc,
                "challenges": {k: str(v) for k, v in prover.challenges.items()}, "oracle_seconds": round(dt, 1)}
        if name.startswith("chain"):
            case["program"] = "chain"
        else:
            case["program"] = "poseidon"
            assert wit["M64"] == poseidon_hash(1, 2)
        out["cases"].append(case)
        print(name, "%.1fs" % dt, flush=True)
    with open(os.path.join(REPO, "tests", "golden", "oracle_proofs.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
