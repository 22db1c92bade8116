"""Shared helpers for the test-suite (fixture loading, digests, seeded inputs)."""
import hashlib
import json
import os
import random

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def digest(ints):
    h = hashlib.sha256()
    for v in ints:
        h.update(int(v).to_bytes(32, "big"))
    return h.hexdigest()


def check_summary(ints, summary):
    ints = [int(v) for v in ints]
    assert len(ints) == summary["n"]
    if "values" in summary:
        assert ints == [int(v) for v in summary["values"]]
    else:
        assert ints[:4] == [int(v) for v in summary["head"]]
        assert ints[-4:] == [int(v) for v in summary["tail"]]
    assert digest(ints) == summary["sha256_be32"]


def rand_vec(seed, n):
    """Same generator as tools/gen_golden.py: random.Random(seed).randrange(r)."""
    rng = random.Random(seed)
    return [rng.randrange(R_MOD) for _ in range(n)]


def pt(v):
    return None if v is None else (int(v[0]), int(v[1]))
