"""The C half of the oracle against the Python oracle (which is pinned to the reference's goldens)."""
import os
import time

from helpers import GOLDEN, check_summary, load, pt, rand_vec
from oracle import c_oracle, g1
from oracle.fr_poly import fft_ints
from oracle.srs import Setup


def test_c_ntt_matches_python_oracle():
    for log_n in (0, 1, 2, 5, 10, 13):
        v = rand_vec(log_n + 900, 1 << log_n)
        assert c_oracle.fr_ntt(v) == fft_ints(v)
        assert c_oracle.fr_ntt(v, True) == fft_ints(v, True)


def test_c_ntt_matches_reference_vector_2_16():
    case = [c for c in load("poly_vectors.json")["cases"] if c["log_n"] == 16][0]
    v = rand_vec(case["seed"], 1 << 16)
    check_summary(c_oracle.fr_ntt(v), case["fft"])
    check_summary(c_oracle.fr_ntt(v, True), case["ifft"])


def test_c_bls12_381_ntt_is_the_dft_over_that_field():
    """oracle_bls_fr_ntt has no reference counterpart (the reference is BN254 throughout): it is pinned by definition —
    X[k] = sum_j x[j] w^(jk) in Python integers with w = 7^((r-1)/N), where 7^((r-1)/2^32) is the `bls12_381` crate's
    published ROOT_OF_UNITY — and by inverse(forward(x)) == x."""
    import random

    r = c_oracle.BLS12_381_FR_MODULUS
    assert pow(7, (r - 1) >> 32, r) == 0x16A2A19EDFE81F20D09B681922C813B4B63683508C2280B93829971F439F0D2B
    assert (r - 1) % (1 << 32) == 0 and (r - 1) % (1 << 33) != 0 and pow(7, (r - 1) // 2, r) == r - 1
    rng = random.Random(12381)
    for log_n in (0, 1, 2, 4, 7):
        n = 1 << log_n
        xs = [rng.randrange(r) for _ in range(n)] if log_n != 4 else [r - 1] * n
        w = pow(7, (r - 1) // n, r)
        want = [sum(x * pow(w, j * k, r) for j, x in enumerate(xs)) % r for k in range(n)]
        raw = b"".join(x.to_bytes(32, "little") for x in xs)
        out = c_oracle.fr_ntt_bytes(raw, False, "bls12_381")
        assert [int.from_bytes(out[32 * i : 32 * i + 32], "little") for i in range(n)] == want, log_n
        assert c_oracle.fr_ntt_bytes(out, True, "bls12_381") == raw


def test_c_bls12_381_ntt_matches_the_committed_vectors():
    """tests/golden/bls12_381_ntt_vectors.json (tools/gen_bls_vectors.py: a recursive transform in Python integers, itself checked
    against the DFT sum) pins oracle_bls_fr_ntt at 2^8 .. 2^13 and 2^16, forward and inverse."""
    import random

    r = c_oracle.BLS12_381_FR_MODULUS
    un = lambda raw: [int.from_bytes(raw[32 * i:32 * i + 32], "little") for i in range(len(raw) // 32)]
    for case in load("bls12_381_ntt_vectors.json")["cases"]:
        rng = random.Random(case["seed"])
        raw = b"".join(rng.randrange(r).to_bytes(32, "little") for _ in range(1 << case["log_n"]))
        check_summary(un(c_oracle.fr_ntt_bytes(raw, False, "bls12_381")), case["fft"])
        check_summary(un(c_oracle.fr_ntt_bytes(raw, True, "bls12_381")), case["ifft"])


def test_c_lincomb_matches_reference_vectors():
    setup = Setup.from_file(os.path.join(GOLDEN, "srs_2048.ptau"))
    P = setup.powers_of_x
    for case in load("lincomb_vectors.json")["cases"]:
        if "seed" in case:
            sc, idx = rand_vec(case["seed"], case["n"]), list(range(case["n"]))
        else:
            sc, idx = [int(s) % g1.R_MOD for s in case["scalars"]], case["points"]
        assert c_oracle.g1_lincomb([P[i] for i in idx], sc) == pt(case["result"]), case["name"]
