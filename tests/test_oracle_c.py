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


def test_c_lincomb_matches_reference_vectors():
    setup = Setup.from_file(os.path.join(GOLDEN, "srs_2048.ptau"))
    P = setup.powers_of_x
    for case in load("lincomb_vectors.json")["cases"]:
        if "seed" in case:
            sc, idx = rand_vec(case["seed"], case["n"]), list(range(case["n"]))
        else:
            sc, idx = [int(s) % g1.R_MOD for s in case["scalars"]], case["points"]
        assert c_oracle.g1_lincomb([P[i] for i in idx], sc) == pt(case["result"]), case["name"]
