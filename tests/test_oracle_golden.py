"""Pins the oracle (CPU restatement) against the reference's own known-answer data and against
vectors produced by the reference's own code (tools/gen_golden.py).  CPU-only."""
import hashlib
import json
import os

import pytest

from helpers import GOLDEN, check_summary, digest, load, pt, rand_vec
from oracle import field, g1
from oracle.circuit import Program
from oracle.fr_poly import Basis, Polynomial, fft_ints
from oracle.plonk_prover import Prover
from oracle.poseidon import poseidon_hash, poseidon_program_lines
from oracle.srs import Setup
from oracle.strobe_merlin import MerlinTranscript, Transcript, sha3_256_selfcheck

R = field.R_MOD


@pytest.fixture(scope="module")
def setup():
    return Setup.from_file(os.path.join(GOLDEN, "srs_2048.ptau"))


# ---------------------------------------------------------------- F1: field + roots of unity
def test_roots_of_unity_kats():
    sv = load("setup_vectors.json")
    assert field.root_of_unity(8) == int(sv["k2_omega8"])  # K2, test.py:30-33
    assert field.root_of_unity(8) == 19540430494807482326159819597004422086093766032135589407132600596362845576832
    for k, v in sv["roots"].items():
        assert field.root_of_unity(2 ** int(k)) == int(v)
    w = field.roots_of_unity(16)
    assert w[0] == 1 and len(w) == 16 and w[15] * w[1] % R == 1
    assert field.inv(0) == 0 and field.div(5, 0) == 0  # py_ecc: x / 0 == 0


# ---------------------------------------------------------------- transcript (merlin, un-vendored)
def test_keccak_against_hashlib():
    for msg in (b"", b"abc", bytes(range(200)), b"x" * 1000):
        assert sha3_256_selfcheck(msg) == hashlib.sha3_256(msg).digest()


def test_merlin_public_vector_and_reference_transcript():
    tv = load("transcript_vectors.json")
    t = MerlinTranscript(b"test protocol")
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == tv["merlin_simple_vector"]
    g = load("k6_proof.json")["proof"]
    t = Transcript(b"plonk")
    beta, gamma = t.round_1(pt(g["a_1"]), pt(g["b_1"]), pt(g["c_1"]))
    alpha, cof = t.round_2(pt(g["z_1"]))
    zeta = t.round_3(pt(g["t_lo_1"]), pt(g["t_mid_1"]), pt(g["t_hi_1"]))
    v = t.round_4(*[int(g[k]) for k in ("a_eval", "b_eval", "c_eval", "s1_eval", "s2_eval", "z_shifted_eval")])
    u = t.round_5(pt(g["W_z_1"]), pt(g["W_zw_1"]))
    got = dict(beta=beta, gamma=gamma, alpha=alpha, fft_cofactor=cof, zeta=zeta, v=v, u=u)
    assert {k: str(x) for k, x in got.items()} == tv["k6_challenges"]
    t2 = Transcript(b"plonk")
    t2.append_scalar(b"x", 12345)
    t2.append_message(b"raw", b"\x00\x01\x02")
    assert str(t2.get_and_append_challenge(b"ch")) == tv["misc_challenge"]


# ---------------------------------------------------------------- setup / K1
def test_setup_from_file_and_k1(setup):
    sv = load("setup_vectors.json")
    assert len(setup.powers_of_x) == sv["n_powers"] == 2048
    assert setup.powers_of_x[0] == (1, 2)
    assert setup.powers_of_x[1] == pt(sv["powers_of_x_1"])
    assert setup.powers_of_x[2047] == pt(sv["powers_of_x_2047"])
    assert digest([p[0] for p in setup.powers_of_x]) == sv["powers_x_digest"]
    assert digest([p[1] for p in setup.powers_of_x]) == sv["powers_y_digest"]
    assert [[str(c) for c in setup.X2[0]], [str(c) for c in setup.X2[1]]] == sv["X2"]
    assert all(g1.is_on_curve(p) for p in setup.powers_of_x[:32])
    c = setup.commit(Polynomial(list(range(1, 9)), Basis.LAGRANGE))
    assert c == pt(sv["k1_expected_test_py"]) == pt(sv["k1_commit_1to8"])  # test.py:18-28


# ---------------------------------------------------------------- poly.py
@pytest.mark.parametrize("case", load("poly_vectors.json")["cases"], ids=lambda c: "log%d_s%d" % (c["log_n"], c["seed"]))
def test_poly_against_reference(case):
    log_n, seed = case["log_n"], case["seed"]
    n = 1 << log_n
    if log_n > 13:
        pytest.skip("2^16 pure-Python case is exercised by the C oracle test")
    vals = rand_vec(seed, n)
    lag, mono = Polynomial(vals, Basis.LAGRANGE), Polynomial(vals, Basis.MONOMIAL)
    check_summary(mono.fft().values, case["fft"])
    check_summary(lag.ifft().values, case["ifft"])
    if "offset" in case:
        off = int(case["offset"])
        if "coset_extend" in case:
            check_summary(lag.to_coset_extended_lagrange(off).values, case["coset_extend"])
        check_summary(lag.coset_extended_lagrange_to_coeffs(off).values, case["coset_to_coeffs"])
    if "add" in case:
        other = rand_vec(seed + 500, n)
        if n >= 4:
            other[1] = 0
            other[3] = vals[3]
        olag = Polynomial(other, Basis.LAGRANGE)
        sc = int(case["scalar"])
        check_summary((lag + olag).values, case["add"])
        check_summary((lag - olag).values, case["sub"])
        check_summary((lag * olag).values, case["mul"])
        check_summary((lag / olag).values, case["div"])
        check_summary((lag + sc).values, case["add_scalar_lagrange"])
        check_summary((lag - sc).values, case["sub_scalar_lagrange"])
        check_summary((mono + sc).values, case["add_scalar_monomial"])
        check_summary((mono - sc).values, case["sub_scalar_monomial"])
        check_summary((lag * sc).values, case["mul_scalar"])
        check_summary((lag / sc).values, case["div_scalar"])
        if "shift" in case:
            check_summary(lag.shift(case["shift_k"]).values, case["shift"])
        assert lag.barycentric_eval(sc) == int(case["barycentric_at_scalar"])


def test_poly_asserts_match_reference_domain():
    a = Polynomial([1, 2, 3, 4], Basis.LAGRANGE)
    m = Polynomial([1, 2, 3, 4], Basis.MONOMIAL)
    with pytest.raises(AssertionError):
        a.fft()  # poly.py:141
    with pytest.raises(AssertionError):
        m.ifft()  # poly.py:132
    with pytest.raises(AssertionError):
        m * m  # poly.py:70
    with pytest.raises(AssertionError):
        a + m  # poly.py:26
    with pytest.raises(AssertionError):
        a.shift(4)  # poly.py:104
    assert fft_ints(fft_ints([5, 6, 7, 8]), True) == [5, 6, 7, 8]


# ---------------------------------------------------------------- curve.py
def test_lincomb_against_reference(setup):
    lv = load("lincomb_vectors.json")
    P = setup.powers_of_x
    for case in lv["cases"]:
        if "seed" in case:
            sc = rand_vec(case["seed"], case["n"])
            idx = list(range(case["n"]))
            if case["name"] != "n2048_seed101":
                continue  # one full-size reference-shaped MSM is enough for the CPU suite (~3 s)
        else:
            sc, idx = [int(s) for s in case["scalars"]], case["points"]
        pairs = [(P[i], s) for i, s in zip(idx, sc)]
        assert g1.ec_lincomb(pairs) == pt(case["result"]), case["name"]
        if len(pairs) <= 64:
            assert g1.ec_lincomb_naive(pairs) == pt(case["result"]), case["name"]
    k8 = lv["k8_int"]
    numbers, factors = [int(x) for x in k8["numbers"]], [int(x) for x in k8["factors"]]
    assert [str(x) for x in g1.multisubset(numbers, [set(s) for s in k8["subsets"]])] == k8["multisubset"]
    assert str(g1.lincomb(numbers, factors)) == k8["lincomb"]
    assert g1.lincomb(numbers, factors) == sum(n * f for n, f in zip(numbers, factors))  # curve.py:139


def test_g1_group_law_edges(setup):
    P = setup.powers_of_x
    assert g1.add(P[3], None) == P[3] and g1.add(None, P[3]) == P[3]
    assert g1.add(P[3], g1.neg(P[3])) is None
    assert g1.add(P[3], P[3]) == g1.double(P[3]) == g1.multiply(P[3], 2)
    assert g1.multiply(P[3], 0) is None
    assert g1.multiply(g1.G1, field.R_MOD) is None
    assert g1.is_on_curve(g1.multiply(P[5], 123456789))


# ---------------------------------------------------------------- compiler
def _program_for(case):
    if "constraints" in case:
        return case["constraints"]
    lines = poseidon_program_lines()
    assert hashlib.sha256("\n".join(lines).encode()).hexdigest() == case["constraints_sha256"]
    return lines


@pytest.mark.parametrize("case", load("compiler_vectors.json")["cases"], ids=lambda c: c["name"])
def test_compiler_against_reference(case):
    program = Program(_program_for(case), case["group_order"])
    pk = program.common_preprocessed_input()
    for key in ("QM", "QL", "QR", "QO", "QC", "S1", "S2", "S3"):
        check_summary(getattr(pk, key).values, case[key])
    assert program.get_public_assignments() == case["public"]
    assert hashlib.sha256(repr([list(w) for w in program.wires()]).encode()).hexdigest() == case["wires_sha256"]
    if "start" in case:
        filled = program.fill_variable_assignments(case["start"])
        keys = sorted(k for k in filled if k is not None)
        assert len(keys) == case["filled_nvars"]
        assert digest([filled[k] for k in keys]) == case["filled_digest"]


def test_k7_poseidon():
    cv = load("compiler_vectors.json")
    assert str(poseidon_hash(1, 2)) == cv["k7_poseidon_hash_1_2"]
    assert poseidon_hash(1, 2) == 2794293468621295827697063340852298151396911537992101357791526496888369165789
    filled = Program(poseidon_program_lines(), 1024).fill_variable_assignments({"L0": 1, "M0": 2})
    assert str(filled["M64"]) == cv["k7_witness_M64"] == cv["k7_poseidon_hash_1_2"]


def test_compiler_errors():
    with pytest.raises(Exception, match="Group order too small"):
        Program(["a <== b * c"] * 9, 8)
    with pytest.raises(Exception, match="Max 2 variables"):
        Program(["e <== a + b * c"], 8)
    with pytest.raises(Exception, match="Disallowed multiplication"):
        Program(["e <== a * a * a"], 8)
    with pytest.raises(Exception, match="Unsupported op"):
        Program(["a foo b"], 8)
    with pytest.raises(Exception, match="Public var declarations must be at the top"):
        Program(["c <== a * b", "c public"], 8).get_public_assignments()


# ---------------------------------------------------------------- K3-K5: verification keys
def _vkey_point(p):
    if p == ["0", "1", "0"]:
        return None  # utils.py:13-14
    assert p[2] == "1"
    return (int(p[0]), int(p[1]))


@pytest.mark.parametrize(
    "fname,lines",
    [
        ("main.plonk.vkey.json", ["c <== a * b"]),  # test.py:43-54
        ("main.plonk.vkey-58.json", ["ab === a - c", "-ab === a * b"]),  # test.py:70-81
        ("main.plonk.vkey-59.json", ["c public", "c === a * b"]),  # test.py:88-99
    ],
)
def test_verification_key_goldens(setup, fname, lines):
    theirs = load(fname)
    pk = Program(lines, 8).common_preprocessed_input()
    for key, poly in (("Qm", pk.QM), ("Ql", pk.QL), ("Qr", pk.QR), ("Qo", pk.QO), ("Qc", pk.QC),
                      ("S1", pk.S1), ("S2", pk.S2), ("S3", pk.S3)):
        assert setup.commit(poly) == _vkey_point(theirs[key]), key
    x2 = theirs["X_2"]
    assert ((int(x2[0][0]), int(x2[0][1])), (int(x2[1][0]), int(x2[1][1]))) == setup.X2
    assert field.root_of_unity(8) == int(theirs["w"])


# ---------------------------------------------------------------- K6: the golden proof
def test_k6_golden_proof(setup):
    k6 = load("k6_proof.json")
    prover = Prover(setup, Program(k6["program"], k6["group_order"]))
    proof = prover.prove(k6["witness"]).flatten()
    for k, v in k6["proof"].items():
        want = pt(v) if isinstance(v, list) else int(v)
        assert proof[k] == want, k
    tv = load("transcript_vectors.json")["k6_challenges"]
    for k, v in prover.challenges.items():
        assert str(v) == tv[k]


# ---------------------------------------------------------------- N4: pairing + verifier (oracle only)
def test_pairing_bilinearity():
    from oracle import pairing as pr

    assert pr.is_on_curve(pr.G2, pr.B2)
    assert pr.multiply(pr.G2, R) is None
    e1 = pr.pairing(pr.G2, g1.G1)
    assert e1 != pr.FQ12.one() and e1 ** R == pr.FQ12.one()
    assert pr.pairing(pr.multiply(pr.G2, 7), g1.multiply(g1.G1, 5)) == e1 ** 35
    x = pr.FQ12(list(range(1, 13)))
    assert x * x.inv() == pr.FQ12.one()


def test_golden_proof_verifies(setup):
    """test.py:272-275 — the reference's golden proof must pass the (restated) complete verifier;
    tampered proofs and wrong public inputs must not."""
    from oracle.verifier import VerificationKey

    k6 = load("k6_proof.json")
    pk = Program(k6["program"], 8).common_preprocessed_input()
    vk = VerificationKey.from_setup(setup, pk)
    proof = {k: (pt(v) if isinstance(v, list) else int(v)) for k, v in k6["proof"].items()}
    assert vk.verify_proof(8, proof, [60])
    assert not vk.verify_proof(8, proof, [61])
    for key in ("a_eval", "z_shifted_eval"):
        bad = dict(proof)
        bad[key] = (bad[key] + 1) % R
        assert not vk.verify_proof(8, bad, [60])
    bad = dict(proof)
    bad["W_z_1"] = g1.double(bad["W_z_1"])
    assert not vk.verify_proof(8, bad, [60])


def test_compressed_proof_bytes_of_the_golden_proof():
    """The oracle's codec on the reference's golden proof (test/proof.pickle) against tests/golden/k6_proof_bytes.json
    (tools/gen_proof_bytes.py writes the encoding's definition out independently), and back."""
    from oracle import g1 as og1

    g = load("k6_proof.json")["proof"]
    flat = {k: (pt(v) if isinstance(v, list) or v is None else int(v)) for k, v in g.items()}
    want = bytes.fromhex(load("k6_proof_bytes.json")["hex"])
    assert og1.proof_to_bytes(flat) == want
    for i, k in enumerate(("a_1", "b_1", "c_1", "z_1", "t_lo_1", "t_mid_1", "t_hi_1", "W_z_1", "W_zw_1")):
        assert og1.decompress(want[32 * i : 32 * i + 32]) == flat[k]
    assert og1.compress(None) == bytes([0x40]) + bytes(31) and og1.decompress(og1.compress(None)) is None
    assert og1.decompress(og1.compress(og1.neg(og1.G1))) == og1.neg(og1.G1)
