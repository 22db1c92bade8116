#!/usr/bin/env python3
"""Golden-vector generator.  BUILD-CONTAINER ONLY (needs /root/reference; never runs on the GPU box).

Imports the reference's OWN files (poly.py, curve.py, setup.py, transcript.py, compiler/*,
test/mini_poseidon.py) unchanged from /root/reference, with tools/refshim/ standing in for the two
un-installed third-party packages, runs them on seeded inputs and writes inputs-by-seed +
expected outputs as small JSON fixtures under tests/golden/.  Also converts the data files the
reference's own tests hold (proof.pickle, vkey JSONs, poseidon_rc.json, the first 2^11 G1 powers +
[1]_2,[x]_2 of the .ptau) into fixtures.  Only data is written — no reference source text.

    python tools/gen_golden.py            # regenerate everything (~1-2 min)
"""
import hashlib
import json
import os
import pickle
import random
import shutil
import sys

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")

sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REPO, "tools", "refshim"))
os.chdir(REF)  # the reference opens fixtures by relative path (test.py:17, mini_poseidon.py:21)

import io, contextlib  # noqa: E402

with contextlib.redirect_stdout(io.StringIO()):
    from curve import Scalar, ec_lincomb, lincomb, multisubset  # noqa: E402
    from poly import Basis, Polynomial  # noqa: E402
    from setup import Setup  # noqa: E402
    from compiler.program import Program  # noqa: E402
    from transcript import Transcript, Message1, Message2, Message3, Message4, Message5  # noqa: E402
    import py_ecc.bn128 as b  # noqa: E402
    from test.mini_poseidon import rc, mds, poseidon_hash  # noqa: E402

R = Scalar.field_modulus


def digest(ints):
    h = hashlib.sha256()
    for v in ints:
        h.update(int(v).to_bytes(32, "big"))
    return h.hexdigest()


def summarise(ints, full_below=65):
    ints = [int(v) for v in ints]
    d = {"n": len(ints), "sha256_be32": digest(ints)}
    if len(ints) < full_below:
        d["values"] = [str(v) for v in ints]
    else:
        d["head"] = [str(v) for v in ints[:4]]
        d["tail"] = [str(v) for v in ints[-4:]]
    return d


def rand_vec(seed, n):
    rng = random.Random(seed)
    return [rng.randrange(R) for _ in range(n)]


def pt(p):
    return None if p is None else [str(p[0].n), str(p[1].n)]


def write(name, obj):
    with open(os.path.join(OUT, name), "w") as f:
        json.dump(obj, f, indent=1, sort_keys=True)
    print("wrote", name)


# --------------------------------------------------------------------------- data-file fixtures
def data_fixtures():
    # K6: the golden proof, unpickled without py_ecc via a find_class stub
    class Stub:
        def __setstate__(self, s):
            self.__dict__.update(s)

    class U(pickle.Unpickler):
        def find_class(self, module, name):
            return type(name, (Stub,), {})

    p = U(open("test/proof.pickle", "rb")).load()
    gold = {}
    for m in (p.msg_1, p.msg_2, p.msg_3, p.msg_4, p.msg_5):
        for k, v in m.__dict__.items():
            gold[k] = [str(v[0].n), str(v[1].n)] if isinstance(v, tuple) else str(v.n)
    write(
        "k6_proof.json",
        {
            "source": "test/proof.pickle (test.py:272-273)",
            "program": ["e public", "c <== a * b", "e <== c * d"],
            "group_order": 8,
            "witness": {"a": 3, "b": 4, "c": 12, "d": 5, "e": 60},
            "proof": gold,
        },
    )
    for name in ("main.plonk.vkey.json", "main.plonk.vkey-58.json", "main.plonk.vkey-59.json", "poseidon_rc.json"):
        shutil.copyfile(os.path.join("test", name), os.path.join(OUT, name))
        print("copied", name)
    # SRS slice in .ptau layout: header (80 B) + first 2^11 G1 points + [1]_2, [x]_2.  Parses
    # identically under setup.py:23-63 (byte 60 = power, G1 @80, G2 generator found by scan).
    raw = open("test/powersOfTau28_hez_final_11.ptau", "rb").read()
    n_g1 = 2 ** raw[60]
    g1_end = 80 + 64 * n_g1
    target = (b.FQ(int.from_bytes(raw[80:112], "little")) / b.G1[0] * b.G2[0].coeffs[0]).n.to_bytes(32, "little")
    g2 = raw.find(target, g1_end)
    mini = raw[:g1_end] + raw[g2 : g2 + 256]
    with open(os.path.join(OUT, "srs_2048.ptau"), "wb") as f:
        f.write(mini)
    print("wrote srs_2048.ptau", len(mini), "bytes; G1 slice sha256", hashlib.sha256(raw[80:g1_end]).hexdigest())


# --------------------------------------------------------------------------- setup / K1 / K2
def setup_vectors(setup):
    dummy = Polynomial(list(map(Scalar, range(1, 9))), Basis.LAGRANGE)
    coeffs = dummy.ifft().values
    k1 = ec_lincomb([(setup.powers_of_x[i], c) for i, c in enumerate(coeffs)])
    write(
        "setup_vectors.json",
        {
            "source": "reference setup.py:23-63 run on test/powersOfTau28_hez_final_11.ptau",
            "n_powers": len(setup.powers_of_x),
            "powers_of_x_1": pt(setup.powers_of_x[1]),
            "powers_of_x_2047": pt(setup.powers_of_x[2047]),
            "powers_x_digest": digest([p[0].n for p in setup.powers_of_x]),
            "powers_y_digest": digest([p[1].n for p in setup.powers_of_x]),
            "X2": [[str(c.n) for c in setup.X2[0].coeffs], [str(c.n) for c in setup.X2[1].coeffs]],
            "k1_commit_1to8": pt(k1),
            "k1_expected_test_py": [
                "16120260411117808045030798560855586501988622612038310041007562782458075125622",
                "3125847109934958347271782137825877642397632921923926105820408033549219695465",
            ],
            "k2_omega8": str(Scalar.root_of_unity(8).n),
            "roots": {str(k): str(Scalar.root_of_unity(2**k).n) for k in (1, 3, 4, 10, 11, 13, 16, 20, 24, 28)},
        },
    )


# --------------------------------------------------------------------------- poly.py vectors
def poly_vectors():
    out = {"source": "reference poly.py run on random.Random(seed).randrange(r) vectors", "cases": []}
    for log_n in (0, 1, 3, 6, 11, 13, 16):
        n = 1 << log_n
        for seed in (1, 2, 3) if log_n <= 11 else (1,):
            vals = rand_vec(seed * 1000 + log_n, n)
            lag = Polynomial([Scalar(v) for v in vals], Basis.LAGRANGE)
            mono = Polynomial([Scalar(v) for v in vals], Basis.MONOMIAL)
            case = {"log_n": log_n, "seed": seed * 1000 + log_n}
            case["fft"] = summarise([x.n for x in mono.fft().values])
            case["ifft"] = summarise([x.n for x in lag.ifft().values])
            if log_n <= 13:
                offset = Scalar(rand_vec(seed + 77, 1)[0])
                case["offset"] = str(offset.n)
                if log_n >= 1 and log_n <= 11:
                    case["coset_extend"] = summarise([x.n for x in lag.to_coset_extended_lagrange(offset).values])
                case["coset_to_coeffs"] = summarise([x.n for x in lag.coset_extended_lagrange_to_coeffs(offset).values])
            if log_n <= 11:
                other = rand_vec(seed * 1000 + log_n + 500, n)
                if n >= 4:
                    other[1] = 0  # exercises x / 0 == 0
                    other[3] = vals[3]
                olag = Polynomial([Scalar(v) for v in other], Basis.LAGRANGE)
                sc = Scalar(rand_vec(seed + 99, 1)[0])
                case["scalar"] = str(sc.n)
                case["add"] = summarise([x.n for x in (lag + olag).values])
                case["sub"] = summarise([x.n for x in (lag - olag).values])
                case["mul"] = summarise([x.n for x in (lag * olag).values])
                case["div"] = summarise([x.n for x in (lag / olag).values])
                case["add_scalar_lagrange"] = summarise([x.n for x in (lag + sc).values])
                case["sub_scalar_lagrange"] = summarise([x.n for x in (lag - sc).values])
                case["add_scalar_monomial"] = summarise([x.n for x in (mono + sc).values])
                case["sub_scalar_monomial"] = summarise([x.n for x in (mono - sc).values])
                case["mul_scalar"] = summarise([x.n for x in (lag * sc).values])
                case["div_scalar"] = summarise([x.n for x in (lag / sc).values])
                if n > 1:
                    k = 1 if n < 8 else 4
                    case["shift_k"] = k
                    case["shift"] = summarise([x.n for x in lag.shift(k).values])
                case["barycentric_at_scalar"] = str(lag.barycentric_eval(sc).n)
            out["cases"].append(case)
            print("  poly log_n", log_n, "seed", seed)
    write("poly_vectors.json", out)


# --------------------------------------------------------------------------- curve.py vectors
def lincomb_vectors(setup):
    out = {"source": "reference curve.py:38-111 ec_lincomb over the .ptau G1 powers", "cases": []}
    P = setup.powers_of_x

    def run(name, points_idx, scalars):
        res = ec_lincomb([(P[i], s) for i, s in zip(points_idx, scalars)])
        out["cases"].append({"name": name, "points": points_idx, "scalars": [str(s) for s in scalars], "result": pt(res)})

    run("n8_seed1", list(range(8)), rand_vec(1, 8))
    run("n8_edge", list(range(8)), [0, 1, R - 1, 2, R - 2, 0, 1, 5])
    run("all_zero", list(range(4)), [0, 0, 0, 0])
    run("cancel_to_identity", [3, 3], [7, R - 7])
    run("duplicates", [1, 1, 2, 2, 1], [3, 4, 5, R - 5, 9])
    run("single", [5], [rand_vec(9, 1)[0]])
    run("scalar_ge_r", [0, 1], [R + 5, 2 * R + 1])  # curve.py:41 reduces mod r
    run("n64_seed2", list(range(64)), rand_vec(2, 64))
    run("n64_small_scalars", list(range(64)), [random.Random(5).randrange(16) for _ in range(64)])
    for seed in (1, 2):
        sc = rand_vec(100 + seed, 2048)
        res = ec_lincomb([(P[i], s) for i, s in enumerate(sc)])
        out["cases"].append({"name": "n2048_seed%d" % (100 + seed), "seed": 100 + seed, "n": 2048, "result": pt(res)})
        print("  lincomb 2048 seed", seed)
    # K8: integer-adder self tests of multisubset / lincomb (curve.py:126-149)
    rng = random.Random(8)
    numbers = [rng.randrange(10**20) for _ in range(40)]
    factors = [rng.randrange(2**256) for _ in range(40)]
    subsets = [sorted(i for i in range(40) if rng.randrange(2)) for _ in range(12)]
    out["k8_int"] = {
        "numbers": [str(x) for x in numbers],
        "factors": [str(x) for x in factors],
        "subsets": subsets,
        "multisubset": [str(x) for x in multisubset(numbers, [set(s) for s in subsets])],
        "lincomb": str(lincomb(numbers, factors)),
    }
    write("lincomb_vectors.json", out)


# --------------------------------------------------------------------------- compiler vectors
def poseidon_program_lines():
    """The generator of test.py:216-239, restated (data for the fixture, not shipped code)."""
    o = ["L0 public", "M0 public", "M64 public", "R0 <== 0"]
    for i in range(64):
        for j, pos in enumerate(("L", "M", "R")):
            f = {"x": i, "r": rc[i][j], "p": pos}
            if i < 4 or i >= 60 or pos == "L":
                o.append("{p}adj{x} <== {p}{x} + {r}".format(**f))
                o.append("{p}sq{x} <== {p}adj{x} * {p}adj{x}".format(**f))
                o.append("{p}qd{x} <== {p}sq{x} * {p}sq{x}".format(**f))
                o.append("{p}qn{x} <== {p}qd{x} * {p}adj{x}".format(**f))
            else:
                o.append("{p}qn{x} <== {p}{x} + {r}".format(**f))
        for j, pos in enumerate(("L", "M", "R")):
            o.append("{p}suma{x} <== Lqn{x} * {m}".format(x=i, p=pos, m=mds[j]))
            o.append("{p}sumb{x} <== {p}suma{x} + Mqn{x} * {m}".format(x=i, p=pos, m=mds[j + 1]))
            o.append("{p}{xp1} <== {p}sumb{x} + Rqn{x} * {m}".format(x=i, xp1=i + 1, p=pos, m=mds[j + 2]))
    return o


FACTORIZATION = """n public
pb0 === pb0 * pb0
pb1 === pb1 * pb1
pb2 === pb2 * pb2
pb3 === pb3 * pb3
qb0 === qb0 * qb0
qb1 === qb1 * qb1
qb2 === qb2 * qb2
qb3 === qb3 * qb3
pb01 <== pb0 + 2 * pb1
pb012 <== pb01 + 4 * pb2
p <== pb012 + 8 * pb3
qb01 <== qb0 + 2 * qb1
qb012 <== qb01 + 4 * qb2
q <== qb012 + 8 * qb3
n <== p * q""".split("\n")


def compiler_vectors(setup):
    out = {"source": "reference compiler/program.py run on the circuits of test.py", "cases": []}
    progs = [
        ("k3_c_eq_ab", ["c <== a * b"], 8, None),
        ("k4_ab_plus_a", ["ab === a - c", "-ab === a * b"], 8, None),
        ("k5_one_public", ["c public", "c === a * b"], 8, None),
        ("k6_three_line", ["e public", "c <== a * b", "e <== c * d"], 8, None),
        ("factorization", FACTORIZATION, 16,
         {"pb3": 1, "pb2": 1, "pb1": 0, "pb0": 1, "qb3": 0, "qb2": 1, "qb1": 1, "qb0": 1}),
        ("misc_grammar", ["x public", "y <== x * x - 45 * x + 987", "z === 9", "-w <== y * z", "u <== 3 - y"], 8,
         {"x": 7, "z": 9}),
        ("poseidon_1024", poseidon_program_lines(), 1024, {"L0": 1, "M0": 2}),
    ]
    for name, lines, n, start in progs:
        program = Program(lines, n)
        pk = program.common_preprocessed_input()
        case = {"name": name, "group_order": n, "n_constraints": len(lines)}
        if len(lines) <= 16:
            case["constraints"] = lines
        else:
            case["constraints_sha256"] = hashlib.sha256("\n".join(lines).encode()).hexdigest()
        for key in ("QM", "QL", "QR", "QO", "QC", "S1", "S2", "S3"):
            case[key] = summarise([x.n for x in getattr(pk, key).values], full_below=17)
        case["public"] = program.get_public_assignments()
        case["wires_sha256"] = hashlib.sha256(repr([w.as_list() for w in program.wires()]).encode()).hexdigest()
        if start is not None:
            filled = program.fill_variable_assignments(start)
            keys = sorted(k for k in filled if k is not None)
            case["start"] = start
            case["filled_digest"] = digest([filled[k] for k in keys])
            case["filled_nvars"] = len(keys)
            if len(keys) <= 32:
                case["filled"] = {k: str(filled[k]) for k in keys}
        out["cases"].append(case)
        print("  compiler", name)
    out["k7_poseidon_hash_1_2"] = str(poseidon_hash(1, 2).n)
    filled = Program(poseidon_program_lines(), 1024).fill_variable_assignments({"L0": 1, "M0": 2})
    out["k7_witness_M64"] = str(filled["M64"])
    out["mds"] = [str(m.n) for m in mds]
    write("compiler_vectors.json", out)


# --------------------------------------------------------------------------- transcript vectors
def transcript_vectors():
    g = json.load(open(os.path.join(OUT, "k6_proof.json")))["proof"]

    def P(k):
        return (b.FQ(int(g[k][0])), b.FQ(int(g[k][1])))

    def S(k):
        return Scalar(int(g[k]))

    t = Transcript(b"plonk")
    beta, gamma = t.round_1(Message1(P("a_1"), P("b_1"), P("c_1")))
    alpha, cof = t.round_2(Message2(P("z_1")))
    zeta = t.round_3(Message3(P("t_lo_1"), P("t_mid_1"), P("t_hi_1")))
    v = t.round_4(Message4(S("a_eval"), S("b_eval"), S("c_eval"), S("s1_eval"), S("s2_eval"), S("z_shifted_eval")))
    u = t.round_5(Message5(P("W_z_1"), P("W_zw_1")))
    t2 = Transcript(b"plonk")
    t2.append_scalar(b"x", Scalar(12345))
    t2.append(b"raw", b"\x00\x01\x02")
    c2 = t2.get_and_append_challenge(b"ch")
    write(
        "transcript_vectors.json",
        {
            "source": "reference transcript.py:58-123 driven with the K6 proof messages",
            "k6_challenges": {k: str(x.n) for k, x in
                              dict(beta=beta, gamma=gamma, alpha=alpha, fft_cofactor=cof, zeta=zeta, v=v, u=u).items()},
            "misc_challenge": str(c2.n),
            "merlin_simple_vector": "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615",
        },
    )


def main():
    os.makedirs(OUT, exist_ok=True)
    data_fixtures()
    with contextlib.redirect_stdout(io.StringIO()):
        setup = Setup.from_file("test/powersOfTau28_hez_final_11.ptau")
    setup_vectors(setup)
    transcript_vectors()
    poly_vectors()
    compiler_vectors(setup)
    lincomb_vectors(setup)


if __name__ == "__main__":
    main()
