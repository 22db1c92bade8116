"""Parity checks of the plonkathon_amd API against the oracle and the committed golden vectors.

The same bodies run twice: under `-m gpu` against the real libplonk_hip.so on an MI355X (the parity
tests proper), and in the CPU suite against the emulated build of the same kernel sources (logic
check only).  They read like the reference's own test.py: build a Setup / Program / Prover, compare
with known answers.
"""
import hashlib
import os

from helpers import GOLDEN, check_summary, digest, load, pt, rand_vec, R_MOD

import plonkathon_amd as pa
from plonkathon_amd.field import Fq, Q_MOD
from plonkathon_amd import Basis, Polynomial, Program, Prover, Scalar, Setup, Transcript
from oracle import field as ofield, g1 as og1
from oracle.circuit import Program as OProgram
from oracle.fr_poly import Basis as OBasis, Polynomial as OPoly, fft_ints
from oracle.plonk_prover import Prover as OProver
from oracle.poseidon import poseidon_hash, poseidon_program_lines
from oracle.srs import Setup as OSetup

PTAU = os.path.join(GOLDEN, "srs_2048.ptau")


def P(ints, basis=Basis.LAGRANGE):
    return Polynomial.from_ints(ints, basis)


def ints(poly):
    return [x.n for x in poly.values]


def affine(p):
    return None if p is None else (p[0].n, p[1].n)


# ------------------------------------------------------------------------------------------ NTT
def ntt_vs_oracle(log_ns, seed0=0):
    for log_n in log_ns:
        n = 1 << log_n
        v = rand_vec(seed0 + log_n, n)
        assert ints(P(v, Basis.MONOMIAL).fft()) == fft_ints(v), ("fft", log_n)
        assert ints(P(v, Basis.LAGRANGE).ifft()) == fft_ints(v, True), ("ifft", log_n)


def ntt_quad_sizes(seed0=60):
    """2^14 = 2^7 x 2^7 and 2^15 = 2^7 x 2^8 on the wave kernels (the sizes no pair of the 4- / 8-element kernels reaches; until
    round 4 four-point column transforms + one pass): plain transforms both ways, in place through the C-ABI, a batch, and the
    fused forms the prover's sizes would use — zero padding + coset factors on the way in (to_coset_extended_lagrange of 2^12
    values -> 2^14) and coset^-1 / N on the way out."""
    from plonkathon_amd import get_context

    _ntt_quad_sizes_body(get_context(), seed0)  # (kernel kinds 5 .. 7 pick the same two kernels for these sizes)


def _ntt_quad_sizes_body(ctx, seed0):
    from plonkathon_amd._lib import check

    ntt_vs_oracle((14, 15), seed0=seed0)
    for log_n in (14, 15):
        n = 1 << log_n
        vs = [rand_vec(seed0 + 7 * log_n + b, n) for b in range(3)]
        buf = ctx.upload_ints(vs[0] + vs[1] + vs[2])
        check(ctx.L.plonk_fr_ntt(ctx.handle, buf.ptr, buf.ptr, log_n, 1, 3))  # inverse, in place, batch of three
        got = ctx.download_ints(buf)
        for b in range(3):
            assert got[b * n:(b + 1) * n] == fft_ints(vs[b], True), ("batch", log_n, b)
    v = rand_vec(seed0 + 99, 1 << 12)
    off = rand_vec(seed0 + 98, 1)[0]
    big = P(v).to_coset_extended_lagrange(Scalar(off))
    want = OPoly(v, OBasis.LAGRANGE).to_coset_extended_lagrange(off)
    assert ints(big) == want.values
    assert ints(big.coset_extended_lagrange_to_coeffs(Scalar(off))) == want.coset_extended_lagrange_to_coeffs(off).values


class ntt_kind:
    """with ntt_kind(k): the context's transforms run on kernel family k (plonk_ntt_select_kernel), back to automatic afterwards."""

    def __init__(self, kind):
        self.kind = kind

    def __enter__(self):
        from plonkathon_amd import get_context
        from plonkathon_amd._lib import check

        ctx = get_context()
        check(ctx.L.plonk_ntt_select_kernel(ctx.handle, self.kind))

    def __exit__(self, *exc):
        from plonkathon_amd import get_context
        from plonkathon_amd._lib import check

        ctx = get_context()
        check(ctx.L.plonk_ntt_select_kernel(ctx.handle, 0))


def ntt_two_pass_exact(log_ns, seed0=4000, batch=1):
    """Two-pass sizes exact against the C half of the oracle, forward out of place and inverse in place, `batch` transforms per call."""
    from oracle import c_oracle
    from plonkathon_amd import get_context
    from plonkathon_amd._lib import check

    ctx = get_context()
    for log_n in log_ns:
        n = 1 << log_n
        vs = [rand_vec(seed0 + log_n + 31 * b, n) for b in range(batch)]
        buf = ctx.upload_ints([x for v in vs for x in v])
        out = ctx.alloc(batch * n)
        check(ctx.L.plonk_fr_ntt(ctx.handle, buf.ptr, out.ptr, log_n, 0, batch))
        got = ctx.download_ints(out)
        for b in range(batch):
            assert got[b * n:(b + 1) * n] == c_oracle.fr_ntt(vs[b]), ("fwd", log_n, b)
        check(ctx.L.plonk_fr_ntt(ctx.handle, buf.ptr, buf.ptr, log_n, 1, batch))
        got = ctx.download_ints(buf)
        for b in range(batch):
            assert got[b * n:(b + 1) * n] == c_oracle.fr_ntt(vs[b], True), ("inv", log_n, b)


def ntt_latency_forms(two_pass=(14, 15, 16, 17, 18), batched=(18,), forced=((16, 9),)):
    """Round 4's two-element-per-thread kernels (2^7, and 2^9 in its latency form), forced with kernel kind 7: alone with random
    and range-driving inputs, and as the passes of 2^14 = 2^7 x 2^7, 2^15 = 2^7 x 2^8, 2^16 = 2^7 x 2^9, 2^17 = 2^8 x 2^9,
    2^18 = 2^9 x 2^9 (column pass on the full inter-pass table up to 2^16 and for the batched call, on the two small tables
    otherwise).  Kind 6 keeps a call off them: the same sizes then run on the four- and eight-element kernels."""
    with ntt_kind(7):
        ntt_vs_oracle((7, 9), seed0=770)
        ntt_extreme_inputs((7, 9))
        ntt_extreme_limbs((7, 9), slots=2)
        ntt_two_pass_exact(two_pass, seed0=4700)
        ntt_two_pass_exact(batched, seed0=4800, batch=2)
        from plonkathon_amd import get_context
        from plonkathon_amd._lib import check

        ctx = get_context()
        for log_n, r1 in forced:  # e.g. 2^16 = 2^9 x 2^7: the 256-thread two-element kernel as a column pass on the one-table twiddles
            try:
                check(ctx.L.plonk_ntt_set_split(ctx.handle, log_n, r1))
                ntt_two_pass_exact((log_n,), seed0=4900 + r1)
            finally:
                check(ctx.L.plonk_ntt_set_split(ctx.handle, log_n, 0))


def ntt_extreme_inputs(log_ns):
    """Inputs that drive the limb-form kernel's range bounds: every element at r - 1 (all partial sums at their maximum),
    alternating 0 / r - 1 (differences at their extremes), a lone r - 1, and vectors whose transform is constant."""
    m = R_MOD
    for log_n in log_ns:
        n = 1 << log_n
        for name, v in (("max", [m - 1] * n), ("alt", [(m - 1) * (i & 1) for i in range(n)]),
                        ("alt8", [(m - 1) * ((i >> 3) & 1) for i in range(n)]), ("one", [m - 1] + [0] * (n - 1)),
                        ("half", [(m - 1) // 2 + (i % 3) for i in range(n)])):
            assert ints(P(v, Basis.MONOMIAL).fft()) == fft_ints(v), ("fft", name, log_n)
            assert ints(P(v, Basis.LAGRANGE).ifft()) == fft_ints(v, True), ("ifft", name, log_n)


def ntt_extreme_limbs(log_ns, slots=0):
    """Inputs whose MONTGOMERY representation (what the kernels hold: x * 2^261 mod r, as 29-bit limbs) has every limb at
    its maximum 2^29 - 1, laid out over the eight register slots of the wave kernels' first butterfly in the patterns
    that maximise its sums and differences.  With the emulator build the range checks of fpl.h run on exactly these."""
    m = R_MOD
    r_inv = pow(1 << 261, -1, m)
    top = (m >> 232) - 1
    max_mont = (top << 232) | ((1 << 232) - 1)          # < m, limbs 0..7 all 2^29 - 1
    hi, lo = max_mont * r_inv % m, 0                    # the canonical values that upload to those representations
    for log_n in log_ns:
        n = 1 << log_n
        slots_n = slots or (8 if log_n & 1 else 4)      # elements per thread of the wave kernel serving this size (2: the latency forms)
        nt = n // slots_n                               # element j * nt + tid sits in register slot j of thread tid
        half = (1 << (slots_n // 2)) - 1
        for name, mask in (("all", 0xFF), ("low_half", half), ("high_half", half << (slots_n // 2)),
                            ("even", 0x55), ("odd", 0xAA), ("pairs", 0x33 if slots_n == 8 else 0x9), ("one", 0x01), ("seven", 0xFE)):
            v = [hi if (mask >> (i // nt)) & 1 else lo for i in range(n)]
            assert ints(P(v, Basis.MONOMIAL).fft()) == fft_ints(v), ("fft", name, log_n)
            assert ints(P(v, Basis.LAGRANGE).ifft()) == fft_ints(v, True), ("ifft", name, log_n)


def ntt_roundtrip_and_linearity(log_n, seed=5):
    """Size-independent properties for sizes the oracle does not reach in seconds."""
    n = 1 << log_n
    a, b = rand_vec(seed, n), rand_vec(seed + 1, n)
    pa_, pb = P(a, Basis.MONOMIAL), P(b, Basis.MONOMIAL)
    fa, fb = pa_.fft(), pb.fft()
    assert ints(fa.ifft()) == a  # ifft(fft(x)) == x
    s = Scalar(rand_vec(seed + 2, 1)[0])
    lhs = (pa_ * s + pb).fft()
    rhs = fa * s + fb
    assert ints(lhs) == ints(rhs)  # linearity
    # X[0] = sum x_j ; X[n/2] = sum (-1)^j x_j   (w^(n/2) = -1)
    out = ints(fa)
    assert out[0] == sum(a) % R_MOD
    assert out[n // 2] == (sum(a[0::2]) - sum(a[1::2])) % R_MOD
    # a delta at index 1 transforms to the root-of-unity table
    d = [0] * n
    d[1] = 1
    roots = ints(P(d, Basis.MONOMIAL).fft())
    w = ofield.root_of_unity(n)
    assert roots[:4] == [1, w, w * w % R_MOD, pow(w, 3, R_MOD)] and roots[n - 1] == pow(w, n - 1, R_MOD)


def bls_ntt_vs_oracle(log_ns, seed0=40, batch=1):
    """The standalone BLS12-381 Fr transform (plonk_bls_fr_ntt) against the C oracle's oracle_bls_fr_ntt: random inputs,
    inputs at the field's extremes (every element r - 1; Montgomery representations with every limb at 2^29 - 1), forward
    and inverse, in place, and batched."""
    import random

    from oracle import c_oracle
    from plonkathon_amd import bls12_381 as bls

    m = bls.MODULUS
    le = lambda v: b"".join(int(x).to_bytes(32, "little") for x in v)
    r_inv = pow(1 << 261, -1, m)
    max_mont = ((((m >> 232) - 1) << 232) | ((1 << 232) - 1)) * r_inv % m
    for log_n in log_ns:
        n = 1 << log_n
        rng = random.Random(seed0 + log_n)
        slots_n = 8 if log_n & 1 else 4
        nt = n // slots_n
        cases = [("random", [rng.randrange(m) for _ in range(n)]), ("max", [m - 1] * n), ("alt", [(m - 1) * (i & 1) for i in range(n)]),
                 ("limbs_all", [max_mont] * n), ("limbs_even", [max_mont if (0x55 >> (i // nt)) & 1 else 0 for i in range(n)]),
                 ("limbs_low", [max_mont if i // nt < slots_n // 2 else 0 for i in range(n)])]
        if log_n > 13:  # (the emulated two-pass sizes take seconds per transform)
            cases = cases[:1]
        for name, v in cases:
            raw = le(v)
            for inverse in (False, True):
                want = c_oracle.fr_ntt_bytes(raw, inverse, "bls12_381")
                d = bls.upload(raw)
                if log_n <= 13 or not inverse:
                    assert bls.download(bls.ntt(d, log_n, inverse)) == want, (name, log_n, inverse)
                if log_n <= 13 or inverse:
                    assert bls.download(bls.ntt(d, log_n, inverse, out=d)) == want, ("in place", name, log_n, inverse)
        if batch > 1:
            vs = [le([rng.randrange(m) for _ in range(n)]) for _ in range(batch)]
            got = bls.download(bls.ntt(bls.upload(b"".join(vs)), log_n, False, batch))
            assert got == b"".join(c_oracle.fr_ntt_bytes(v, False, "bls12_381") for v in vs), ("batch", log_n)
    # the delta at index 1 transforms to the powers of the crate's root of unity squared down; bad inputs are refused
    n = 1 << log_ns[0]
    roots = bls.ntt_ints([0, 1] + [0] * (n - 2))
    w = bls.root_of_unity(n)
    assert roots[:3] == [1, w, w * w % m] and roots[n - 1] == pow(w, n - 1, m)
    assert bls.root_of_unity(1 << 32) == 0x16A2A19EDFE81F20D09B681922C813B4B63683508C2280B93829971F439F0D2B
    for bad in (lambda: bls.upload(le([m])), lambda: bls.ntt(bls.upload(le([1] * 64)), 6), lambda: bls.ntt(bls.upload(bytes(32 << 7)), 27)):
        try:
            bad()
        except AssertionError:
            pass  # (the host wrapper's own size check)
        except Exception as e:
            assert "BLS12-381" in str(e) or "canonical" in str(e), e
        else:
            raise AssertionError("bad BLS12-381 input accepted")


def bls_golden(max_log_n=13):
    """plonk_bls_fr_* against the committed known-answer vectors (tests/golden/bls12_381_ntt_vectors.json, written by
    tools/gen_bls_vectors.py from the definition in Python integers — independent of the C oracle): fft, ifft and, where the
    fixture has them, the coset forms."""
    import random

    from plonkathon_amd import bls12_381 as bls

    m = bls.MODULUS
    un = lambda raw: [int.from_bytes(raw[32 * i:32 * i + 32], "little") for i in range(len(raw) // 32)]
    for case in load("bls12_381_ntt_vectors.json")["cases"]:
        log_n = case["log_n"]
        if log_n > max_log_n:
            continue
        rng = random.Random(case["seed"])
        raw = b"".join(rng.randrange(m).to_bytes(32, "little") for _ in range(1 << log_n))
        d = bls.upload(raw)
        check_summary(un(bls.download(bls.ntt(d, log_n))), case["fft"])
        check_summary(un(bls.download(bls.ntt(d, log_n, True))), case["ifft"])
        if "offset" in case:
            off = int(case["offset"])
            check_summary(un(bls.download(bls.coset_extend(d, log_n, off))), case["coset_extend"])
            check_summary(un(bls.download(bls.coset_to_coeffs(d, log_n, off))), case["coset_to_coeffs"])


def bls_coset_vs_oracle(log_ns, seed0=70, batch=1):
    """plonk_bls_fr_coset_extend / plonk_bls_fr_coset_to_coeffs against poly.py:156-177 spelled out over the BLS12-381 scalar
    field: coefficients by the C oracle's inverse transform, the offset powers in Python integers, the 4n-point forward transform
    by the C oracle again; then the way back; a non-canonical offset is refused."""
    import random

    from oracle import c_oracle
    from plonkathon_amd import bls12_381 as bls

    m = bls.MODULUS
    le = lambda v: b"".join(int(x).to_bytes(32, "little") for x in v)
    un = lambda raw: [int.from_bytes(raw[32 * i:32 * i + 32], "little") for i in range(len(raw) // 32)]
    for log_n in log_ns:
        n = 1 << log_n
        rng = random.Random(seed0 + log_n)
        off = rng.randrange(2, m)
        vs = [[rng.randrange(m) for _ in range(n)] for _ in range(batch)]
        want = []
        for v in vs:
            coeffs = un(c_oracle.fr_ntt_bytes(le(v), True, "bls12_381"))
            p, scaled = 1, []
            for c in coeffs:
                scaled.append(c * p % m)
                p = p * off % m
            want.append(c_oracle.fr_ntt_bytes(le(scaled + [0] * (3 * n)), False, "bls12_381"))
        d = bls.upload(le([x for v in vs for x in v]))
        big = bls.coset_extend(d, log_n, off, batch)
        assert bls.download(big) == b"".join(want), ("coset_extend", log_n)
        back = bls.coset_to_coeffs(big, log_n + 2, off, batch)        # the coefficients of the degree < n polynomial, zero above
        got = un(bls.download(back))
        for b, v in enumerate(vs):
            coeffs = un(c_oracle.fr_ntt_bytes(le(v), True, "bls12_381"))
            assert got[b * 4 * n:b * 4 * n + n] == coeffs and not any(got[b * 4 * n + n:(b + 1) * 4 * n]), ("coset_to_coeffs", log_n, b)
    try:
        bls.coset_extend(bls.upload(bytes(32 << 8)), 8, m)
    except Exception as e:
        assert "canonical" in str(e), e
    else:
        raise AssertionError("non-canonical offset accepted")


def round_kernels_vs_oracle(log_ns=(3, 4, 6)):
    """The fused round kernels of the C-ABI on their own against the oracle's Polynomial arithmetic (the reference's
    formulas of prover.py:121-146 and 188-203 spelled out operator by operator), plus plonk_fr_powers / plonk_fr_equal."""
    import ctypes

    from plonkathon_amd import get_context
    from plonkathon_amd._lib import check
    from plonkathon_amd.field import le32

    ctx = get_context()
    for log_n in log_ns:
        n = 1 << log_n
        A, B, C, S1, S2, S3 = (rand_vec(900 + 10 * log_n + k, n) for k in range(6))
        beta, gamma, alpha, cof = rand_vec(990 + log_n, 4)
        # a zero denominator factor at row 1 (py_ecc: x / 0 == 0 zeroes the ratio, poly.py:85-100)
        A[1] = (-(beta * S1[1] + gamma)) % R_MOD
        w = ofield.root_of_unity(n)
        roots = [pow(w, i, R_MOD) for i in range(n)]
        Z, closes_want = [1], None
        for i in range(n):
            num = (A[i] + beta * roots[i] + gamma) * (B[i] + 2 * beta * roots[i] + gamma) * (C[i] + 3 * beta * roots[i] + gamma) % R_MOD
            den = (A[i] + beta * S1[i] + gamma) * (B[i] + beta * S2[i] + gamma) * (C[i] + beta * S3[i] + gamma) % R_MOD
            Z.append(Z[-1] * num * ofield.inv(den) % R_MOD)
        closes_want = Z.pop() == 1
        dev = [ctx.upload_ints(v) for v in (A, B, C, S1, S2, S3)]
        out, closes = ctx.alloc(n), ctypes.c_int(-1)
        check(ctx.L.plonk_fr_grand_product(ctx.handle, *[d.ptr for d in dev], log_n, le32(beta), le32(gamma), out.ptr, ctypes.byref(closes)))
        assert ctx.download_ints(out) == Z, ("grand product", log_n)
        assert bool(closes.value) == closes_want
        # quotient on the coset: 14 arbitrary Lagrange vectors -> their extensions -> the fused pass vs operator arithmetic
        lag = [OPoly(rand_vec(700 + 20 * log_n + k, n), OBasis.LAGRANGE) for k in range(14)]
        big = [p.to_coset_extended_lagrange(cof) for p in lag]
        a_, b_, c_, pi, z, ql, qr, qm, qo, qc, s1, s2, s3, l0 = big
        mu = ofield.root_of_unity(4 * n)
        X = OPoly([cof * pow(mu, k, R_MOD) % R_MOD for k in range(4 * n)], OBasis.LAGRANGE)
        ZH = OPoly([(pow(x, n, R_MOD) - 1) % R_MOD for x in X.values], OBasis.LAGRANGE)
        rl = lambda t1, t2: t1 + t2 * beta + gamma
        want = (
            a_ * ql + b_ * qr + a_ * b_ * qm + c_ * qo + pi + qc
            + (rl(a_, X) * rl(b_, X * 2) * rl(c_, X * 3) * z - rl(a_, s1) * rl(b_, s2) * rl(c_, s3) * z.shift(4)) * alpha
            + (z - 1) * l0 * (alpha * alpha % R_MOD)
        ) / ZH
        dbig = [ctx.upload_ints(p.values) for p in big]
        ptrs = (ctypes.c_void_p * 14)(*[d.ptr for d in dbig])
        q = ctx.alloc(4 * n)
        check(ctx.L.plonk_fr_quotient(ctx.handle, log_n, ptrs, le32(cof), le32(alpha), le32(beta), le32(gamma), q.ptr))
        assert ctx.download_ints(q) == want.values, ("quotient", log_n)
        # powers / equal
        assert ints(Polynomial.powers(cof, mu, 4 * n)) == X.values
        same = ctx.upload_ints(want.values)
        eq = ctypes.c_int(-1)
        check(ctx.L.plonk_fr_equal(ctx.handle, q.ptr, same.ptr, 4 * n, ctypes.byref(eq)))
        assert eq.value == 1
        other = list(want.values)
        other[-1] = (other[-1] + 1) % R_MOD
        check(ctx.L.plonk_fr_equal(ctx.handle, q.ptr, ctx.upload_ints(other).ptr, 4 * n, ctypes.byref(eq)))
        assert eq.value == 0
        zeros = P([0] * n + [5] + [0] * (n - 1))
        assert zeros.is_zero(0, n) and not zeros.is_zero(0, n + 1) and zeros.is_zero(n + 1, 2 * n) and zeros.value_at(n).n == 5
        assert P(other) != P(want.values) and P(other) == P(other) and Polynomial._from_device(q, Basis.LAGRANGE, 4 * n) == P(want.values)


def g1_encoding_cases(setup):
    """The compressed G1 / proof encoding: the K6 golden bytes from Prover.prove().to_bytes() and from the lock-step prover's
    device-side packing, round trips, random points and both roots against the oracle's codec, malformed inputs."""
    import random

    from plonkathon_amd import BatchProver, Proof, g1_compress, g1_decompress

    k6 = load("k6_proof.json")
    want = bytes.fromhex(load("k6_proof_bytes.json")["hex"])
    program = Program(k6["program"], k6["group_order"])
    wit = {k: int(v) for k, v in k6["witness"].items()}
    proof = Prover(setup, program).prove(dict(wit))
    blob = proof.to_bytes()
    assert blob == want and len(blob) == 480
    assert flat(Proof.from_bytes(blob)) == flat(proof)
    bp = BatchProver(setup, program)
    bp.upload([dict(wit), dict(wit)])
    bp.run()
    raw, status = bp.download_compressed()
    assert status == b"\0\0" and raw[:480] == want and raw[480:] == want
    # points: multiples of the generator, their negatives (the other root), the identity
    rng = random.Random(77)
    pts = [None, og1.G1, og1.neg(og1.G1)]
    for _ in range(20):
        q = og1.multiply(og1.G1, rng.randrange(1, R_MOD))
        pts += [q, og1.neg(q)]
    enc = g1_compress([None if q is None else (Fq(q[0]), Fq(q[1])) for q in pts])
    assert enc == b"".join(og1.compress(q) for q in pts)
    back = g1_decompress(enc)
    assert [affine(q) for q in back] == [None if q is None else (q[0], q[1]) for q in pts]
    assert [og1.decompress(enc[32 * i : 32 * i + 32]) for i in range(len(pts))] == [None if q is None else (q[0], q[1]) for q in pts]
    # malformed: flag bits 00; x >= p; infinity with x != 0; x with no point on the curve
    bad = [bytes(32), bytes([0x80 | 0x3F]) + b"\xff" * 31, bytes([0x40]) + bytes(30) + b"\x01"]
    x = 1
    while pow((x**3 + 3) % Q_MOD, (Q_MOD - 1) // 2, Q_MOD) == 1:
        x += 1
    bad.append(bytes([0x80]) + x.to_bytes(32, "big")[1:])
    for b in bad:
        try:
            g1_decompress(b)
        except ValueError:
            pass
        else:
            raise AssertionError("decoder accepted %s" % b.hex())
        try:
            og1.decompress(b)
        except ValueError:
            pass
        else:
            raise AssertionError("oracle decoder accepted %s" % b.hex())


def async_upload_and_device_gather(setup, rccl=False):
    """plonk_prover_upload_variables_async from a page-locked buffer gives the proofs of the synchronous upload; a value that
    is not below r marks its proof (status bit 3); plonk_gather_proofs_device through a one-rank communicator returns the
    records and status bytes of plonk_prover_download for several provers in one call (768- and 480-byte forms)."""
    import ctypes

    from plonkathon_amd import BatchProver, get_context
    from plonkathon_amd import distributed as D
    from plonkathon_amd._lib import check
    from plonkathon_amd.batch import _pack_witnesses

    ctx = get_context()
    n = 16
    lines = ["x0 public"] + ["x%d <== x%d * x%d" % (i + 1, i, i) for i in range(n - 1)]
    program = Program(lines, n)
    wits = [program.fill_variable_assignments({"x0": 3 + i}) for i in range(5)]
    a, b = BatchProver(setup, program), BatchProver(setup, program)
    a.upload(wits)
    a.run()
    want, st = a.download_raw()
    assert not any(st)
    V = len(b.variables)
    blob = _pack_witnesses(wits, b.variables, R_MOD)
    pinned = ctx.host_alloc(len(blob))
    pinned[: len(blob)] = blob
    for _ in range(2):  # twice: the second upload must wait for the first batch's gather kernels, not trample them
        b.upload_values_async(pinned, len(wits))
        b.run()
        got, st = b.download_raw()
        assert got == want and not any(st)
    assert b.download_compressed()[0][:480] == BatchProver.decode(want[:768]).to_bytes()
    bad = bytearray(blob)
    bad[32 * (2 * V + 1) : 32 * (2 * V + 2)] = R_MOD.to_bytes(32, "little")  # proof 2, variable 1: r itself
    pinned[: len(blob)] = bytes(bad)
    b.upload_values_async(pinned, len(wits))
    b.run()
    st = b.download_raw()[1]
    assert st[2] & 8 and not any(x & 8 for i, x in enumerate(st) if i != 2)
    # the verdict belongs to the batch it was raised for: a valid batch uploaded next as wire COLUMNS
    # (plonk_prover_upload_witness never runs the checked conversion) must not inherit it (prover.hip)
    b._upload_columns(wits)
    b.run()
    got, st = b.download_raw()
    assert got == want and not any(st), list(st)
    pinned[: len(blob)] = blob
    b.upload_values_async(pinned, len(wits))
    b.run()
    comm = D.RcclComm(ctx, 0, 1) if rccl else None
    if comm is not None:
        a.run()
        gathered, status = D.gather_proofs_device([a, b], len(wits), 2 * len(wits), comm)
        assert gathered.parts[0] == want + want and status == bytes(2 * len(wits)) and gathered.complete()
        out = ctypes.create_string_buffer(2 * len(wits) * 480 + 16)
        handles = (ctypes.c_void_p * 2)(a._h, b._h)
        check(ctx.L.plonk_gather_proofs_device(comm._h, handles, 2, len(wits), 1, out))
        assert out.raw[: 480 * len(wits)] == a.download_compressed()[0]
        comm.close()
    else:
        assert b.download_raw()[0] == want
    ctx.host_free(pinned)


def poly_golden(max_log_n):
    """Every operator of poly.py on the reference-generated vectors (tests/golden/poly_vectors.json)."""
    for case in load("poly_vectors.json")["cases"]:
        log_n, seed = case["log_n"], case["seed"]
        if log_n > max_log_n:
            continue
        n = 1 << log_n
        vals = rand_vec(seed, n)
        lag, mono = P(vals, Basis.LAGRANGE), P(vals, Basis.MONOMIAL)
        check_summary(ints(mono.fft()), case["fft"])
        check_summary(ints(lag.ifft()), case["ifft"])
        if "offset" in case:
            off = Scalar(int(case["offset"]))
            if "coset_extend" in case:
                check_summary(ints(lag.to_coset_extended_lagrange(off)), case["coset_extend"])
            check_summary(ints(lag.coset_extended_lagrange_to_coeffs(off)), case["coset_to_coeffs"])
        if "add" in case:
            other = rand_vec(seed + 500, n)
            if n >= 4:
                other[1] = 0
                other[3] = vals[3]
            olag = P(other)
            sc = Scalar(int(case["scalar"]))
            check_summary(ints(lag + olag), case["add"])
            check_summary(ints(lag - olag), case["sub"])
            check_summary(ints(lag * olag), case["mul"])
            check_summary(ints(lag / olag), case["div"])
            check_summary(ints(lag + sc), case["add_scalar_lagrange"])
            check_summary(ints(lag - sc), case["sub_scalar_lagrange"])
            check_summary(ints(mono + sc), case["add_scalar_monomial"])
            check_summary(ints(mono - sc), case["sub_scalar_monomial"])
            check_summary(ints(lag * sc), case["mul_scalar"])
            check_summary(ints(lag / sc), case["div_scalar"])
            if "shift" in case:
                check_summary(ints(lag.shift(case["shift_k"])), case["shift"])
            assert lag.barycentric_eval(sc).n == int(case["barycentric_at_scalar"])


def poly_asserts():
    import pytest

    a, m = P([1, 2, 3, 4]), P([1, 2, 3, 4], Basis.MONOMIAL)
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
    with pytest.raises(AssertionError):
        a + P([1, 2])  # poly.py:25
    with pytest.raises(AssertionError):
        Polynomial([1, 2], Basis.LAGRANGE)  # poly.py:15 — values must be Scalars
    assert a == P([1, 2, 3, 4]) and not (a == m)
    assert (a / Scalar(0)) == P([0, 0, 0, 0])  # x / 0 == 0
    x = Scalar.roots_of_unity(4)[2]
    assert a.barycentric_eval(x) == OPoly([1, 2, 3, 4], OBasis.LAGRANGE).barycentric_eval(x.n)


def batched_operators(setup, sizes=(8, 64, 2048)):
    """The batch forms of the operators against the single forms and against Python integers: Polynomial.linear_combination
    (plonk_fr_lincomb) = a run of `* Scalar`, `+` (poly.py:23-83), barycentric_eval_many = barycentric_eval per polynomial
    (poly.py:181-195), view = values[start:stop] without a copy, Setup.commit_many = commit / commit_coeffs per polynomial
    (setup.py:66-72) in both bases and for scattered as well as consecutive scalar vectors."""
    import pytest

    for n in sizes:
        vecs = [rand_vec(900 + k, n) for k in range(6)]
        polys = [P(v) for v in vecs]
        scal = rand_vec(950, 6)
        scal[1], scal[2] = 0, R_MOD - 1
        const = rand_vec(951, 1)[0]
        got = Polynomial.linear_combination(list(zip(polys, map(Scalar, scal))), Scalar(const))
        want = [(const + sum(s * v[i] for s, v in zip(scal, vecs))) % R_MOD for i in range(n)]
        assert ints(got) == want and got.basis == Basis.LAGRANGE
        # the same through the operators, term by term
        acc = polys[0] * Scalar(scal[0])
        for p_, s_ in zip(polys[1:], scal[1:]):
            acc = acc + p_ * Scalar(s_)
        assert (acc + Scalar(const)) == got
        assert ints(Polynomial.linear_combination([(polys[3], Scalar(1))])) == vecs[3]
        # twenty terms (the limit), repeated operands
        many = [(polys[k % 6], Scalar(k + 1)) for k in range(20)]
        assert ints(Polynomial.linear_combination(many)) == [sum((k + 1) * vecs[k % 6][i] for k in range(20)) % R_MOD for i in range(n)]
        with pytest.raises(AssertionError):
            Polynomial.linear_combination(many + [(polys[0], Scalar(1))])
        # evaluations: off the domain, ON the domain (x - w^i = 0 for one i: that term counts 0, as py_ecc's x / 0) and at 0
        w = Scalar.root_of_unity(n)
        xs = [Scalar(rand_vec(960 + k, 1)[0]) for k in range(4)] + [w**3, Scalar(0)]
        assert Polynomial.barycentric_eval_many(list(zip(polys, xs))) == [p_.barycentric_eval(x) for p_, x in zip(polys, xs)]
        assert Polynomial.barycentric_eval_many([(polys[2], xs[0])] * 16) == [polys[2].barycentric_eval(xs[0])] * 16
        # views share storage
        big = P(vecs[0] + vecs[1] + vecs[2])
        v1 = big.view(n, 2 * n)
        assert ints(v1) == vecs[1] and len(v1) == n and ints(v1.view(1, 3)) == vecs[1][1:3]
        assert ints(v1 + P(vecs[2])) == [(a + b) % R_MOD for a, b in zip(vecs[1], vecs[2])]
        # commitments: consecutive views (read in place) and scattered buffers (gathered), both bases
        if n <= 64 or n == 2048:
            trio = [big.view(k * n, (k + 1) * n) for k in range(3)]
            assert setup.commit_many(trio) == [setup.commit(p_) for p_ in polys[:3]]
            assert setup.commit_many([polys[4], polys[1]]) == [setup.commit(polys[4]), setup.commit(polys[1])]
            mono = [P(v, Basis.MONOMIAL) for v in vecs[:2]]
            assert setup.commit_many(mono) == [setup.commit_coeffs(m) for m in mono]
            bigm = P(vecs[0] + vecs[1], Basis.MONOMIAL)
            assert setup.commit_many([bigm.view(0, n), bigm.view(n, 2 * n)]) == [setup.commit_coeffs(m) for m in mono]


# ------------------------------------------------------------------------------------------ MSM / commit
def setup_k1():
    sv = load("setup_vectors.json")
    setup = Setup.from_file(PTAU)
    dummy = Polynomial(list(map(Scalar, [1, 2, 3, 4, 5, 6, 7, 8])), Basis.LAGRANGE)
    commitment = setup.commit(dummy)  # test.py:18-28
    assert commitment == (
        16120260411117808045030798560855586501988622612038310041007562782458075125622,
        3125847109934958347271782137825877642397632921923926105820408033549219695465,
    )
    assert affine(setup.powers_of_x[1]) == pt(sv["powers_of_x_1"])
    assert [[str(c.n) for c in setup.X2[0].coeffs], [str(c.n) for c in setup.X2[1].coeffs]] == sv["X2"]
    vk = setup.verification_key(Program(["c <== a * b"], 8).common_preprocessed_input())
    assert vk.w == 19540430494807482326159819597004422086093766032135589407132600596362845576832  # test.py:30-33
    return setup


def _vkey_point(p):
    if p == ["0", "1", "0"]:
        return None
    return (int(p[0]), int(p[1]))


def vkey_goldens(setup):
    for fname, lines in (
        ("main.plonk.vkey.json", ["c <== a * b"]),
        ("main.plonk.vkey-58.json", ["ab === a - c", "-ab === a * b"]),
        ("main.plonk.vkey-59.json", ["c public", "c === a * b"]),
    ):
        theirs = load(fname)
        vk = setup.verification_key(Program(lines, 8).common_preprocessed_input())
        for key in ("Qm", "Ql", "Qr", "Qo", "Qc", "S1", "S2", "S3"):
            assert affine(getattr(vk, key)) == _vkey_point(theirs[key]), (fname, key)
        assert vk.w == int(theirs["w"])


def lincomb_golden(setup, full_size=True):
    lv = load("lincomb_vectors.json")
    Pts = setup.powers_of_x
    for case in lv["cases"]:
        if "seed" in case:
            if not full_size:
                continue
            sc, idx = rand_vec(case["seed"], case["n"]), list(range(case["n"]))
            coeffs = P(sc, Basis.MONOMIAL)
            assert affine(setup.commit_coeffs(coeffs)) == pt(case["result"]), case["name"]
            continue
        sc, idx = [int(s) for s in case["scalars"]], case["points"]
        got = pa.ec_lincomb([(Pts[i], s) for i, s in zip(idx, sc)])
        assert affine(got) == pt(case["result"]), case["name"]
    assert pa.ec_lincomb([(None, 5), (Pts[2], 1)]) == Pts[2]
    # duplicate bases: every addition inside a bucket hits P == +-Q, the case the MSM's fast formulas defer to
    # the bucket reduction (csrc/msm.hip); 40 x the same point with the same scalar, then with cancelling signs
    g2, s7 = affine(Pts[2]), 0x1234567890ABCDEF1234567890ABCDEF
    assert affine(pa.ec_lincomb([(Pts[2], s7)] * 40)) == og1.multiply(g2, 40 * s7 % R_MOD)
    assert pa.ec_lincomb([(Pts[2], s7), (Pts[2], -s7)] * 9) is None
    mixed = [(Pts[2], 3), (Pts[5], 11), (Pts[2], 3), (None, 9), (Pts[5], 11), (Pts[2], 3)]
    assert affine(pa.ec_lincomb(mixed)) == og1.ec_lincomb([(None if p is None else affine(p), k) for p, k in mixed])
    assert pa.ec_mul(Pts[3], 0) is None
    assert affine(pa.ec_mul(Pts[3], Scalar(7))) == og1.multiply(affine(Pts[3]), 7)
    # the generic entry points (curve.py:59-111) on G1 points: the same sums through the MSM
    some = [Pts[1], Pts[4], None, Pts[6]]
    assert affine(pa.lincomb(some, [5, 2**200 + 3, 9, 1])) == og1.ec_lincomb([(None if p is None else affine(p), k) for p, k in zip(some, [5, 2**200 + 3, 9, 1])])
    subs = pa.multisubset(some, [[0, 1], [], [2], [0, 1, 3]])
    assert subs[1] is None and subs[2] is None
    assert affine(subs[0]) == og1.ec_lincomb([(affine(Pts[1]), 1), (affine(Pts[4]), 1)])
    assert affine(subs[3]) == og1.ec_lincomb([(affine(Pts[1]), 1), (affine(Pts[4]), 1), (affine(Pts[6]), 1)])


def msm_vs_oracle(setup, n, seed, batch=1):
    osetup = OSetup.from_file(PTAU)
    for b in range(batch):
        sc = rand_vec(seed + b, n)
        want = og1.ec_lincomb([(osetup.powers_of_x[i], s) for i, s in enumerate(sc)])
        assert affine(setup.commit_coeffs(P(sc, Basis.MONOMIAL))) == want


def msm_linearity(setup, n, seed=314):
    """commit(a) + commit(b) == commit(a + b) and commit(k a) == k commit(a): size-independent properties."""
    a, b = rand_vec(seed, n), rand_vec(seed + 1, n)
    k = rand_vec(seed + 2, 1)[0]
    ca = affine(setup.commit_coeffs(P(a, Basis.MONOMIAL)))
    cb = affine(setup.commit_coeffs(P(b, Basis.MONOMIAL)))
    cab = affine(setup.commit_coeffs(P([(x + y) % R_MOD for x, y in zip(a, b)], Basis.MONOMIAL)))
    cka = affine(setup.commit_coeffs(P([k * x % R_MOD for x in a], Basis.MONOMIAL)))
    assert cab == og1.add(ca, cb)
    assert cka == og1.multiply(ca, k)


def lincomb_fuzz(setup, rounds, seed=2024):
    """Random small linear combinations drawn to hit the corner cases together: repeated and negated bases,
    identity points, zero / tiny / top-of-range scalars, scalars that cancel."""
    import random

    rng = random.Random(seed)
    Pts = setup.powers_of_x
    special = [0, 1, 2, R_MOD - 1, R_MOD - 2, (R_MOD - 1) // 2, 1 << 253, (1 << 16) - 1, 1 << 17, (1 << 34) + 1]
    for _ in range(rounds):
        n = rng.randrange(1, 24)
        pool = [Pts[rng.randrange(0, 12)] for _ in range(rng.randrange(1, 5))]
        pairs = []
        for _ in range(n):
            p = None if rng.random() < 0.1 else rng.choice(pool)
            k = rng.choice(special) if rng.random() < 0.5 else rng.randrange(R_MOD)
            if pairs and rng.random() < 0.2:  # cancel or double the previous term
                p, k = pairs[-1][0], (R_MOD - pairs[-1][1]) % R_MOD if rng.random() < 0.5 else pairs[-1][1]
            pairs.append((p, k))
        want = og1.ec_lincomb([(None if p is None else affine(p), k) for p, k in pairs])
        assert affine(pa.ec_lincomb(pairs)) == want, pairs


def msm_extreme_scalars(setup):
    """Scalars at the top of the range: the signed-digit recoding must not overflow its last window."""
    osetup = OSetup.from_file(PTAU)
    sc = [R_MOD - 1, R_MOD - 2, 1, 0, 1 << 253, (R_MOD - 1) // 2, (1 << 254) % R_MOD, R_MOD - (1 << 200), R_MOD - 1]
    want = og1.ec_lincomb([(osetup.powers_of_x[i], s) for i, s in enumerate(sc)])
    assert affine(setup.commit_coeffs(P(sc, Basis.MONOMIAL))) == want


# ------------------------------------------------------------------------------------------ transcript
def transcript_golden():
    tv = load("transcript_vectors.json")
    t = Transcript(b"test protocol")
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == tv["merlin_simple_vector"]
    g = load("k6_proof.json")["proof"]

    def Pt(k):
        return (pa.Fq(int(g[k][0])), pa.Fq(int(g[k][1])))

    def S(k):
        return Scalar(int(g[k]))

    t = Transcript(b"plonk")
    beta, gamma = t.round_1(pa.Message1(Pt("a_1"), Pt("b_1"), Pt("c_1")))
    alpha, cof = t.round_2(pa.Message2(Pt("z_1")))
    zeta = t.round_3(pa.Message3(Pt("t_lo_1"), Pt("t_mid_1"), Pt("t_hi_1")))
    v = t.round_4(pa.Message4(S("a_eval"), S("b_eval"), S("c_eval"), S("s1_eval"), S("s2_eval"), S("z_shifted_eval")))
    u = t.round_5(pa.Message5(Pt("W_z_1"), Pt("W_zw_1")))
    got = dict(beta=beta, gamma=gamma, alpha=alpha, fft_cofactor=cof, zeta=zeta, v=v, u=u)
    assert {k: str(x.n) for k, x in got.items()} == tv["k6_challenges"]
    t2 = Transcript(b"plonk")
    t2.append_scalar(b"x", Scalar(12345))
    t2.append(b"raw", b"\x00\x01\x02")
    assert str(t2.get_and_append_challenge(b"ch").n) == tv["misc_challenge"]


# ------------------------------------------------------------------------------------------ prover
def flat(proof):
    out = {}
    for k, v in proof.flatten().items():
        out[k] = affine(v) if isinstance(v, tuple) or v is None else v.n
    return out


def prover_k6(setup):
    """prover_test (test.py:136-146) against test/proof.pickle (K6)."""
    k6 = load("k6_proof.json")
    program = Program(k6["program"], k6["group_order"])
    prover = Prover(setup, program)
    proof = flat(prover.prove(dict(k6["witness"])))
    for k, v in k6["proof"].items():
        want = pt(v) if isinstance(v, list) else int(v)
        assert proof[k] == want, k


def prover_vs_oracle(setup, lines, group_order, start, name=""):
    oprog = OProgram(lines, group_order)
    wit = oprog.fill_variable_assignments(start)
    want = OProver(OSetup.from_file(PTAU), oprog).prove(dict(wit)).flatten()
    program = Program(lines, group_order)
    assert program.fill_variable_assignments(start) == wit
    got = flat(Prover(setup, program).prove(dict(wit)))
    assert got == want, name


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
FACTORIZATION_START = {"pb3": 1, "pb2": 1, "pb1": 0, "pb0": 1, "qb3": 0, "qb2": 1, "qb1": 1, "qb0": 1}


def prover_factorization(setup):
    """factorization_test, test.py:171-213."""
    prover_vs_oracle(setup, FACTORIZATION, 16, FACTORIZATION_START, "factorization")


# ------------------------------------------------------------------------------------------ batched prover
def chain_lines(n):
    return ["x0 public"] + ["x%d <== x%d * x%d" % (i + 1, i, i) for i in range(n - 1)]


def batch_prover_k6(setup):
    """The GPU-resident lock-step prover reproduces test/proof.pickle (and the challenges)."""
    k6 = load("k6_proof.json")
    bp = pa.BatchProver(setup, Program(k6["program"], k6["group_order"]))
    proofs = bp.prove_batch([dict(k6["witness"]), dict(k6["witness"])])
    for proof in proofs:
        got = flat(proof)
        for k, v in k6["proof"].items():
            want = pt(v) if isinstance(v, list) else int(v)
            assert got[k] == want, k
    tv = load("transcript_vectors.json")["k6_challenges"]
    for b in (0, 1):
        for k, v in bp.challenges(b).items():
            assert str(v.n) == tv[k], k


def batch_prover_vs_oracle(setup, lines, group_order, starts):
    oprog = OProgram(lines, group_order)
    osetup = OSetup.from_file(PTAU)
    program = Program(lines, group_order)
    wits = [oprog.fill_variable_assignments(s) for s in starts]
    got = [flat(p) for p in pa.BatchProver(setup, program).prove_batch([dict(w) for w in wits])]
    for w, g in zip(wits, got):
        want = OProver(osetup, oprog).prove(dict(w)).flatten()
        assert g == want


def batch_prover_public_input_counts(setup):
    """0, 1, 3 and 8 public inputs take the sparse Lagrange-basis route for PI, 10 the dense transform route."""
    for l in (0, 1, 3, 8, 10):
        pubs = ["p%d public" % i for i in range(l)]
        body = ["q%d <== p%d * p%d" % (i, i, (i + 1) % l) for i in range(l)] if l else []
        lines = pubs + body + ["c <== a * b", "d <== c + a"]
        start = {"a": 5, "b": 7}
        start.update({"p%d" % i: 11 + 3 * i for i in range(l)})
        batch_prover_vs_oracle(setup, lines, 32, [start, dict(start, a=9)])


def batch_prover_rejects_bad_witness(setup):
    import pytest

    program = Program(["e public", "c <== a * b", "e <== c * d"], 8)
    bp = pa.BatchProver(setup, program)
    with pytest.raises(pa.ProofError):
        bp.prove({"a": 3, "b": 4, "c": 13, "d": 5, "e": 65})  # c != a*b: gate constraint fails
    with pytest.raises(KeyError):
        bp.prove({"a": 3, "b": 4})


def batch_prover_fixture_cases(setup, names, batch_copies=1):
    """group_order 2^10 / 2^11 proofs against tests/golden/oracle_proofs.json."""
    fx = {c["name"]: c for c in load("oracle_proofs.json")["cases"]}
    for name in names:
        case = fx[name]
        n = case["group_order"]
        lines = chain_lines(n) if case["program"] == "chain" else poseidon_program_lines()
        program = Program(lines, n)
        wit = program.fill_variable_assignments({k: int(v) for k, v in case["start"].items()})
        bp = pa.BatchProver(setup, program)
        proofs = bp.prove_batch([dict(wit) for _ in range(batch_copies)])
        for b, proof in enumerate(proofs):
            got = flat(proof)
            for k, v in case["proof"].items():
                want = pt(v) if isinstance(v, list) else int(v)
                assert got[k] == want, (name, b, k)
            for k, v in bp.challenges(b).items():
                assert str(v.n) == case["challenges"][k], (name, b, k)


# ------------------------------------------------------------------------------------------ edges / errors
def edge_and_error_paths(setup):
    """Empty / minimal / oversize inputs and the C-ABI's argument checks (mapped to the reference's asserts)."""
    import ctypes
    import pytest
    from plonkathon_amd import get_context
    from plonkathon_amd._lib import check

    ctx = get_context()
    # minimal sizes
    one = P([7], Basis.MONOMIAL)
    assert ints(one.fft()) == [7] and ints(P([7]).ifft()) == [7]
    assert P([7]).barycentric_eval(Scalar(3)) == OPoly([7], OBasis.LAGRANGE).barycentric_eval(3)
    assert ints(P([5, 6]).shift(1)) == [6, 5]
    assert affine(setup.commit_coeffs(P([5], Basis.MONOMIAL))) == og1.multiply(affine(setup.powers_of_x[0]), 5)
    assert setup.commit_coeffs(P([0, 0], Basis.MONOMIAL)) is None  # identity, py_ecc None
    with pytest.raises(ValueError):
        pa.ec_lincomb([])  # curve.py:93 max() of an empty sequence
    # reference asserts
    with pytest.raises(AssertionError):
        P([1, 2, 3], Basis.MONOMIAL).fft()  # not a power of two
    with pytest.raises(AssertionError):
        setup.commit(P([1] * 4096))  # setup.py:70: more coefficients than powers in the SRS
    # C-ABI argument checks surface as AssertionError with a message
    with pytest.raises(AssertionError, match="canonical"):
        ctx.upload_ints([R_MOD])
    buf = ctx.alloc(4)
    with pytest.raises(AssertionError, match="2-adicity"):
        check(ctx.L.plonk_fr_ntt(ctx.handle, buf.ptr, buf.ptr, 29, 0, 1))
    with pytest.raises(AssertionError, match="shift"):
        check(ctx.L.plonk_fr_rotate(ctx.handle, buf.ptr, ctx.alloc(4).ptr, 4, 4))
    with pytest.raises(AssertionError, match="exceeds"):
        xy, fl = ctypes.create_string_buffer(64), ctypes.create_string_buffer(1)
        check(ctx.L.plonk_g1_msm(ctx.handle, setup.device_bases().handle, ctx.alloc(4096).ptr, 4096, 1, 4096, xy, fl))
    with pytest.raises(AssertionError, match="window_bits"):
        check(ctx.L.plonk_msm_configure(ctx.handle, 20, 0))
    # a batch larger than one workgroup row and ragged batch sizes through the lock-step prover
    program = Program(["e public", "c <== a * b", "e <== c * d"], 8)
    bp = pa.BatchProver(setup, program)
    wits = [{"a": 3 + i, "b": 4, "c": (3 + i) * 4, "d": 5, "e": (3 + i) * 20} for i in range(67)]
    proofs = bp.prove_batch(wits)
    osetup = OSetup.from_file(PTAU)
    for i in (0, 1, 63, 64, 66):
        assert flat(proofs[i]) == OProver(osetup, OProgram(["e public", "c <== a * b", "e <== c * d"], 8)).prove(dict(wits[i])).flatten()
    proofs2 = bp.prove_batch(wits[:5])  # smaller batch on the same prover: buffers are reused
    assert [flat(p) for p in proofs2] == [flat(p) for p in proofs[:5]]


def batch_prover_tiny_group_orders(setup):
    """group_order 2 and 4 (the round-5 weights used to alias a buffer that is too small below n = 8)."""
    import pytest

    batch_prover_vs_oracle(setup, ["c <== a * b", "d <== c + a"], 4, [{"a": 3 + i, "b": 4 + i} for i in range(16)])
    # group_order 2: a commitment is the identity, where the reference's transcript raises (transcript.py:65-67 on
    # None; the oracle reproduces that TypeError) -> the batched prover reports it per proof instead of crashing
    lines = ["c <== a * b"]
    with pytest.raises(TypeError):
        OProver(OSetup.from_file(PTAU), OProgram(lines, 2)).prove(OProgram(lines, 2).fill_variable_assignments({"a": 3, "b": 4}))
    bp = pa.BatchProver(setup, Program(lines, 2))
    bp.upload([Program(lines, 2).fill_variable_assignments({"a": 3 + i, "b": 4}) for i in range(16)])
    bp.run()
    assert all(st & 1 for st in bp.download_raw()[1])
    with pytest.raises(pa.ProofError):
        bp.download()


def batch_prover_resident_batch_is_checked(setup):
    """run / download with a batch size other than the resident one must fail (the buffers are laid out for the
    uploaded B), not read the wrong strides."""
    import ctypes
    import pytest
    from plonkathon_amd import _lib

    bp = pa.BatchProver(setup, Program(["e public", "c <== a * b", "e <== c * d"], 8))
    wits = [{"a": 3 + i, "b": 4, "c": (3 + i) * 4, "d": 5, "e": (3 + i) * 20} for i in range(4)]
    bp.upload(wits)
    with pytest.raises(_lib.BackendError, match="resident"):
        bp.run(3)
    bp.run()
    with pytest.raises(_lib.BackendError, match="resident"):
        bp.download_raw(2)
    assert len(bp.download()) == 4
    # the column form of the upload (plonk_prover_upload_witness) gives the same proofs as the per-variable form
    want = bp.download_raw()[0]
    bp._upload_columns(wits)
    bp.run()
    assert bp.download_raw()[0] == want


def msm_deferred_overflow(setup_unused=None):
    """More exceptional additions than the per-MSM deferred list holds (all bases equal: every step after the
    first is a doubling): the MSM must be recomputed by the general-formula kernel, on both methods."""
    from plonkathon_amd import get_context

    ctx = get_context()
    g = (1, 2)
    n = 700  # > MSM_DEFER_CAP = 256 deferred additions in one window
    pairs = [((Fq(1), Fq(2)), 1 + (i % 3)) for i in range(n)]
    want = og1.multiply(g, sum(k for _, k in pairs))
    try:
        ctx.msm_lookup(1)  # bucket method
        assert affine(pa.ec_lincomb(pairs)) == want
        ctx.msm_lookup(2, 4)  # forced 4-bit lookup table over these bases
        assert affine(pa.ec_lincomb(pairs)) == want
        # cancelling pairs: the sum is the identity
        pairs2 = [((Fq(1), Fq(2)), 5)] * 300 + [((Fq(1), Fq(Q_MOD - 2)), 5)] * 300
        assert pa.ec_lincomb(pairs2) is None
        # the same with ONE workgroup per MSM: 256 lanes <= 700 scalars, so the lookup kernel walks scalars t, t + 256, ..
        # (the batch order of round 5) and the deferred list / the recomputation are reached from that order too
        ctx.msm_configure(0, 1)
        assert affine(pa.ec_lincomb(pairs)) == want
        assert pa.ec_lincomb(pairs2) is None
        Pts = [affine(p) for p in Setup.from_file(PTAU).powers_of_x[:300]]
        mixed = [(p, 7 + 3 * i) for i, p in enumerate(Pts)] + [(Pts[5], 11), (Pts[5], R_MOD - 11), (Pts[17], 1)] * 3  # ragged: 309 scalars
        assert affine(pa.ec_lincomb([((Fq(p[0]), Fq(p[1])), k) for p, k in mixed])) == og1.ec_lincomb(mixed)
    finally:
        ctx.msm_lookup(0)
        ctx.msm_configure(0, 0)


def comb_table_shapes(shapes, seed=5150):
    """The comb-table MSM (csrc/msm_comb.h) against the oracle over (teeth h, number of bases n, workgroups per MSM[, top tables]): the lane
    partition of msm_comb_kernel — q = 256 / a lanes per column, left-over lanes crossing columns, pieces per column — depends
    on a = ceil(254 / h) and n alone, so small tables over ARBITRARY bases (forced mode: a table per ec_lincomb call) walk every
    branch of it: n below / at / above the lanes of a column, ragged tails, a > n, one scalar, scalar 0 / 1 / r - 1 / even / odd,
    an identity base, a repeated base (equal pieces in one column's tree: the general-formula kernel takes the MSM)."""
    import random

    from plonkathon_amd import get_context

    ctx = get_context()
    rng = random.Random(seed)
    g = (1, 2)
    special = [0, 1, 2, R_MOD - 1, R_MOD - 2, (R_MOD - 1) // 2, 1 << 253, (1 << 127) + 1]
    try:
        for shape in shapes:
            h, n, groups = shape[:3]
            top = len(shape) > 3 and bool(shape[3])  # the comb of h teeth with top tables (floor(254 / h) columns + joint tables)
            ctx.msm_lookup(2, h, 0, top=top)
            ctx.msm_configure(0, groups)
            ks = [rng.randrange(1, R_MOD) for _ in range(n)]
            pts = [og1.multiply(g, k) for k in ks[: min(n, 24)]]
            pts = [pts[i % len(pts)] for i in range(n)]  # (24 distinct points, repeated: the oracle multiplies in pure Python)
            if n > 3:
                pts[3] = None
            sc = [special[i % len(special)] if i % 3 == 0 else rng.randrange(R_MOD) for i in range(n)]
            pairs = [(None if q is None else (Fq(q[0]), Fq(q[1])), k) for q, k in zip(pts, sc)]
            got = pa.ec_lincomb(pairs)
            want = og1.ec_lincomb(list(zip(pts, sc)))
            assert (None if got is None else affine(got)) == want, (h, n, groups, top)
            # a result that IS the identity: every scalar zero (the recoding turns 0 into r: the sum cancels in the last Horner /
            # butterfly addition, which the finalize kernel resolves itself), and a pair s P + (r - s) P among zeros
            assert pa.ec_lincomb([(q, 0) for q, _ in pairs]) is None, (h, n, groups, "all scalars zero")
            if n >= 2 and pairs[0][0] is not None:
                k = rng.randrange(1, R_MOD)
                canc = [(pairs[0][0], k), (pairs[0][0], R_MOD - k)] + [(q, 0) for q, _ in pairs[2:]]
                assert pa.ec_lincomb(canc) is None, (h, n, groups, "cancelling pair")
    finally:
        ctx.msm_lookup(0)
        ctx.msm_configure(0, 0)


def lagrange_srs_paths(setup):
    """Setup.commit goes through the Lagrange-basis SRS: it must equal ifft + coefficient-basis MSM (setup.py:66-72) at
    several sizes, and a BatchProver committing rounds 1-2 from Lagrange values must produce the same proofs."""
    import random

    rng = random.Random(77)
    for log_n in (0, 1, 3, 6):
        n = 1 << log_n
        vals = [rng.randrange(R_MOD) for _ in range(n)]
        a = setup.commit(P(vals))
        b = setup.commit_coeffs(P(vals).ifft())
        assert affine(a) == affine(b)
        want = og1.ec_lincomb(list(zip([affine(p) for p in setup.powers_of_x[:n]], OPoly(vals, OBasis.LAGRANGE).ifft().values)))
        assert affine(a) == want
    assert setup.commit(P([0] * 8)) is None
    lines, n = ["e public", "c <== a * b", "e <== c * d"], 8
    wits = [{"a": 3 + i, "b": 4, "c": (3 + i) * 4, "d": 5, "e": (3 + i) * 20} for i in range(5)]
    ref = [flat(p) for p in pa.BatchProver(setup, Program(lines, n)).prove_batch(wits)]
    got = [flat(p) for p in pa.BatchProver(setup, Program(lines, n), lagrange_commits=True).prove_batch(wits)]
    assert got == ref


def lagrange_srs_by_ntt(log_ns):
    """the Lagrange-basis SRS by an inverse DFT over the group (g1_ntt.hip: n log n group operations) must be
    the SAME points the n-MSM route gives (both end in the unique affine representative), and commitments over it must equal
    ifft + coefficient-basis MSM (setup.py:66-72) and the oracle."""
    import os
    import random

    rng = random.Random(88)
    osetup = OSetup.from_file(PTAU)
    for log_n in log_ns:
        n = 1 << log_n
        vals = [rng.randrange(R_MOD) for _ in range(n)]
        unit = [[1 if j == i else 0 for j in range(n)] for i in sorted({0, 1 % n, n // 2, n - 1})]
        got = {}
        for route in ("ntt", "msm"):
            os.environ["PLONK_LAGRANGE_SRS"] = route
            try:
                s = Setup.from_file(PTAU)  # a fresh Setup: the view is cached per device copy of the SRS
                got[route] = [affine(s.commit(P(v))) for v in [vals] + unit]
            finally:
                del os.environ["PLONK_LAGRANGE_SRS"]
        assert got["ntt"] == got["msm"], log_n
        if n <= 64:
            coeffs = OPoly(vals, OBasis.LAGRANGE).ifft().values
            assert got["ntt"][0] == og1.ec_lincomb(list(zip(osetup.powers_of_x[:n], coeffs))), log_n


import plonkathon_amd.kzg as pk  # noqa: E402


def lagrange_srs_beyond_2e12(log_n=13):
    """Above 2^12 only the group NTT builds the view (round 4 fell back to ifft + MSM there).  The .ptau slice holds 2^11 points, so
    the base set is synthetic — s_j G for random s_j, made by one batched MSM of size 1 — which is all the transform needs: it is
    linear in the points.  commit(values) over the view == commit_coeffs(ifft(values)), and the first view point is checked
    against the oracle: L_0 = (1/n) sum_j P_j."""
    import ctypes
    import random

    from plonkathon_amd import get_context
    from plonkathon_amd._lib import check
    from plonkathon_amd.kzg import _DeviceBases, _msm

    n = 1 << log_n
    ctx = get_context()
    rng = random.Random(13)
    sc = [rng.randrange(1, R_MOD) for _ in range(n)]
    h = ctypes.c_void_p()
    check(ctx.L.plonk_srs_load_affine(ctx.handle, (1).to_bytes(32, "little") + (2).to_bytes(32, "little"), 1, ctypes.byref(h)))
    gen = _DeviceBases(ctx, h, 1)
    buf = ctx.upload_ints(sc)
    pts = _msm(gen, buf.ptr, 1, n, 1)
    s = Setup(pts)
    vals = [rng.randrange(R_MOD) for _ in range(n)]
    b = s.commit_coeffs(P(vals).ifft())
    if log_n > pk.LAGRANGE_SRS_EAGER_LOG:
        # a one-off commit of a large size takes the reference's route (ifft + one MSM): no view, no staging, no second table
        assert affine(s.commit(P(vals))) == affine(b) and log_n not in s.device_bases()._views
    a = s.commit(P(vals))
    assert log_n in s.device_bases()._views  # a size committed again gets its view (by the group NTT: log_n > 12)
    assert affine(a) == affine(b)
    first = affine(s.commit(P([1] + [0] * (n - 1))))
    assert first == og1.multiply((1, 2), sum(sc) * pow(n, -1, R_MOD) % R_MOD)


def lookup_table_is_shared_across_contexts():
    """One lookup table per (device, SRS): a second context / Setup over the same bytes attaches to it."""
    from plonkathon_amd import Context, Setup

    a, b = Context(0), Context(0)
    sa, sb = Setup.from_file(PTAU), Setup.from_file(PTAU)
    try:
        a.msm_lookup(2, 5)
        b.msm_lookup(2, 5)
        coeffs = list(range(1, 9))
        pa_ = P(coeffs, Basis.MONOMIAL)
        da, db = sa.device_bases(a), sb.device_bases(b)
        import ctypes
        from plonkathon_amd.kzg import _msm

        buf_a, buf_b = a.upload_ints(coeffs), b.upload_ints(coeffs)  # (kept alive across the calls)
        ra = _msm(da, buf_a.ptr, 8, 1, 8)[0]
        rb = _msm(db, buf_b.ptr, 8, 1, 8)[0]
        assert affine(ra) == affine(rb)
        ia, ib = da.lookup_info(), db.lookup_info()
        assert ia["bits"] == ib["bits"] == 5 and ia["bytes"] == ib["bytes"] > 0
        assert ia["sharers"] == ib["sharers"] == 2, (ia, ib)
        del db, sb
        import gc

        gc.collect()
        assert da.lookup_info()["sharers"] == 1
    finally:
        a.msm_lookup(0)
        b.msm_lookup(0)


def lookup_table_colliding_key():
    """ADVICE r03 / VERDICT r04 #7: the registry key is a 64-bit hash, so two DIFFERENT base sets may share it.  With the key
    forced to one value (PLONK_TEST_SRS_KEY) a second SRS with other bases must not attach to the first one's table — it builds
    its own and commits correctly — while a third with the first one's bases still shares it."""
    import os

    from plonkathon_amd import Context, Setup
    from plonkathon_amd.kzg import _msm

    a, b, c = Context(0), Context(0), Context(0)
    sa = Setup.from_file(PTAU)
    pts = sa.powers_of_x[:64]
    other = Setup(list(reversed(pts)))           # the same 64 points in another order: same size, other base set
    same = Setup(list(pts))
    first = Setup(list(pts))
    coeffs = list(range(3, 3 + 64))
    os.environ["PLONK_ENABLE_TEST_HOOKS"] = "1"  # (without it the library ignores PLONK_TEST_SRS_KEY)
    os.environ["PLONK_TEST_SRS_KEY"] = "0x1234"
    try:
        for ctx in (a, b, c):
            ctx.msm_lookup(2, 4)
        d1, d2, d3 = first.device_bases(a), other.device_bases(b), same.device_bases(c)
        bufs = [ctx.upload_ints(coeffs) for ctx in (a, b, c)]
        r1 = _msm(d1, bufs[0].ptr, 64, 1, 64)[0]
        r2 = _msm(d2, bufs[1].ptr, 64, 1, 64)[0]   # same key, other bases: its own table
        r3 = _msm(d3, bufs[2].ptr, 64, 1, 64)[0]   # same key, same bases: shares the first table
        assert affine(r1) == og1.ec_lincomb([(affine(p), k) for p, k in zip(pts, coeffs)])
        assert affine(r2) == og1.ec_lincomb([(affine(p), k) for p, k in zip(reversed(pts), coeffs)])
        assert affine(r3) == affine(r1) and affine(r2) != affine(r1)
        i1, i2, i3 = d1.lookup_info(), d2.lookup_info(), d3.lookup_info()
        assert i1["bits"] == i2["bits"] == i3["bits"] == 4
        assert i1["sharers"] == 2 and i3["sharers"] == 2 and i2["sharers"] == 1, (i1, i2, i3)
    finally:
        del os.environ["PLONK_TEST_SRS_KEY"]
        del os.environ["PLONK_ENABLE_TEST_HOOKS"]
        for ctx in (a, b, c):
            ctx.msm_lookup(0)


# ------------------------------------------------------------------------------------------ product-side verifier
def verifier_cases(setup, full_size=False):
    """The reference's own verifier tests on the product's `VerificationKey` (plonk_pairing_check on the host):
    `verifier_test_unoptimized` / `verifier_test_full` on test/proof.pickle (test.py:59-67, 149-168, 272-275),
    prove -> verify for the factorisation circuit (test.py:171-213), rejection of tampered proofs and wrong public inputs,
    and the pairing itself through bilinearity against oracle-computed multiples."""
    from oracle import pairing as opairing

    g = load("k6_proof.json")

    def Pt(k):
        return (pa.Fq(int(g["proof"][k][0])), pa.Fq(int(g["proof"][k][1])))

    def S(k):
        return Scalar(int(g["proof"][k]))

    golden = pa.Proof(pa.Message1(Pt("a_1"), Pt("b_1"), Pt("c_1")), pa.Message2(Pt("z_1")),
                      pa.Message3(Pt("t_lo_1"), Pt("t_mid_1"), Pt("t_hi_1")),
                      pa.Message4(S("a_eval"), S("b_eval"), S("c_eval"), S("s1_eval"), S("s2_eval"), S("z_shifted_eval")),
                      pa.Message5(Pt("W_z_1"), Pt("W_zw_1")))
    program = Program(g["program"], g["group_order"])
    vk = setup.verification_key(program.common_preprocessed_input())
    public = [int(g["witness"]["e"])]
    assert vk.verify_proof_unoptimized(8, golden, public)  # test.py:59-67
    assert vk.verify_proof(8, golden, public)              # test.py:149-168
    assert not vk.verify_proof(8, golden, [public[0] + 1])
    assert not vk.verify_proof_unoptimized(8, golden, [public[0] + 1])
    import copy

    bad = copy.deepcopy(golden)
    bad.msg_4.b_eval = bad.msg_4.b_eval + 1
    assert not vk.verify_proof(8, bad, public) and not vk.verify_proof_unoptimized(8, bad, public)
    bad = copy.deepcopy(golden)
    bad.msg_5.W_zw_1 = golden.msg_5.W_z_1
    assert not vk.verify_proof(8, bad, public)
    # the same challenges as the prover drew
    tv = load("transcript_vectors.json")["k6_challenges"]
    beta, gamma, alpha, zeta, v, u = vk.compute_challenges(golden)
    assert [str(x.n) for x in (beta, gamma, alpha, zeta, v, u)] == [tv[k] for k in ("beta", "gamma", "alpha", "zeta", "v", "u")]
    # prove -> verify (test.py:171-213)
    fprog = Program(FACTORIZATION, 16)
    fwit = fprog.fill_variable_assignments(FACTORIZATION_START)
    fproof = pa.BatchProver(setup, fprog).prove(dict(fwit))
    fvk = setup.verification_key(fprog.common_preprocessed_input())
    assert fvk.verify_proof(16, fproof, [fwit["n"]]) and fvk.verify_proof_unoptimized(16, fproof, [fwit["n"]])
    assert not fvk.verify_proof(16, fproof, [fwit["n"] + 1])
    # bilinearity: e(aP, bQ) e(-(ab)P, Q) == 1, and non-degeneracy
    a, b = 0x1234567890ABCDEF1234567, R_MOD - 5
    Pa, Pab = og1.multiply((1, 2), a), og1.multiply((1, 2), a * b % R_MOD)
    Qb = opairing.multiply(opairing.G2, b)
    fq2 = lambda q: (pa.kzg.Fq2(q[0].c), pa.kzg.Fq2(q[1].c))
    neg = lambda p: (p[0], (-p[1]) % Q_MOD)
    assert pa.pairing_check([(Pa, fq2(Qb)), (neg(Pab), pa.G2)])
    assert not pa.pairing_check([(Pa, fq2(Qb)), (neg(Pa), pa.G2)])
    assert not pa.pairing_check([((1, 2), pa.G2)])
    assert pa.pairing_check([(None, pa.G2), ((1, 2), pa.G2), (neg((1, 2)), pa.G2)])
    import pytest

    with pytest.raises(AssertionError, match="curve"):
        pa.pairing_check([((1, 3), pa.G2)])
    if full_size:  # Poseidon at group_order 2^10 (test.py:242-259): prove on the GPU, verify with the product's verifier
        lines = poseidon_program_lines()
        pprog = Program(lines, 1024)
        pwit = pprog.fill_variable_assignments({"L0": 1, "M0": 2})
        pproof = pa.BatchProver(setup, pprog).prove(dict(pwit))
        pvk = setup.verification_key(pprog.common_preprocessed_input())
        pub = [pwit[v] for v in pprog.get_public_assignments()]
        assert pvk.verify_proof(1024, pproof, pub) and pvk.verify_proof_unoptimized(1024, pproof, pub)


# ------------------------------------------------------------------------------------------ proofs verify
def proofs_verify(setup, lines, group_order, start, public):
    """Independent acceptance: the GPU's proof passes the oracle's pairing-based verifier against a
    verification key the GPU committed (Setup.verification_key)."""
    from oracle import pairing
    from oracle.verifier import VerificationKey

    program = Program(lines, group_order)
    wit = program.fill_variable_assignments(start)
    proof = flat(pa.BatchProver(setup, program).prove(dict(wit)))
    vk = setup.verification_key(program.common_preprocessed_input())
    x2 = (pairing.FQ2([c.n for c in vk.X_2[0].coeffs]), pairing.FQ2([c.n for c in vk.X_2[1].coeffs]))
    ovk = VerificationKey(group_order, *[affine(getattr(vk, k)) for k in ("Qm", "Ql", "Qr", "Qo", "Qc", "S1", "S2", "S3")],
                          x2, vk.w.n)
    pub = [wit[v] for v in public]
    assert ovk.verify_proof(group_order, proof, pub)
    bad = dict(proof)
    bad["b_eval"] = (bad["b_eval"] + 1) % R_MOD
    assert not ovk.verify_proof(group_order, bad, pub)
