"""Host build (g++, tests/emu) of the device field / curve primitives against Python integers:
fp.h (packed, 9x29-bit lazy-carry product scanning, R = 2^261), g1.h (XYZZ formulas incl. the
exceptional cases) and the lazy-limb accumulator of fpl.h."""
import ctypes
import os
import random
import subprocess

import pytest

from conftest import EMU_DIR
from helpers import GOLDEN
from oracle import field, g1
from oracle.srs import Setup

R261 = 1 << 261


@pytest.fixture(scope="module")
def probe():
    subprocess.run(["make", "-s", "-C", EMU_DIR, "libfp_probe.so"], check=True)
    return ctypes.CDLL(os.path.join(EMU_DIR, "libfp_probe.so"))


def _w(x):
    return (ctypes.c_uint32 * 8)(*[(x >> (32 * i)) & 0xFFFFFFFF for i in range(8)])


def _f(o, off=0):
    return sum(int(o[off + i]) << (32 * i) for i in range(8))


def _call(fn, op, a, b=0):
    out = (ctypes.c_uint32 * 8)()
    fn(op, _w(a), _w(b), out)
    return _f(out)


BLS_R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001  # BLS12-381 Fr (ntt_bls.hip): 255 bits, 4m > 2^256


@pytest.mark.parametrize("name,m", [("probe_fr", field.R_MOD), ("probe_fq", field.Q_MOD), ("probe_bls_fr", BLS_R)])
def test_field_ops(probe, name, m):
    fn = getattr(probe, name)
    Rm, Ri = R261 % m, pow(R261 % m, -1, m)
    rng = random.Random(1)
    edge = [0, 1, 2, m - 1, m - 2, Rm, (m - 1) // 2, (1 << 253) % m]
    for it in range(1500):
        a = rng.choice(edge) if it < 64 and it % 2 else rng.randrange(m)
        b = rng.choice(edge) if it < 64 else rng.randrange(m)
        assert _call(fn, 0, a, b) == (a + b) % m
        assert _call(fn, 1, a, b) == (a - b) % m
        assert _call(fn, 2, a, b) == a * b * Ri % m
        assert _call(fn, 7, a) == a * a * Ri % m
        assert _call(fn, 4, a) == a * Rm % m
        assert _call(fn, 5, a) == a * Ri % m
        assert _call(fn, 6, a) == (-a) % m
        want_inv = field.inv(a, m) * Rm % m  # inverse of 0 is 0
        assert _call(fn, 3, a * Rm % m) == want_inv  # division-step inversion
        if it < 40:
            assert _call(fn, 8, a * Rm % m) == want_inv  # Fermat cross-check
    # inputs that make the division steps take unusually many / few rounds
    for a in [1 << k for k in range(0, 254, 7)] + [m - (1 << k) for k in range(0, 254, 11)] + [3, m // 3, (m + 1) // 2]:
        assert _call(fn, 3, a % m * Rm % m) == field.inv(a % m, m) * Rm % m


def _pt_words(p, m, Rm):
    x, y = (0, 0) if p is None else (p[0] * Rm % m, p[1] * Rm % m)
    return [(x >> (32 * i)) & 0xFFFFFFFF for i in range(8)] + [(y >> (32 * i)) & 0xFFFFFFFF for i in range(8)]


def _pt_out(out, m, Ri):
    x, y = _f(out) * Ri % m, _f(out, 8) * Ri % m
    return None if x == 0 and y == 0 else (x, y)


def test_g1_formulas_and_exceptional_cases(probe):
    m = field.Q_MOD
    Rm, Ri = R261 % m, pow(R261 % m, -1, m)
    G = g1.G1
    pts = [None, G, g1.multiply(G, 2), g1.multiply(G, 3), g1.neg(G), g1.multiply(G, 123456789), g1.neg(g1.multiply(G, 2))]
    for p in pts:
        for q in pts:
            for op, want in ((0, g1.add(p, q)), (1, g1.add(p, q)), (2, g1.double(p)),
                             (3, g1.add(g1.double(p), q)), (4, g1.add(g1.double(p), g1.double(q)))):
                out = (ctypes.c_uint32 * 16)()
                probe.probe_g1(op, (ctypes.c_uint32 * 16)(*_pt_words(p, m, Rm)), (ctypes.c_uint32 * 16)(*_pt_words(q, m, Rm)), out)
                assert _pt_out(out, m, Ri) == want, (op, p, q)


def test_lazy_accumulator_chain(probe):
    m = field.Q_MOD
    Rm, Ri = R261 % m, pow(R261 % m, -1, m)
    P = Setup.from_file(os.path.join(GOLDEN, "srs_2048.ptau")).powers_of_x
    rng = random.Random(5)

    def run(pts, negs):
        arr = (ctypes.c_uint32 * (16 * len(pts)))(*sum([_pt_words(p, m, Rm) for p in pts], []))
        out = (ctypes.c_uint32 * 16)()
        probe.probe_g1l_chain(arr, (ctypes.c_int * len(pts))(*negs), len(pts), out)
        return _pt_out(out, m, Ri)

    def ref(pts, negs):
        acc = None
        for p, n in zip(pts, negs):
            acc = g1.add(acc, g1.neg(p) if n else p)
        return acc

    for _ in range(12):
        n = rng.randrange(1, 300)
        pts = [P[rng.randrange(2048)] for _ in range(n)]
        negs = [rng.randrange(2) for _ in range(n)]
        assert run(pts, negs) == ref(pts, negs)
    G = P[0]
    for pts, negs in (([G, G], [0, 0]), ([G, G], [0, 1]), ([G, G, P[1]], [0, 1, 0]), ([G] * 5, [0] * 5),
                      ([P[3], P[4], P[3], P[4]], [0, 0, 1, 1]), ([P[5]] * 9, [0] * 9), ([None, G, None], [0, 0, 0]),
                      # a negated point into the empty accumulator, then additions: R = S2 - Y1 with Y1 = -y (the emulator
                      # build's range checks in fpl.h abort if Y1 is left un-normalised)
                      ([G, P[1], P[2]], [1, 0, 1]), ([P[7], P[8], P[9], P[10]], [1, 1, 0, 0]), ([None, P[6], P[5]], [0, 1, 0])):
        assert run(pts, negs) == ref(pts, negs)


def test_signed_limb_arithmetic_at_the_bounds(probe):
    """fpl.h on raw signed limbs at the edges of its documented operand ranges (differences of normalised values with
    every limb at +-(2^29 - 1), values up to the 128 m^2 product bound): congruence, output range and normalisation."""
    m = field.Q_MOD
    Ri = pow(R261 % m, -1, m)
    L = (1 << 29) - 1
    rng = random.Random(9)
    I9 = ctypes.c_int32 * 9

    def val(l):
        return sum(int(v) << (29 * i) for i, v in enumerate(l))

    def limbs_of(v):  # normalised limbs of a (possibly negative) value
        out = []
        for i in range(8):
            out.append(v & L)
            v >>= 29
        return out + [v]

    def diff_operand(kind, top):
        """difference of two normalised values: limbs 0..7 within [-L, L]; limb 8 sets the size (|value| ~ top * m)"""
        if kind == 0:
            lo = [L] * 8
        elif kind == 1:
            lo = [-L] * 8
        elif kind == 2:
            lo = [L if i % 2 else -L for i in range(8)]
        else:
            lo = [rng.randrange(-L, L + 1) for _ in range(8)]
        return lo + [(top * m) >> 232]

    def call(op, a, b=None, c=None, d=None):
        z = [0] * 9
        out = I9()
        probe.probe_fpl(op, I9(*a), I9(*(b or z)), I9(*(c or z)), I9(*(d or z)), out)
        return list(out)

    def check_product(r, want):
        assert all(0 <= v <= L for v in r[:8]), r
        assert -m < val(r) < 2 * m
        assert val(r) % m == want % m

    tops = [-9, -6, -3, -1, 0, 1, 2, 5, 9]
    for ka in range(4):
        for kb in range(4):
            for ta in tops:
                for tb in tops:
                    a, b = diff_operand(ka, ta), diff_operand(kb, tb)
                    if abs(val(a)) * abs(val(b)) > 128 * m * m:
                        continue
                    check_product(call(0, a, b), val(a) * val(b) * Ri)
                    if ta == tb and ka == kb:
                        check_product(call(1, a), val(a) * val(a) * Ri)
                    c, d = diff_operand(kb, -tb), diff_operand(ka, ta)
                    if abs(val(a) * val(b)) + abs(val(c) * val(d)) <= 128 * m * m:
                        check_product(call(2, a, b, c, d), (val(a) * val(b) + val(c) * val(d)) * Ri)
    # one operand a SUM of two normalised values (limbs to 2^30), the other normalised
    a = [2 * L] * 8 + [(3 * m) >> 232]
    b = [L] * 8 + [(2 * m) >> 232]
    check_product(call(0, a, b), val(a) * val(b) * Ri)
    # carry sweep of the X3 shape: r2 - ppp - 2q
    for _ in range(200):
        x = [rng.randrange(-3 * L, L + 1) for _ in range(8)] + [rng.randrange(-(1 << 26), 1 << 26)]
        r = call(3, x)
        assert all(0 <= v <= L for v in r[:8]) and val(r) == val(x)
    # canonical conversion and the piece form, over the accumulator's ranges
    for lo, hi, op in ((-127, 127, 4), (-7, 5, 5), (-1, 2, 6)):
        for _ in range(200):
            v = rng.randrange(lo * m + 1, hi * m)
            out = call(op, limbs_of(v))
            got = sum((out[i] & 0xFFFFFFFF) << (32 * i) for i in range(8))
            if op == 4:
                assert got == v % m
            else:
                assert got % m == v % m and 0 <= got < 4 * m
        for v in (lo * m + 1, hi * m - 1, 0, -1, 1, m, -m):
            if lo * m < v < hi * m:
                out = call(op, limbs_of(v))
                got = sum((out[i] & 0xFFFFFFFF) << (32 * i) for i in range(8))
                assert got % m == v % m


@pytest.mark.parametrize("fn_name,m,gen,reach", [("probe_fpl_shoup", field.R_MOD, 5, 128), ("probe_fpl_shoup_bls", BLS_R, 7, 56)])
def test_shoup_multiplication_by_a_constant(probe, fn_name, m, gen, reach):
    """fpl_mul_shoup (the NTT kernels' twiddle multiplication): the Shoup pair (w, floor(w 2^261 / r)) derived from the
    Montgomery form of a constant, and a * w for operands across the documented range (|a| < 128 r for BN254, < 56 r for
    BLS12-381 Fr where R / r is 70.7 instead of 169) — congruence mod r, result within (-1.8 r, 2.8 r), limbs normalised."""
    L = (1 << 29) - 1
    rng = random.Random(11)
    I9, U8, I27 = ctypes.c_int32 * 9, ctypes.c_uint32 * 8, ctypes.c_int32 * 27

    def val(l):
        return sum(int(v) << (29 * i) for i, v in enumerate(l))

    consts = [0, 1, 2, m - 1, m - 2, (m - 1) // 2, gen, pow(gen, (m - 1) // 2048, m)] + [rng.randrange(m) for _ in range(24)]
    for w in consts:
        wt = w * R261 % m
        words = U8(*[(wt >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
        operands = []
        for top in (-120, -55, -17, -2, -1, 0, 1, 2, 16, 55, 127):  # |value| up to 128 r
            for kind in range(4):
                lo = [L] * 8 if kind == 0 else [-L] * 8 if kind == 1 else [2 * L if i % 2 else -L for i in range(8)] if kind == 2 else [rng.randrange(-L, 2 * L) for _ in range(8)]
                operands.append(lo + [(top * m) >> 232])
        operands.append([int(1.26 * (1 << 30))] * 8 + [0])   # the largest limbs fpl_mul's callers produce
        operands.append([-int(1.26 * (1 << 30))] * 8 + [0])
        for a in operands:
            if abs(val(a)) >= reach * m:
                continue
            out = I27()
            getattr(probe, fn_name)(I9(*a), words, out)
            out = list(out)
            r, wl, wp = out[:9], out[9:18], out[18:]
            assert val(wl) == w and val(wp) == (w << 261) // m
            assert all(0 <= v <= L for v in r[:8]), r
            assert val(r) % m == val(a) * w % m
            assert -1.8 * m < val(r) < 2.8 * m
