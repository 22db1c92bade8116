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


@pytest.mark.parametrize("name,m", [("probe_fr", field.R_MOD), ("probe_fq", field.Q_MOD)])
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
                      ([P[3], P[4], P[3], P[4]], [0, 0, 1, 1]), ([P[5]] * 9, [0] * 9), ([None, G, None], [0, 0, 0])):
        assert run(pts, negs) == ref(pts, negs)
