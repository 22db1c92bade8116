"""Polynomial algebra over BN254 Fr on Python ints.  (oracle: test infrastructure only)

Restates /root/reference/poly.py line by line, keeping its algorithmic shape (recursive
even/odd radix-2 FFT on raw ints, per-element inversions in `/` and `barycentric_eval`) so the
timed CPU baseline has the reference's cost profile.  Values are canonical ints, not objects.
"""
from enum import Enum

from .field import R_MOD, inv, root_of_unity, roots_of_unity


class Basis(Enum):  # poly.py:5-7
    LAGRANGE = 1
    MONOMIAL = 2


def _fft(vals, modulus, roots):
    """poly.py:117-127 — recursive DIT; natural order in and out."""
    if len(vals) == 1:
        return vals
    L = _fft(vals[::2], modulus, roots[::2])
    R = _fft(vals[1::2], modulus, roots[::2])
    o = [0] * len(vals)
    for i, (x, y) in enumerate(zip(L, R)):
        y_times_root = y * roots[i]
        o[i] = (x + y_times_root) % modulus
        o[i + len(L)] = (x - y_times_root) % modulus
    return o


def fft_ints(vals, inverse=False):
    """poly.py:113-145 on a bare list of ints (no basis bookkeeping)."""
    n = len(vals)
    roots = roots_of_unity(n)  # poly.py:129 — recomputed on every call, as the reference does
    nvals = [v % R_MOD for v in vals]
    if inverse:
        invlen = inv(n)  # poly.py:134
        rev = [roots[0]] + roots[1:][::-1]  # poly.py:135
        return [x * invlen % R_MOD for x in _fft(nvals, R_MOD, rev)]
    return _fft(nvals, R_MOD, roots)


class Polynomial:
    """poly.py:10-21.  `values` are ints in [0, r)."""

    def __init__(self, values, basis):
        assert isinstance(basis, Basis)
        self.values = [int(v) % R_MOD for v in values]
        self.basis = basis

    def __eq__(self, other):
        return self.basis == other.basis and self.values == other.values

    # poly.py:23-65 — poly±poly pointwise; ±scalar broadcasts in LAGRANGE, touches only the
    # constant term in MONOMIAL.
    def _addsub(self, other, sign):
        if isinstance(other, Polynomial):
            assert len(self.values) == len(other.values)
            assert self.basis == other.basis
            return Polynomial(
                [(x + sign * y) % R_MOD for x, y in zip(self.values, other.values)], self.basis
            )
        other = int(other)
        if self.basis == Basis.LAGRANGE:
            return Polynomial([(x + sign * other) % R_MOD for x in self.values], self.basis)
        return Polynomial([(self.values[0] + sign * other) % R_MOD] + self.values[1:], self.basis)

    def __add__(self, other):
        return self._addsub(other, 1)

    def __sub__(self, other):
        return self._addsub(other, -1)

    def __mul__(self, other):  # poly.py:68-83
        if isinstance(other, Polynomial):
            assert self.basis == Basis.LAGRANGE
            assert self.basis == other.basis
            assert len(self.values) == len(other.values)
            return Polynomial([x * y % R_MOD for x, y in zip(self.values, other.values)], self.basis)
        other = int(other)
        return Polynomial([x * other % R_MOD for x in self.values], self.basis)

    def __truediv__(self, other):  # poly.py:85-100 — one inversion per element, x/0 == 0
        if isinstance(other, Polynomial):
            assert self.basis == Basis.LAGRANGE
            assert self.basis == other.basis
            assert len(self.values) == len(other.values)
            return Polynomial([x * inv(y) % R_MOD for x, y in zip(self.values, other.values)], self.basis)
        other = int(other)
        return Polynomial([x * inv(other) % R_MOD for x in self.values], self.basis)

    def shift(self, shift):  # poly.py:102-109
        assert self.basis == Basis.LAGRANGE
        assert shift < len(self.values)
        return Polynomial(self.values[shift:] + self.values[:shift], self.basis)

    def fft(self, inv=False):  # poly.py:113-145
        if inv:
            assert self.basis == Basis.LAGRANGE
            return Polynomial(fft_ints(self.values, True), Basis.MONOMIAL)
        assert self.basis == Basis.MONOMIAL
        return Polynomial(fft_ints(self.values, False), Basis.LAGRANGE)

    def ifft(self):  # poly.py:147-148
        return self.fft(True)

    def to_coset_extended_lagrange(self, offset):  # poly.py:156-163
        assert self.basis == Basis.LAGRANGE
        offset = int(offset)
        group_order = len(self.values)
        x_powers = self.ifft().values
        x_powers = [pow(offset, i, R_MOD) * x % R_MOD for i, x in enumerate(x_powers)] + [0] * (
            group_order * 3
        )
        return Polynomial(x_powers, Basis.MONOMIAL).fft()

    def coset_extended_lagrange_to_coeffs(self, offset):  # poly.py:169-177
        assert self.basis == Basis.LAGRANGE
        shifted_coeffs = self.ifft().values
        inv_offset = inv(int(offset))
        return Polynomial(
            [v * pow(inv_offset, i, R_MOD) % R_MOD for i, v in enumerate(shifted_coeffs)],
            Basis.MONOMIAL,
        )

    def barycentric_eval(self, x):  # poly.py:181-195
        assert self.basis == Basis.LAGRANGE
        x = int(x) % R_MOD
        order = len(self.values)
        roots = roots_of_unity(order)
        s = 0
        for value, root in zip(self.values, roots):
            s += value * root % R_MOD * inv(x - root)
        return (pow(x, order, R_MOD) - 1) * inv(order) % R_MOD * (s % R_MOD) % R_MOD
