"""`Polynomial` — the reference's Fr-vector type (/root/reference/poly.py:10-195) with the values
resident in MI355X HBM and every operation executed by libplonk_hip.so.

Same constructor, `basis` semantics, operators, asserts and method names as the reference, so code
written against `poly.Polynomial` runs unchanged.  `values` is materialised lazily as a
`list[Scalar]` only when host code reads it.
"""
import ctypes
from enum import Enum

from . import _lib
from ._lib import OP_ADD, OP_DIV, OP_MUL, OP_SUB, check
from .backend import DeviceBuffer, DeviceView, get_context
from .field import Scalar, le32


class Basis(Enum):  # poly.py:5-7
    LAGRANGE = 1
    MONOMIAL = 2


def _log2_exact(n):
    assert n >= 1 and n & (n - 1) == 0, "length must be a power of two"
    return n.bit_length() - 1


class Polynomial:
    def __init__(self, values, basis):  # poly.py:14-18
        assert all(isinstance(x, Scalar) for x in values)
        assert isinstance(basis, Basis)
        self._values = list(values)
        self._dev = None
        self._n = len(self._values)
        self.basis = basis

    # ---- construction from / access to device storage ---------------------------------------
    @classmethod
    def _from_device(cls, buf: DeviceBuffer, basis: Basis, n=None):
        p = cls.__new__(cls)
        p._values = None
        p._dev = buf
        p._n = buf.n if n is None else n
        p.basis = basis
        return p

    @classmethod
    def from_ints(cls, ints, basis):
        """Convenience constructor from plain ints (reduced mod r)."""
        return cls([Scalar(int(v)) for v in ints], basis)

    @property
    def values(self):
        if self._values is None:
            self._values = [Scalar(v) for v in get_context().download_ints(self._dev, self._n)]
        return self._values

    @values.setter
    def values(self, v):
        self._values = list(v)
        self._n = len(self._values)
        self._dev = None

    def __len__(self):
        return self._n

    def device(self) -> DeviceBuffer:
        if self._dev is None:
            self._dev = get_context().upload_ints([x.n for x in self._values])
        return self._dev

    @classmethod
    def from_bytes(cls, raw, basis):
        """From canonical 32-byte little-endian elements back to back (no per-element Python objects)."""
        return cls._from_device(get_context().upload_bytes(raw), basis)

    @classmethod
    def powers(cls, first, base, n, basis=Basis.LAGRANGE):
        """[first * base^k for k < n] built on the device: Scalar.roots_of_unity (curve.py:19-24), the coset points
        fft_cofactor * mu^k of prover.py:160-161."""
        ctx = get_context()
        out = ctx.alloc(n)
        check(ctx.L.plonk_fr_powers(ctx.handle, le32(Scalar(first).n), le32(Scalar(base).n), n, out.ptr))
        return cls._from_device(out, basis, n)

    def value_at(self, i):
        """values[i] without materialising the rest."""
        if self._values is not None:
            return self._values[i]
        return Scalar(get_context().download_ints(self._dev, 1, offset=i)[0])

    def is_zero(self, start=0, stop=None):
        """values[start:stop] == [0] * (stop - start), decided on the device (prover.py:205-208, 288, 299)."""
        stop = self._n if stop is None else stop
        ctx = get_context()
        eq = ctypes.c_int(0)
        check(ctx.L.plonk_fr_equal(ctx.handle, self.device().at(start), None, stop - start, ctypes.byref(eq)))
        return bool(eq.value)

    def __eq__(self, other):  # poly.py:20-21
        if self.basis != other.basis:
            return False
        if self._values is None or other._values is None:  # at least one side lives in HBM: compare there
            if self._n != other._n:
                return False
            ctx = get_context()
            eq = ctypes.c_int(0)
            check(ctx.L.plonk_fr_equal(ctx.handle, self.device().ptr, other.device().ptr, self._n, ctypes.byref(eq)))
            return bool(eq.value)
        return self.values == other.values

    # ---- pointwise operators ------------------------------------------------------------------
    def _binary(self, other, op):
        ctx = get_context()
        out = ctx.alloc(self._n)
        check(ctx.L.plonk_fr_pointwise(ctx.handle, op, self.device().ptr, other.device().ptr, out.ptr, self._n))
        return Polynomial._from_device(out, self.basis)

    def _scalar(self, other, op, constant_term_only=False):
        ctx = get_context()
        out = ctx.alloc(self._n)
        check(ctx.L.plonk_fr_scalar_op(ctx.handle, op, self.device().ptr, le32(other.n), out.ptr, self._n,
                                       1 if constant_term_only else 0))
        return Polynomial._from_device(out, self.basis)

    def __add__(self, other):  # poly.py:23-43
        if isinstance(other, Polynomial):
            assert len(self) == len(other)
            assert self.basis == other.basis
            return self._binary(other, OP_ADD)
        assert isinstance(other, Scalar)
        return self._scalar(other, OP_ADD, self.basis != Basis.LAGRANGE)

    def __sub__(self, other):  # poly.py:45-65
        if isinstance(other, Polynomial):
            assert len(self) == len(other)
            assert self.basis == other.basis
            return self._binary(other, OP_SUB)
        assert isinstance(other, Scalar)
        return self._scalar(other, OP_SUB, self.basis != Basis.LAGRANGE)

    def __mul__(self, other):  # poly.py:68-83
        if isinstance(other, Polynomial):
            assert self.basis == Basis.LAGRANGE
            assert self.basis == other.basis
            assert len(self) == len(other)
            return self._binary(other, OP_MUL)
        assert isinstance(other, Scalar)
        return self._scalar(other, OP_MUL)

    def __truediv__(self, other):  # poly.py:85-100
        if isinstance(other, Polynomial):
            assert self.basis == Basis.LAGRANGE
            assert self.basis == other.basis
            assert len(self) == len(other)
            return self._binary(other, OP_DIV)
        assert isinstance(other, Scalar)
        return self._scalar(other, OP_DIV)

    def shift(self, shift: int):  # poly.py:102-109
        assert self.basis == Basis.LAGRANGE
        assert shift < len(self)
        ctx = get_context()
        out = ctx.alloc(self._n)
        check(ctx.L.plonk_fr_rotate(ctx.handle, self.device().ptr, out.ptr, self._n, shift))
        return Polynomial._from_device(out, self.basis)

    # ---- NTT family ---------------------------------------------------------------------------
    def fft(self, inv=False):  # poly.py:113-145
        if inv:
            assert self.basis == Basis.LAGRANGE
        else:
            assert self.basis == Basis.MONOMIAL
        ctx = get_context()
        out = ctx.alloc(self._n)
        check(ctx.L.plonk_fr_ntt(ctx.handle, self.device().ptr, out.ptr, _log2_exact(self._n), 1 if inv else 0, 1))
        return Polynomial._from_device(out, Basis.MONOMIAL if inv else Basis.LAGRANGE)

    def ifft(self):  # poly.py:147-148
        return self.fft(True)

    def to_coset_extended_lagrange(self, offset):  # poly.py:156-163
        assert self.basis == Basis.LAGRANGE
        offset = Scalar(offset)
        ctx = get_context()
        out = ctx.alloc(4 * self._n)
        check(ctx.L.plonk_fr_coset_extend(ctx.handle, self.device().ptr, out.ptr, _log2_exact(self._n), le32(offset.n), 1))
        return Polynomial._from_device(out, Basis.LAGRANGE)

    def coset_extended_lagrange_to_coeffs(self, offset):  # poly.py:169-177
        assert self.basis == Basis.LAGRANGE
        offset = Scalar(offset)
        ctx = get_context()
        out = ctx.alloc(self._n)
        check(ctx.L.plonk_fr_coset_to_coeffs(ctx.handle, self.device().ptr, out.ptr, _log2_exact(self._n), le32(offset.n), 1))
        return Polynomial._from_device(out, Basis.MONOMIAL)

    def barycentric_eval(self, x):  # poly.py:181-195
        assert self.basis == Basis.LAGRANGE
        x = Scalar(x)
        ctx = get_context()
        out = ctypes.create_string_buffer(32)
        check(ctx.L.plonk_fr_barycentric(ctx.handle, self.device().ptr, _log2_exact(self._n), le32(x.n), out))
        return Scalar(int.from_bytes(out.raw, "little"))

    # ---- batches of the operations above (one launch / one synchronisation for a run of them) ------------------
    @classmethod
    def barycentric_eval_many(cls, pairs):
        """[p.barycentric_eval(x) for p, x in pairs] (poly.py:181-195) for up to 16 polynomials of one size: one kernel and one
        host synchronisation instead of one round trip per evaluation (plonk_fr_barycentric_many)."""
        pairs = [(p, Scalar(x)) for p, x in pairs]
        assert pairs and all(p.basis == Basis.LAGRANGE and len(p) == len(pairs[0][0]) for p, _ in pairs)
        ctx = get_context()
        ptrs = (ctypes.c_void_p * len(pairs))(*[p.device().ptr.value for p, _ in pairs])
        out = ctypes.create_string_buffer(32 * len(pairs))
        check(ctx.L.plonk_fr_barycentric_many(ctx.handle, len(pairs), ptrs, _log2_exact(len(pairs[0][0])), b"".join(le32(x.n) for _, x in pairs), out))
        return [Scalar(int.from_bytes(out.raw[32 * k : 32 * k + 32], "little")) for k in range(len(pairs))]

    @classmethod
    def linear_combination(cls, terms, constant=0):
        """constant + sum(p * s for p, s in terms) with the reference's operator semantics (poly.py:23-83: every p in the
        LAGRANGE basis, so a Scalar addend is added to every value) in ONE pass over the data (plonk_fr_lincomb; up to 20 terms)
        — what a run of `Polynomial * Scalar`, `+`, `-` such as prover.py:245-288 computes with a launch per operator."""
        terms = [(p, Scalar(s)) for p, s in terms]
        assert terms and all(p.basis == Basis.LAGRANGE and len(p) == len(terms[0][0]) for p, _ in terms)
        ctx = get_context()
        n = len(terms[0][0])
        out = ctx.alloc(n)
        ptrs = (ctypes.c_void_p * len(terms))(*[p.device().ptr.value for p, _ in terms])
        check(ctx.L.plonk_fr_lincomb(ctx.handle, len(terms), ptrs, b"".join(le32(s.n) for _, s in terms), le32(Scalar(constant).n), out.ptr, n))
        return cls._from_device(out, Basis.LAGRANGE, n)

    def view(self, start, stop, basis=None):
        """values[start:stop] as a Polynomial that SHARES this one's device storage (no copy; read-only by convention, as every
        operator returns a new Polynomial)."""
        return Polynomial._from_device(DeviceView(self.device(), start, stop - start), basis or self.basis, stop - start)

    # ---- helpers used by the prover -------------------------------------------------------------
    def slice(self, start, stop, basis=None):
        """Device-side copy of values[start:stop]."""
        ctx = get_context()
        n = stop - start
        out = ctx.alloc(n)
        check(ctx.L.plonk_mem_d2d(ctx.handle, out.ptr, self.device().at(start), 32 * n))
        return Polynomial._from_device(out, basis or self.basis)
