"""Host-side field elements: `Scalar` (BN254 Fr) and `Fq` (BN254 base field).

Mirrors the interface the reference gets from py_ecc's `FQ` through `curve.Scalar`
(/root/reference/curve.py:10-27): `Scalar(int)`, `.n`, + - * / ** with ints or Scalars on either
side, `x / 0 == 0`, `root_of_unity`, `roots_of_unity`.  Only a handful of these live on the host per
proof (challenges, evaluations); vectors live on the GPU inside `Polynomial`.
"""

R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
Q_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583
primitive_root = 5  # curve.py:5


class _PrimeFieldElement:
    field_modulus = None
    __slots__ = ("n",)

    def __init__(self, val):
        if isinstance(val, _PrimeFieldElement):
            self.n = val.n
        elif isinstance(val, int):
            self.n = val % self.field_modulus
        else:
            raise TypeError("Expected an int or field element, but got {}".format(type(val)))

    @staticmethod
    def _raw(other):
        if isinstance(other, _PrimeFieldElement):
            return other.n
        if isinstance(other, int):
            return other
        raise TypeError("Expected an int or field element, but got {}".format(type(other)))

    @classmethod
    def _inv(cls, a):
        a %= cls.field_modulus
        return 0 if a == 0 else pow(a, -1, cls.field_modulus)

    def __add__(self, other):
        return type(self)(self.n + self._raw(other))

    __radd__ = __add__

    def __sub__(self, other):
        return type(self)(self.n - self._raw(other))

    def __rsub__(self, other):
        return type(self)(self._raw(other) - self.n)

    def __mul__(self, other):
        return type(self)(self.n * self._raw(other))

    __rmul__ = __mul__

    def __truediv__(self, other):
        return type(self)(self.n * self._inv(self._raw(other)))

    def __rtruediv__(self, other):
        return type(self)(self._raw(other) * self._inv(self.n))

    def __pow__(self, e):
        return type(self)(pow(self.n, e, self.field_modulus))

    def __neg__(self):
        return type(self)(-self.n)

    def __eq__(self, other):
        if isinstance(other, _PrimeFieldElement):
            return self.n == other.n
        if isinstance(other, int):
            return self.n == other
        return NotImplemented

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    def __hash__(self):
        return hash(self.n)

    def __int__(self):
        return self.n

    def __repr__(self):
        return repr(self.n)

    @classmethod
    def zero(cls):
        return cls(0)

    @classmethod
    def one(cls):
        return cls(1)


class Scalar(_PrimeFieldElement):
    field_modulus = R_MOD
    __slots__ = ()

    @classmethod
    def root_of_unity(cls, group_order: int):  # curve.py:14-16
        return cls(pow(primitive_root, (R_MOD - 1) // group_order, R_MOD))

    @classmethod
    def roots_of_unity(cls, group_order: int):  # curve.py:19-24
        w = cls.root_of_unity(group_order).n
        o, cur = [], 1
        for _ in range(max(group_order, 2)):
            o.append(cls(cur))
            cur = cur * w % R_MOD
        return o


class Fq(_PrimeFieldElement):
    field_modulus = Q_MOD
    __slots__ = ()


def le32(x: int) -> bytes:
    return int(x).to_bytes(32, "little")
