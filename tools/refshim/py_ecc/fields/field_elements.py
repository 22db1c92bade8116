"""Subset of py_ecc 6.0.0 `FQ` semantics used by the reference."""


class FQ:
    field_modulus = None

    def __init__(self, val):
        if isinstance(val, FQ):
            self.n = val.n
        elif isinstance(val, int):
            self.n = val % self.field_modulus
        else:
            raise TypeError("Expected an int or FQ object, but got {}".format(type(val)))

    @staticmethod
    def _n(other):
        if isinstance(other, FQ):
            return other.n
        if isinstance(other, int):
            return other
        raise TypeError("Expected an int or FQ object, but got {}".format(type(other)))

    def __add__(self, other):
        return type(self)((self.n + self._n(other)) % self.field_modulus)

    __radd__ = __add__

    def __mul__(self, other):
        return type(self)((self.n * self._n(other)) % self.field_modulus)

    __rmul__ = __mul__

    def __sub__(self, other):
        return type(self)((self.n - self._n(other)) % self.field_modulus)

    def __rsub__(self, other):
        return type(self)((self._n(other) - self.n) % self.field_modulus)

    @classmethod
    def _inv(cls, a):
        a %= cls.field_modulus
        return 0 if a == 0 else pow(a, -1, cls.field_modulus)  # inverse of 0 is 0

    def __truediv__(self, other):
        return type(self)(self.n * self._inv(self._n(other)) % self.field_modulus)

    def __rtruediv__(self, other):
        return type(self)(self._inv(self.n) * self._n(other) % self.field_modulus)

    def __pow__(self, other):
        return type(self)(pow(self.n, other, self.field_modulus))

    def __eq__(self, other):
        if isinstance(other, FQ):
            return self.n == other.n
        if isinstance(other, int):
            return self.n == other
        return NotImplemented

    def __ne__(self, other):
        return not self == other

    def __hash__(self):
        return hash(self.n)

    def __neg__(self):
        return type(self)(-self.n)

    def __repr__(self):
        return repr(self.n)

    def __int__(self):
        return self.n

    @classmethod
    def one(cls):
        return cls(1)

    @classmethod
    def zero(cls):
        return cls(0)
