"""Subset of py_ecc 6.0.0 `bn128` (affine BN254 G1; minimal FQ2 for the G2 constants)."""
from py_ecc.fields.field_elements import FQ as _FQ

curve_order = 21888242871839275222246405745257275088548364400416034343698204186575808495617
field_modulus = 21888242871839275222246405745257275088696311157297823662689037894645226208583


class FQ(_FQ):
    field_modulus = field_modulus


class FQ2:
    """a + b*i with i^2 = -1."""

    def __init__(self, coeffs):
        self.coeffs = tuple(FQ(c) for c in coeffs)

    def __eq__(self, other):
        return isinstance(other, FQ2) and self.coeffs == other.coeffs

    def __add__(self, o):
        return FQ2([self.coeffs[0] + o.coeffs[0], self.coeffs[1] + o.coeffs[1]])

    def __sub__(self, o):
        return FQ2([self.coeffs[0] - o.coeffs[0], self.coeffs[1] - o.coeffs[1]])

    def __mul__(self, o):
        a, b = self.coeffs
        c, d = o.coeffs
        return FQ2([a * c - b * d, a * d + b * c])

    def inv(self):
        a, b = self.coeffs
        norm = a * a + b * b
        return FQ2([a / norm, (FQ(0) - b) / norm])


b = FQ(3)
b2 = FQ2([3, 0]) * FQ2([9, 1]).inv()
G1 = (FQ(1), FQ(2))
G2 = (
    FQ2([
        10857046999023057135944570762232829481370756359578518086990519993285655852781,
        11559732032986387107991004021392285783925812861821192530917403151452391805634,
    ]),
    FQ2([
        8495653923123431417604973247489272438418190587263600148770280649306958101930,
        4082367875863433681332203403145435568316851327593401208105741076214120093531,
    ]),
)
Z1 = None
Z2 = None


def is_on_curve(pt, b_):
    if pt is None:
        return True
    x, y = pt
    return y * y - x * x * x == b_


def double(pt):
    if pt is None:
        return None
    x, y = pt
    m = 3 * x * x / (2 * y)
    newx = m * m - 2 * x
    newy = -m * newx + m * x - y
    return (newx, newy)


def add(p1, p2):
    if p1 is None or p2 is None:
        return p1 if p2 is None else p2
    x1, y1 = p1
    x2, y2 = p2
    if x2 == x1 and y2 == y1:
        return double(p1)
    elif x2 == x1:
        return None
    m = (y2 - y1) / (x2 - x1)
    newx = m * m - x1 - x2
    newy = -m * newx + m * x1 - y1
    return (newx, newy)


def multiply(pt, n):
    if n == 0:
        return None
    elif n == 1:
        return pt
    elif not n % 2:
        return multiply(double(pt), n // 2)
    else:
        return add(multiply(double(pt), int(n // 2)), pt)


def neg(pt):
    if pt is None:
        return None
    x, y = pt
    return (x, -y)


def pairing(*_a, **_k):
    raise NotImplementedError("pairing is off the prover hot path; not provided by the shim")
