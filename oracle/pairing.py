"""BN254 optimal-ate pairing on plain Python ints.  (oracle: test infrastructure only)

py_ecc 6.0.0's `bn128_pairing` / `bn128_curve` (third-party, not under /root/reference) restated from its
published algorithm: Fq12 = Fq[w]/(w^12 - 18 w^6 + 82), G2 on the twist y^2 = x^3 + 3/(9+i), Miller loop
over ate_loop_count = 29793968203157093288 with the two Frobenius correction steps, final exponentiation
(p^12 - 1)/r.  Used only by oracle/verifier.py to check that proofs VERIFY (the reference's
TESTING_verifier_DO_NOT_OPEN.py:148-160 pairing check); it is off the prover hot path.  Nothing in the
reference pins the pairing numerically, so the tests check it through bilinearity and through the golden
proof test/proof.pickle, which must verify.
"""
from .field import Q_MOD as P, R_MOD as N, inv

FQ12_MOD = [82, 0, 0, 0, 0, 0, -18, 0, 0, 0, 0, 0]
FQ2_MOD = [1, 0]


class FQP:
    """Element of Fq[x]/(x^deg + sum mod_coeffs[i] x^i)."""

    degree = 0
    mod_coeffs = ()

    def __init__(self, coeffs):
        assert len(coeffs) == self.degree
        self.c = [int(x) % P for x in coeffs]

    def __add__(self, o):
        return type(self)([a + b for a, b in zip(self.c, o.c)])

    def __sub__(self, o):
        return type(self)([a - b for a, b in zip(self.c, o.c)])

    def __neg__(self):
        return type(self)([-a for a in self.c])

    def __eq__(self, o):
        return type(self) is type(o) and self.c == o.c

    def __mul__(self, o):
        if isinstance(o, int):
            return type(self)([a * o for a in self.c])
        d = self.degree
        b = [0] * (2 * d - 1)
        for i, x in enumerate(self.c):
            if x:
                for j, y in enumerate(o.c):
                    b[i + j] += x * y
        for exp in range(d - 2, -1, -1):
            top = b.pop()
            if top:
                for i, m in enumerate(self.mod_coeffs):
                    if m:
                        b[exp + i] -= top * m
        return type(self)(b)

    __rmul__ = __mul__

    def inv(self):
        """Extended Euclid on polynomials over Fq (the algorithm py_ecc's FQP.inv uses)."""
        d = self.degree

        def deg(p):
            k = len(p) - 1
            while k and p[k] == 0:
                k -= 1
            return k

        def poly_div(a, b):  # quotient of a / b
            dega, degb = deg(a), deg(b)
            temp, o = list(a), [0] * len(a)
            ib = inv(b[degb], P)
            for i in range(dega - degb, -1, -1):
                o[i] = temp[degb + i] * ib % P
                for c in range(degb + 1):
                    temp[c + i] = (temp[c + i] - o[i] * b[c]) % P
            return o[: deg(o) + 1]

        lm, hm = [1] + [0] * d, [0] * (d + 1)
        low, high = [x % P for x in self.c] + [0], [x % P for x in self.mod_coeffs] + [1]
        while deg(low):
            r = poly_div(high, low)
            r += [0] * (d + 1 - len(r))
            nm, new = list(hm), list(high)
            for i in range(d + 1):
                for j in range(d + 1 - i):
                    nm[i + j] = (nm[i + j] - lm[i] * r[j]) % P
                    new[i + j] = (new[i + j] - low[i] * r[j]) % P
            lm, low, hm, high = nm, new, lm, low
        il = inv(low[0], P)
        return type(self)([x * il for x in lm[:d]])

    def __truediv__(self, o):
        if isinstance(o, int):
            return self * inv(o, P)
        return self * o.inv()

    def __pow__(self, e):
        result = type(self).one()
        base = self
        while e:
            if e & 1:
                result = result * base
            base = base * base
            e >>= 1
        return result

    @classmethod
    def one(cls):
        return cls([1] + [0] * (cls.degree - 1))

    @classmethod
    def zero(cls):
        return cls([0] * cls.degree)


class FQ2(FQP):
    degree = 2
    mod_coeffs = FQ2_MOD


class FQ12(FQP):
    degree = 12
    mod_coeffs = FQ12_MOD


# ---- curve over extension fields (affine, None = identity) --------------------------------------------
B2 = FQ2([3, 0]) / FQ2([9, 1])
B12 = FQ12([3] + [0] * 11)
G2 = (
    FQ2([10857046999023057135944570762232829481370756359578518086990519993285655852781,
         11559732032986387107991004021392285783925812861821192530917403151452391805634]),
    FQ2([8495653923123431417604973247489272438418190587263600148770280649306958101930,
         4082367875863433681332203403145435568316851327593401208105741076214120093531]),
)


def is_on_curve(pt, b):
    if pt is None:
        return True
    x, y = pt
    return y * y - x * x * x == b


def double(pt):
    if pt is None:
        return None
    x, y = pt
    m = (x * x * 3) / (y * 2)
    nx = m * m - x * 2
    ny = -(m * nx) + m * x - y
    return (nx, ny)


def add(p1, p2):
    if p1 is None or p2 is None:
        return p1 if p2 is None else p2
    x1, y1 = p1
    x2, y2 = p2
    if x2 == x1 and y2 == y1:
        return double(p1)
    if x2 == x1:
        return None
    m = (y2 - y1) / (x2 - x1)
    nx = m * m - x1 - x2
    ny = -(m * nx) + m * x1 - y1
    return (nx, ny)


def multiply(pt, n):
    if n == 0 or pt is None:
        return None
    result, addend = None, pt
    while n:
        if n & 1:
            result = add(result, addend)
        addend = double(addend)
        n >>= 1
    return result


def neg(pt):
    return None if pt is None else (pt[0], -pt[1])


W = FQ12([0, 1] + [0] * 10)


def twist(pt):
    """G2 point over Fq2 -> the isomorphic curve over Fq12."""
    if pt is None:
        return None
    x, y = pt
    xc = [x.c[0] - x.c[1] * 9, x.c[1]]
    yc = [y.c[0] - y.c[1] * 9, y.c[1]]
    nx = FQ12([xc[0]] + [0] * 5 + [xc[1]] + [0] * 5)
    ny = FQ12([yc[0]] + [0] * 5 + [yc[1]] + [0] * 5)
    return (nx * (W ** 2), ny * (W ** 3))


def cast_to_fq12(pt):
    if pt is None:
        return None
    x, y = pt
    return (FQ12([x] + [0] * 11), FQ12([y] + [0] * 11))


ATE_LOOP_COUNT = 29793968203157093288
LOG_ATE_LOOP_COUNT = 63


def linefunc(P1, P2, T):
    x1, y1 = P1
    x2, y2 = P2
    xt, yt = T
    if x1 != x2:
        m = (y2 - y1) / (x2 - x1)
        return m * (xt - x1) - (yt - y1)
    if y1 == y2:
        m = (x1 * x1 * 3) / (y1 * 2)
        return m * (xt - x1) - (yt - y1)
    return xt - x1


def miller_loop(Q, Pt):
    if Q is None or Pt is None:
        return FQ12.one()
    R, f = Q, FQ12.one()
    for i in range(LOG_ATE_LOOP_COUNT, -1, -1):
        f = f * f * linefunc(R, R, Pt)
        R = double(R)
        if ATE_LOOP_COUNT & (2 ** i):
            f = f * linefunc(R, Q, Pt)
            R = add(R, Q)
    Q1 = (Q[0] ** P, Q[1] ** P)
    nQ2 = (Q1[0] ** P, -(Q1[1] ** P))
    f = f * linefunc(R, Q1, Pt)
    R = add(R, Q1)
    f = f * linefunc(R, nQ2, Pt)
    return f ** ((P ** 12 - 1) // N)


def pairing(Q, Pt):
    """e(Pt, Q) with Q in G2 (Fq2 coordinates) and Pt in G1 (int tuple or None)."""
    assert is_on_curve(Q, B2)
    return miller_loop(twist(Q), cast_to_fq12(Pt))
