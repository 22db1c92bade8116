"""BN254 scalar field Fr and base field Fq on plain Python ints.  (oracle: test infrastructure only)

Follows /root/reference/curve.py:1-27 (`Scalar`, `root_of_unity`, `roots_of_unity`) and the
py_ecc 6.0.0 `FQ` semantics the path relies on: a / b = a * b^-1 with the
inverse of 0 DEFINED AS 0 (no exception), x ** 0 == 1.
"""

# curve.py:10-11  Scalar.field_modulus = b.curve_order
R_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617
# py_ecc.bn128.field_modulus
Q_MOD = 21888242871839275222246405745257275088696311157297823662689037894645226208583
# curve.py:5 / utils.py:7
PRIMITIVE_ROOT = 5
TWO_ADICITY = 28


def inv(a: int, m: int = R_MOD) -> int:
    """py_ecc FQ division semantics: inverse of 0 is 0."""
    a %= m
    if a == 0:
        return 0
    return pow(a, -1, m)


def div(a: int, b: int, m: int = R_MOD) -> int:
    return a * inv(b, m) % m


def root_of_unity(group_order: int) -> int:
    """curve.py:14-16: Scalar(5) ** ((r - 1) // group_order)."""
    return pow(PRIMITIVE_ROOT, (R_MOD - 1) // group_order, R_MOD)


def roots_of_unity(group_order: int) -> list:
    """curve.py:19-24: [1, w, w^2, ...] by repeated multiplication."""
    o = [1, root_of_unity(group_order)]
    while len(o) < group_order:
        o.append(o[-1] * o[1] % R_MOD)
    return o[:group_order] if group_order >= 1 else []
