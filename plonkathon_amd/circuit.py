"""Circuit front end: constraint text -> gate selectors, copy-constraint permutation, witness fill.

Host-side counterpart of /root/reference/compiler/{assembly,program,utils}.py — the code that
produces the hot path's inputs (QL..QC, S1..S3, wire assignments).  Same `Program` /
`CommonPreprocessedInput` surface and the same observable behaviour (pinned by
the zkrepl verification-key goldens), written as a flat term/factor parser and an O(n) permutation
builder (the reference recomputes roots_of_unity(n) per cell: 15 s at n = 2^11).

Grammar notes that matter for parity (compiler/assembly.py:71-100): tokens are separated by single
spaces; `+`/`-` tokens split terms, `*` tokens split factors; a `-` sign applies to EVERY factor of
the term it precedes (so `- 45 * x` contributes +45*x, `- 45 * x * y` contributes -45*x*y); a leading
`-` glued to a token flips that factor only.
"""
import enum
import functools
from dataclasses import dataclass
from typing import Optional

from .field import R_MOD, Scalar
from .polynomial import Basis, Polynomial

OUTPUT_COEFF, PUBLIC_FLAG = "$output_coeff", "$public"


class Column(enum.IntEnum):
    """compiler/utils.py:6-19: the three wire columns.  An IntEnum, so `S[Column.LEFT]` and `S[1]` name the same entry of
    `Program.make_s_polynomials()` and the members order as the reference's `__lt__` does."""

    LEFT = 1
    RIGHT = 2
    OUTPUT = 3

    @staticmethod
    def variants():
        return [Column.LEFT, Column.RIGHT, Column.OUTPUT]


@functools.total_ordering
class Cell:
    """compiler/utils.py:22-51: one cell of the wire table; ordered by (row, column), labelled w^row * column — the value the
    permutation argument stores for it (`Program.permutation_columns` computes the same labels for all cells at once)."""

    __slots__ = ("column", "row")

    def __init__(self, column, row: int):
        self.column, self.row = Column(column), int(row)

    def _key(self):
        return (self.row, int(self.column))

    def __eq__(self, other):
        return isinstance(other, Cell) and self._key() == other._key()

    def __lt__(self, other):
        return self._key() < other._key() if isinstance(other, Cell) else NotImplemented

    def __hash__(self):
        return hash(self._key())

    def __repr__(self):
        return "(%d, %d)" % self._key()

    def label(self, group_order: int) -> Scalar:
        assert self.row < group_order
        return Scalar(pow(Scalar.root_of_unity(group_order).n, self.row, R_MOD) * int(self.column) % R_MOD)


def is_valid_variable_name(name: str) -> bool:  # compiler/utils.py:59-60
    return len(name) > 0 and name.isalnum() and name[0] not in "0123456789"


def get_product_key(key1, key2):  # compiler/utils.py:54-56
    parts = [p for p in (key1 or "").split("*") + (key2 or "").split("*") if p]
    return "*".join(sorted(parts))


def _parse_factor(token: str, negative: bool):
    while token and token[0] == "-":
        token, negative = token[1:], not negative
    if not token:
        raise Exception("empty operand in expression")
    sign = -1 if negative else 1
    if token.isnumeric():
        return "", int(token) * sign
    if is_valid_variable_name(token):
        return token, sign
    raise Exception("ok wtf is {}".format(token))


def parse_expression(tokens):
    """tokens (already split on spaces) -> {monomial key: integer coefficient}."""
    totals = {}
    term, negative = [], False

    def flush(term_tokens, neg):
        if not term_tokens:
            raise Exception("empty term in expression")
        key, coeff, expect_operand = "", 1, True
        for tok in term_tokens:
            if tok == "*":
                if expect_operand:
                    raise Exception("two operators in a row")
                expect_operand = True
                continue
            if not expect_operand:
                raise Exception("No ops, expected sub-expr to be a unit: {}".format(tok))
            k, c = _parse_factor(tok, neg)
            key, coeff = get_product_key(key, k), coeff * c
            expect_operand = False
        if expect_operand:
            raise Exception("expression ends with an operator")
        totals[key] = totals.get(key, 0) + coeff

    for tok in tokens:
        if tok in ("+", "-"):
            flush(term, negative)
            term, negative = [], tok == "-"
        else:
            term.append(tok)
    flush(term, negative)
    return totals


@dataclass
class GateWires:  # compiler/assembly.py:8-16
    L: Optional[str]
    R: Optional[str]
    O: Optional[str]

    def as_list(self):
        return [self.L, self.R, self.O]


@dataclass
class Gate:  # compiler/assembly.py:19-27
    L: Scalar
    R: Scalar
    M: Scalar
    O: Scalar
    C: Scalar


@dataclass
class AssemblyEqn:  # compiler/assembly.py:30-59
    wires: GateWires
    coeffs: dict

    def L(self):
        return Scalar(-self.coeffs.get(self.wires.L, 0))

    def R(self):
        if self.wires.R != self.wires.L:
            return Scalar(-self.coeffs.get(self.wires.R, 0))
        return Scalar(0)

    def C(self):
        return Scalar(-self.coeffs.get("", 0))

    def O(self):
        return Scalar(self.coeffs.get(OUTPUT_COEFF, 1))

    def M(self):
        if None in self.wires.as_list():
            return Scalar(0)
        return Scalar(-self.coeffs.get(get_product_key(self.wires.L, self.wires.R), 0))

    def gate(self):
        return Gate(self.L(), self.R(), self.M(), self.O(), self.C())


def eq_to_assembly(eq: str) -> AssemblyEqn:  # compiler/assembly.py:122-166
    tokens = eq.rstrip("\n").split(" ")
    op = tokens[1]
    if op == "public":
        return AssemblyEqn(GateWires(tokens[0], None, None), {tokens[0]: -1, OUTPUT_COEFF: 0, PUBLIC_FLAG: True})
    if op not in ("<==", "==="):
        raise Exception("Unsupported op: {}".format(op))
    out, rhs = tokens[0], tokens[2:]
    coeffs = parse_expression(rhs)
    if out[0] == "-":
        out = out[1:]
        coeffs[OUTPUT_COEFF] = -1
    if not is_valid_variable_name(out):
        raise Exception("Invalid out variable name: {}".format(out))
    seen = []
    for tok in rhs:
        name = tok.lstrip("-")
        if is_valid_variable_name(name) and name not in seen:
            seen.append(name)
    if len(seen) > 2:
        raise Exception("Max 2 variables, found {}".format(seen))
    allowed = set(seen) | {"", OUTPUT_COEFF}
    if len(seen) == 1:
        seen.append(seen[0])
    if len(seen) == 2:
        allowed.add(get_product_key(seen[0], seen[1]))
    for key in coeffs:
        if key not in allowed:
            raise Exception("Disallowed multiplication: {}".format(key))
    wires = seen + [None] * (2 - len(seen)) + [out]
    return AssemblyEqn(GateWires(*wires), coeffs)


@dataclass
class CommonPreprocessedInput:  # compiler/program.py:10-30
    group_order: int
    QM: Polynomial
    QL: Polynomial
    QR: Polynomial
    QO: Polynomial
    QC: Polynomial
    S1: Polynomial
    S2: Polynomial
    S3: Polynomial


class Program:
    def __init__(self, constraints, group_order: int):  # compiler/program.py:37-42
        if len(constraints) > group_order:
            raise Exception("Group order too small")
        self.constraints = [eq_to_assembly(c) for c in constraints]
        self.group_order = group_order

    @classmethod
    def from_str(cls, constraints: str, group_order: int):  # compiler/program.py:59-62
        return cls([line.strip() for line in constraints.split("\n")], group_order)

    def coeffs(self):
        return [c.coeffs for c in self.constraints]

    def wires(self):
        return [c.wires for c in self.constraints]

    def wiring_table(self):
        """(variables, cell_index): the circuit's variables in a fixed order and, per wire column L / R / O and row, the
        index of the variable that cell carries — len(variables) for an empty cell or a padding row (witness[None] = 0,
        prover.py:94-95).  What round 1 needs to build A, B, C from one encoding of each variable's value
        (prover.py:97-103) without a Python loop over the rows.  Cached."""
        if getattr(self, "_wiring", None) is None:
            import numpy as np

            rows = [w.as_list() for w in self.wires()]
            variables = sorted({v for row in rows for v in row if v is not None}, key=str)
            pos = {v: i for i, v in enumerate(variables)}
            cell_index = np.full((3, self.group_order), len(variables), dtype=np.int64)
            for i, row in enumerate(rows):
                for j, v in enumerate(row):
                    if v is not None:
                        cell_index[j, i] = pos[v]
            self._wiring = (tuple(variables), cell_index)
        return self._wiring

    def common_preprocessed_input(self) -> CommonPreprocessedInput:  # compiler/program.py:44-57
        L, R, M, O, C = self.make_gate_polynomials()
        S = self.make_s_polynomials()
        return CommonPreprocessedInput(self.group_order, M, L, R, O, C, S[1], S[2], S[3])

    def permutation_columns(self):
        """The three sigma columns as int lists: variable uses sorted by (row, column), each cell's
        label w^row * column stored at the NEXT use (compiler/program.py:70-113)."""
        n = self.group_order
        w = Scalar.root_of_unity(n).n
        roots, cur = [], 1
        for _ in range(n):
            roots.append(cur)
            cur = cur * w % R_MOD
        uses = {}
        for row, c in enumerate(self.constraints):
            for col, var in enumerate(c.wires.as_list(), start=1):
                uses.setdefault(var, []).append((row, col))
        unused = uses.setdefault(None, [])
        for row in range(len(self.constraints), n):
            unused.extend(((row, 1), (row, 2), (row, 3)))
        sigma = {1: [0] * n, 2: [0] * n, 3: [0] * n}
        for cells in uses.values():
            cells = sorted(set(cells))
            for i, (row, col) in enumerate(cells):
                nrow, ncol = cells[(i + 1) % len(cells)]
                sigma[ncol][nrow] = roots[row] * col % R_MOD
        return sigma

    def make_s_polynomials(self):
        sigma = self.permutation_columns()
        return {Column(k): Polynomial.from_ints(v, Basis.LAGRANGE) for k, v in sigma.items()}

    def gate_columns(self):
        """(L, R, M, O, C) selector columns as int lists (compiler/program.py:134-155)."""
        n = self.group_order
        cols = [[0] * n for _ in range(5)]
        for i, c in enumerate(self.constraints):
            g = c.gate()
            for col, v in zip(cols, (g.L, g.R, g.M, g.O, g.C)):
                col[i] = v.n
        return cols

    def make_gate_polynomials(self):
        return tuple(Polynomial.from_ints(col, Basis.LAGRANGE) for col in self.gate_columns())

    def get_public_assignments(self):  # compiler/program.py:116-130
        out, closed = [], False
        for coeff in self.coeffs():
            if coeff.get(PUBLIC_FLAG, False) is True:
                if closed:
                    raise Exception("Public var declarations must be at the top")
                name = [k for k in coeff if "$" not in str(k)][0]
                if coeff != {PUBLIC_FLAG: True, OUTPUT_COEFF: 0, name: -1}:
                    raise Exception("Malformatted coeffs: {}".format(coeff))
                out.append(name)
            else:
                closed = True
        return out

    def fill_variable_assignments(self, starting_assignments):  # compiler/program.py:161-192
        out = {k: int(v) % R_MOD for k, v in starting_assignments.items()}
        out[None] = 0
        for c in self.constraints:
            wl, wr, wo = c.wires.as_list()
            k = c.coeffs
            oc = k.get(OUTPUT_COEFF, 1)
            if wo is None or oc not in (-1, 1):
                continue
            value = (
                k.get("", 0)
                + out[wl] * k.get(wl, 0)
                + (out[wr] * k.get(wr, 0) if wr != wl else 0)
                + out[wl] * out[wr] * k.get(get_product_key(wl, wr), 0)
            ) * oc % R_MOD
            if wo in out:
                if out[wo] != value:
                    raise Exception("Failed assertion: {} = {}".format(out[wo], value))
            else:
                out[wo] = value
        return out
