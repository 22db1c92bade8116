"""Constraint-language compiler: text constraints -> selector / permutation vectors, witness fill.
(oracle: test infrastructure only)

Restates /root/reference/compiler/{utils,assembly,program}.py.  These are callers of the hot
path (they determine its inputs); behaviours are pinned by the vkey goldens K3-K5 and the golden
proof K6.  One deliberate difference: `Cell.label` (compiler/utils.py:45-47)
recomputes `roots_of_unity(n)` per cell (O(n^2)); here the root table is built once per Program.
"""
from .field import R_MOD, roots_of_unity
from .fr_poly import Basis, Polynomial

LEFT, RIGHT, OUTPUT = 1, 2, 3  # compiler/utils.py:6-9 Column values


def is_valid_variable_name(name: str) -> bool:  # compiler/utils.py:59-60
    return len(name) > 0 and name.isalnum() and name[0] not in "0123456789"


def get_product_key(key1, key2):  # compiler/utils.py:54-56
    members = sorted((key1 or "").split("*") + (key2 or "").split("*"))
    return "*".join([x for x in members if x])


def evaluate(exprs, first_is_negative=False):
    """compiler/assembly.py:71-100 — tokens -> {term: coefficient}; + and - before *."""
    if "+" in exprs:
        i = exprs.index("+")
        L = evaluate(exprs[:i], first_is_negative)
        R = evaluate(exprs[i + 1 :], False)
        return {x: L.get(x, 0) + R.get(x, 0) for x in set(L) | set(R)}
    if "-" in exprs:
        i = exprs.index("-")
        L = evaluate(exprs[:i], first_is_negative)
        R = evaluate(exprs[i + 1 :], True)
        return {x: L.get(x, 0) + R.get(x, 0) for x in set(L) | set(R)}
    if "*" in exprs:
        i = exprs.index("*")
        L = evaluate(exprs[:i], first_is_negative)
        R = evaluate(exprs[i + 1 :], first_is_negative)
        o = {}
        for k1 in L:
            for k2 in R:
                o[get_product_key(k1, k2)] = L[k1] * R[k2]
        return o
    if len(exprs) > 1:
        raise Exception("No ops, expected sub-expr to be a unit: {}".format(exprs[1]))
    if exprs[0][0] == "-":
        return evaluate([exprs[0][1:]], not first_is_negative)
    if exprs[0].isnumeric():
        return {"": int(exprs[0]) * (-1 if first_is_negative else 1)}
    if is_valid_variable_name(exprs[0]):
        return {exprs[0]: -1 if first_is_negative else 1}
    raise Exception("ok wtf is {}".format(exprs[0]))


class AssemblyEqn:
    """compiler/assembly.py:29-59: wires (L, R, O variable names) + coefficient map."""

    def __init__(self, wires, coeffs):
        self.wires = tuple(wires)  # (L, R, O)
        self.coeffs = coeffs

    def gate(self):
        """Returns (L, R, M, O, C) as ints mod r.  compiler/assembly.py:37-59."""
        wl, wr, wo = self.wires
        c = self.coeffs
        gl = -c.get(wl, 0)
        gr = -c.get(wr, 0) if wr != wl else 0
        gc = -c.get("", 0)
        go = c.get("$output_coeff", 1)
        gm = -c.get(get_product_key(wl, wr), 0) if None not in self.wires else 0
        return tuple(v % R_MOD for v in (gl, gr, gm, go, gc))


def eq_to_assembly(eq: str) -> AssemblyEqn:
    """compiler/assembly.py:122-166."""
    tokens = eq.rstrip("\n").split(" ")
    if tokens[1] in ("<==", "==="):
        out = tokens[0]
        coeffs = evaluate(tokens[2:])
        if out[0] == "-":
            out = out[1:]
            coeffs["$output_coeff"] = -1
        if not is_valid_variable_name(out):
            raise Exception("Invalid out variable name: {}".format(out))
        variables = []
        for t in tokens[2:]:
            var = t.lstrip("-")
            if is_valid_variable_name(var) and var not in variables:
                variables.append(var)
        allowed = variables + ["", "$output_coeff"]
        if len(variables) == 0:
            pass
        elif len(variables) == 1:
            variables.append(variables[0])
            allowed.append(get_product_key(*variables))
        elif len(variables) == 2:
            allowed.append(get_product_key(*variables))
        else:
            raise Exception("Max 2 variables, found {}".format(variables))
        for key in coeffs:
            if key not in allowed:
                raise Exception("Disallowed multiplication: {}".format(key))
        wires = variables + [None] * (2 - len(variables)) + [out]
        return AssemblyEqn(wires, coeffs)
    if tokens[1] == "public":
        return AssemblyEqn((tokens[0], None, None), {tokens[0]: -1, "$output_coeff": 0, "$public": True})
    raise Exception("Unsupported op: {}".format(tokens[1]))


class CommonPreprocessedInput:  # compiler/program.py:10-30
    def __init__(self, group_order, QM, QL, QR, QO, QC, S1, S2, S3):
        self.group_order = group_order
        self.QM, self.QL, self.QR, self.QO, self.QC = QM, QL, QR, QO, QC
        self.S1, self.S2, self.S3 = S1, S2, S3


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

    def common_preprocessed_input(self):  # compiler/program.py:44-57
        L, R, M, O, C = self.make_gate_polynomials()
        S = self.make_s_polynomials()
        return CommonPreprocessedInput(self.group_order, M, L, R, O, C, S[LEFT], S[RIGHT], S[OUTPUT])

    def make_s_polynomials(self):
        """compiler/program.py:70-113.  Cells are (row, column) so tuple order == Cell.__lt__."""
        n = self.group_order
        uses = {None: set()}
        for row, c in enumerate(self.constraints):
            for column, value in zip((LEFT, RIGHT, OUTPUT), c.wires):
                uses.setdefault(value, set()).add((row, column))
        for row in range(len(self.constraints), n):
            for column in (LEFT, RIGHT, OUTPUT):
                uses[None].add((row, column))
        roots = roots_of_unity(n)
        S = {LEFT: [0] * n, RIGHT: [0] * n, OUTPUT: [0] * n}
        for _, cells in uses.items():
            cells = sorted(cells)
            for i, (row, column) in enumerate(cells):
                nrow, ncol = cells[(i + 1) % len(cells)]
                S[ncol][nrow] = roots[row] * column % R_MOD  # Cell.label, compiler/utils.py:45-47
        return {k: Polynomial(v, Basis.LAGRANGE) for k, v in S.items()}

    def get_public_assignments(self):  # compiler/program.py:116-130
        o = []
        no_more_allowed = False
        for coeff in self.coeffs():
            if coeff.get("$public", False) is True:
                if no_more_allowed:
                    raise Exception("Public var declarations must be at the top")
                var_name = [x for x in list(coeff.keys()) if "$" not in str(x)][0]
                if coeff != {"$public": True, "$output_coeff": 0, var_name: -1}:
                    raise Exception("Malformatted coeffs: {}".format(coeff))
                o.append(var_name)
            else:
                no_more_allowed = True
        return o

    def make_gate_polynomials(self):  # compiler/program.py:134-155
        n = self.group_order
        cols = [[0] * n for _ in range(5)]
        for i, c in enumerate(self.constraints):
            for col, v in zip(cols, c.gate()):
                col[i] = v
        return tuple(Polynomial(col, Basis.LAGRANGE) for col in cols)  # (L, R, M, O, C)

    def fill_variable_assignments(self, starting_assignments):  # compiler/program.py:161-192
        out = {k: v % R_MOD for k, v in starting_assignments.items()}
        out[None] = 0
        for c in self.constraints:
            in_L, in_R, output = c.wires
            coeffs = c.coeffs
            out_coeff = coeffs.get("$output_coeff", 1)
            product_key = get_product_key(in_L, in_R)
            if output is not None and out_coeff in (-1, 1):
                new_value = (
                    coeffs.get("", 0)
                    + out[in_L] * coeffs.get(in_L, 0)
                    + out[in_R] * coeffs.get(in_R, 0) * (1 if in_R != in_L else 0)
                    + out[in_L] * out[in_R] * coeffs.get(product_key, 0)
                ) * out_coeff % R_MOD
                if output in out:
                    if out[output] != new_value:
                        raise Exception("Failed assertion: {} = {}".format(out[output], new_value))
                else:
                    out[output] = new_value
        return out
