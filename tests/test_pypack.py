"""The CPython packing helper of BatchProver.upload (plonkathon_amd/_pypack.so: ints / witness dictionaries -> 32-byte
little-endian words reduced mod r) against int.to_bytes, including the values its digit-copy fast path must hand to the
general path (negative, >= modulus, >= 2^256, non-int objects)."""
import random

import pytest

from oracle.field import R_MOD

_pypack = pytest.importorskip("plonkathon_amd._pypack")


class Wrapped:
    def __init__(self, v):
        self.v = v

    def __int__(self):
        return self.v


def want(v):
    return (int(v) % R_MOD).to_bytes(32, "little")


def test_pack_le32_edges_and_random_values():
    edge = [0, 1, R_MOD - 1, R_MOD, R_MOD + 1, 2**256 - 1, 2**256, 2**270 - 1, 2**270, 2**300, -1, -R_MOD, -R_MOD - 5,
            2**30 - 1, 2**30, 2**60 - 1, 2**60, 2**64 - 1, 2**64, 2**90, 2**120, 2**180, 2**240 - 1, 2**255, R_MOD >> 1, True,
            Wrapped(7), Wrapped(R_MOD + 3)]
    got = _pypack.pack_le32(edge, R_MOD)
    assert [got[32 * i:32 * i + 32] for i in range(len(edge))] == [want(v) for v in edge]
    rng = random.Random(3)
    vals = [rng.getrandbits(rng.randrange(1, 300)) * (1 if rng.random() < 0.9 else -1) for _ in range(5000)]
    got = _pypack.pack_le32(vals, R_MOD)
    assert got == b"".join(want(v) for v in vals)


def test_pack_dicts_le32_orders_by_key_and_reports_missing_keys():
    rng = random.Random(4)
    keys = ["a", "b", None, 3, "c"]
    wits = [{k: rng.randrange(-R_MOD, 2 * R_MOD) for k in keys} for _ in range(7)]
    got = _pypack.pack_dicts_le32(wits, keys, R_MOD)
    assert got == b"".join(want(w[k]) for w in wits for k in keys)
    with pytest.raises(KeyError):
        _pypack.pack_dicts_le32([{"a": 1}], ["a", "zz"], R_MOD)
    assert _pypack.pack_dicts_le32([], keys, R_MOD) == b""


def test_pack_dicts_le32_walk_shortcut_never_changes_the_result():
    """Dictionaries that share keys and order take the walking shortcut; every deviation must fall back to look-ups."""
    rng = random.Random(8)
    keys = ["v%d" % i for i in range(40)]

    def ref(ws):
        return b"".join(want(w[k]) for w in ws for k in keys)

    same = [{k: rng.randrange(R_MOD) for k in keys} for _ in range(5)]
    assert _pypack.pack_dicts_le32(same, keys, R_MOD) == ref(same)
    # the dictionaries list the keys in another order than `keys`, with keys the circuit does not use in between
    order = keys[::-1]
    extra = [dict([("unused", 1)] + [(k, rng.randrange(R_MOD)) for k in order] + [("tail", 2)]) for _ in range(4)]
    assert _pypack.pack_dicts_le32(extra, keys, R_MOD) == ref(extra)
    # deviations inside the batch: other order, other size, equal-but-distinct key objects, values off the fast path
    odd = [dict(same[0]),
           {k: rng.randrange(R_MOD) for k in order},                       # same keys, other order
           dict({k: rng.randrange(R_MOD) for k in keys}, more=3),           # one key more
           {"".join(list(k)): rng.randrange(R_MOD) for k in keys},          # equal strings, new objects
           {k: (-5 if i == 7 else R_MOD + 9 if i == 8 else Wrapped(11) if i == 9 else rng.randrange(R_MOD)) for i, k in enumerate(keys)},
           {k: 2**300 + i for i, k in enumerate(keys)}]
    assert _pypack.pack_dicts_le32(odd, keys, R_MOD) == ref(odd)
    # a later dictionary that lacks a key raises KeyError(key) whatever the first one looked like
    broken = [dict(same[0]), {k: 1 for k in keys[:-1]}]
    with pytest.raises(KeyError):
        _pypack.pack_dicts_le32(broken, keys, R_MOD)
    with pytest.raises(KeyError):
        _pypack.pack_dicts_le32([{k: 1 for k in keys[1:]}, dict(same[0])], keys, R_MOD)
    # a key listed twice is served by the general path
    assert _pypack.pack_dicts_le32(same, keys + keys[:3], R_MOD) == b"".join(want(w[k]) for w in same for k in keys + keys[:3])


def test_generic_lincomb_and_multisubset_on_integers():
    """The product's `lincomb` / `multisubset` with the reference's generic signature (curve.py:59-111) on the group the reference's
    own self-test uses — plain integers (its K8 vector, test.py via tools/gen_golden.py) — and on a custom adder.  Host only: G1
    points go to the GPU (tests/test_gpu_parity.py::test_lincomb_golden)."""
    import json
    import os

    from plonkathon_amd import lincomb, multisubset

    k8 = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "lincomb_vectors.json")))["k8_int"]
    numbers, factors = [int(x) for x in k8["numbers"]], [int(x) for x in k8["factors"]]
    assert [str(x) for x in multisubset(numbers, [set(s) for s in k8["subsets"]])] == k8["multisubset"]
    assert str(lincomb(numbers, factors)) == k8["lincomb"]
    assert lincomb(numbers, factors) == sum(n * f for n, f in zip(numbers, factors))  # curve.py:139
    m = 2**61 - 1
    assert lincomb([3, 5, 7], [10, 0, 2**70 + 1], adder=lambda x, y: (x + y) % m, zero=0) == (30 + 7 * (2**70 + 1)) % m
    assert multisubset(["a", "b", "c"], [[0, 2], [], [1]], adder=lambda x, y: x + y, zero="") == ["ac", "", "b"]
    try:
        lincomb([], [])
    except ValueError:
        pass
    else:
        raise AssertionError("an empty linear combination must raise, as curve.py:93 does")


def test_column_and_cell_helpers():
    """compiler/utils.py:6-51 in the product namespace: ordering, labels w^row * column, and `Column` keys that also answer to 1, 2, 3."""
    from plonkathon_amd import Cell, Column, Program
    from plonkathon_amd.field import R_MOD, Scalar

    assert Column.LEFT < Column.RIGHT < Column.OUTPUT and Column.variants() == [Column.LEFT, Column.RIGHT, Column.OUTPUT]
    c = Cell(Column.RIGHT, 3)
    assert repr(c) == "(3, 2)" and c < Cell(Column.LEFT, 4) and c == Cell(2, 3) and len({c, Cell(2, 3)}) == 1
    assert sorted([Cell(3, 1), Cell(1, 2), Cell(2, 1)]) == [Cell(2, 1), Cell(3, 1), Cell(1, 2)]
    w = Scalar.root_of_unity(8).n
    assert c.label(8).n == pow(w, 3, R_MOD) * 2 % R_MOD
    try:
        Cell(Column.LEFT, 8).label(8)
    except AssertionError:
        pass
    else:
        raise AssertionError("row >= group_order must be refused")
    assert set(Program(["c <== a * b"], 8).permutation_columns()) == {1, 2, 3} == {int(k) for k in Column.variants()}

