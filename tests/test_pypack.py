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
