"""Trusted-setup loader and KZG commit.  (oracle: test infrastructure only)

Follows /root/reference/setup.py:23-63 (`Setup.from_file`, snarkjs .ptau layout) and the contract
of the blanked `Setup.commit` (setup.py:66-72, pinned by test.py:18-28): Lagrange values ->
ifft -> ec_lincomb with powers_of_x.
"""
from .field import Q_MOD, inv
from .fr_poly import Basis, Polynomial
from .g1 import G1, ec_lincomb

SETUP_FILE_G1_STARTPOS = 80  # setup.py:11
SETUP_FILE_POWERS_POS = 60  # setup.py:12
# py_ecc.bn128.G2 x-coordinate, c0 coefficient (published constant)
G2_X_C0 = 10857046999023057135944570762232829481370756359578518086990519993285655852781


class Setup:
    def __init__(self, powers_of_x, X2):
        self.powers_of_x = powers_of_x  # list of affine (x, y)
        self.X2 = X2  # ((x_c0, x_c1), (y_c0, y_c1)) ints

    @classmethod
    def from_bytes(cls, contents: bytes):
        powers = 2 ** contents[SETUP_FILE_POWERS_POS]  # setup.py:27
        values = [
            int.from_bytes(contents[i : i + 32], "little")
            for i in range(SETUP_FILE_G1_STARTPOS, SETUP_FILE_G1_STARTPOS + 32 * powers * 2, 32)
        ]
        assert max(values) < Q_MOD  # setup.py:36
        factor = values[0] * inv(G1[0], Q_MOD) % Q_MOD  # setup.py:39 — the Montgomery R mod q
        inv_factor = inv(factor, Q_MOD)
        values = [x * inv_factor % Q_MOD for x in values]  # setup.py:40
        powers_of_x = [(values[2 * i], values[2 * i + 1]) for i in range(powers)]
        # setup.py:45-51 — byte-wise scan for the (Montgomery-encoded) G2 generator
        pos = SETUP_FILE_G1_STARTPOS + 32 * powers * 2
        target = factor * G2_X_C0 % Q_MOD
        tbytes = target.to_bytes(32, "little")
        pos = contents.find(tbytes, pos)
        assert pos >= 0
        enc = contents[pos + 32 * 4 : pos + 32 * 8]  # setup.py:53
        xv = [int.from_bytes(enc[i : i + 32], "little") * inv_factor % Q_MOD for i in range(0, 128, 32)]
        X2 = ((xv[0], xv[1]), (xv[2], xv[3]))
        return cls(powers_of_x, X2)

    @classmethod
    def from_file(cls, filename):
        with open(filename, "rb") as f:
            return cls.from_bytes(f.read())

    def commit(self, values: Polynomial):
        """setup.py:66-72."""
        assert values.basis == Basis.LAGRANGE
        coeffs = values.ifft().values
        assert len(coeffs) <= len(self.powers_of_x)
        return ec_lincomb([(self.powers_of_x[i], c) for i, c in enumerate(coeffs)])
