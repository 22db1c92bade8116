"""Merlin v1.0 transcript over STROBE-128 / Keccak-f[1600], and plonkathon's Fiat-Shamir layer.
(oracle: test infrastructure only)

`merlin` is a third-party dependency that is NOT under /root/reference (pyproject.toml:12,
pinned at rev 805d0678 in poetry.lock:255-269).  It is restated here from the published Merlin
v1.0 / STROBE v1.0.2 specification; conformance is pinned by the merlin
crate's public "simple transcript" vector and, end to end, by test/proof.pickle (K6).

The plonkathon layer follows /root/reference/transcript.py:58-123.
"""
from .field import R_MOD

_RC = [
    0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000,
    0x000000000000808B, 0x0000000080000001, 0x8000000080008081, 0x8000000000008009,
    0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
    0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003,
    0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
    0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008,
]
_ROT = [
    [0, 36, 3, 41, 18],
    [1, 44, 10, 45, 2],
    [62, 6, 43, 15, 61],
    [28, 55, 25, 21, 56],
    [27, 20, 39, 8, 14],
]
_M64 = (1 << 64) - 1


def _rol(x, n):
    n %= 64
    return ((x << n) | (x >> (64 - n))) & _M64 if n else x


def keccak_f1600(state: bytearray) -> None:
    """In-place permutation of a 200-byte state (lanes little-endian, lane (x,y) at 8*(x+5y))."""
    a = [[int.from_bytes(state[8 * (x + 5 * y) : 8 * (x + 5 * y) + 8], "little") for y in range(5)] for x in range(5)]
    for rnd in range(24):
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ _rol(c[(x + 1) % 5], 1) for x in range(5)]
        for x in range(5):
            for y in range(5):
                a[x][y] ^= d[x]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                b[y][(2 * x + 3 * y) % 5] = _rol(a[x][y], _ROT[x][y])
        for x in range(5):
            for y in range(5):
                a[x][y] = b[x][y] ^ ((~b[(x + 1) % 5][y]) & _M64 & b[(x + 2) % 5][y])
        a[0][0] ^= _RC[rnd]
    for x in range(5):
        for y in range(5):
            state[8 * (x + 5 * y) : 8 * (x + 5 * y) + 8] = a[x][y].to_bytes(8, "little")


def sha3_256_selfcheck(data: bytes) -> bytes:
    """Plain SHA3-256 sponge built on keccak_f1600 — used only to self-check the permutation
    against hashlib.sha3_256 in the tests."""
    rate = 136
    st = bytearray(200)
    msg = bytearray(data) + b"\x06"
    while len(msg) % rate:
        msg += b"\x00"
    msg[-1] |= 0x80
    for off in range(0, len(msg), rate):
        for i in range(rate):
            st[i] ^= msg[off + i]
        keccak_f1600(st)
    return bytes(st[:32])


FLAG_I, FLAG_A, FLAG_C, FLAG_T, FLAG_M, FLAG_K = 1, 2, 4, 8, 16, 32
STROBE_R = 166


class Strobe128:
    def __init__(self, protocol_label: bytes):
        st = bytearray(200)
        st[0:6] = bytes([1, STROBE_R + 2, 1, 0, 1, 96])
        st[6:18] = b"STROBEv1.0.2"
        keccak_f1600(st)
        self.st = st
        self.pos = 0
        self.pos_begin = 0
        self.cur_flags = 0
        self.meta_ad(protocol_label, False)

    def _run_f(self):
        self.st[self.pos] ^= self.pos_begin
        self.st[self.pos + 1] ^= 0x04
        self.st[STROBE_R + 1] ^= 0x80
        keccak_f1600(self.st)
        self.pos = 0
        self.pos_begin = 0

    def _absorb(self, data: bytes):
        for b in data:
            self.st[self.pos] ^= b
            self.pos += 1
            if self.pos == STROBE_R:
                self._run_f()

    def _squeeze(self, n: int) -> bytes:
        out = bytearray(n)
        for i in range(n):
            out[i] = self.st[self.pos]
            self.st[self.pos] = 0
            self.pos += 1
            if self.pos == STROBE_R:
                self._run_f()
        return bytes(out)

    def _begin_op(self, flags: int, more: bool):
        if more:
            assert self.cur_flags == flags
            return
        assert not (flags & FLAG_T)
        old_begin = self.pos_begin
        self.pos_begin = self.pos + 1
        self.cur_flags = flags
        self._absorb(bytes([old_begin, flags]))
        if (flags & (FLAG_C | FLAG_K)) and self.pos != 0:
            self._run_f()

    def meta_ad(self, data: bytes, more: bool):
        self._begin_op(FLAG_M | FLAG_A, more)
        self._absorb(data)

    def ad(self, data: bytes, more: bool):
        self._begin_op(FLAG_A, more)
        self._absorb(data)

    def prf(self, n: int, more: bool) -> bytes:
        self._begin_op(FLAG_I | FLAG_A | FLAG_C, more)
        return self._squeeze(n)


class MerlinTranscript:
    def __init__(self, label: bytes):
        self.strobe = Strobe128(b"Merlin v1.0")
        self.append_message(b"dom-sep", label)

    def append_message(self, label: bytes, message: bytes) -> None:
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(len(message).to_bytes(4, "little"), True)
        self.strobe.ad(message, False)

    def challenge_bytes(self, label: bytes, n: int) -> bytes:
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(n.to_bytes(4, "little"), True)
        return self.strobe.prf(n, False)


class Transcript(MerlinTranscript):
    """transcript.py:58-123.  Points are affine (x, y) int tuples; scalars are ints."""

    def append_scalar(self, label: bytes, item: int):  # transcript.py:62-63
        self.append_message(label, int(item).to_bytes(32, "big"))

    def append_point(self, label: bytes, item):  # transcript.py:65-67 (None crashes, as upstream)
        self.append_message(label, int(item[0]).to_bytes(32, "big"))
        self.append_message(label, int(item[1]).to_bytes(32, "big"))

    def get_and_append_challenge(self, label: bytes) -> int:  # transcript.py:69-75
        while True:
            challenge_bytes = self.challenge_bytes(label, 255)
            f = int.from_bytes(challenge_bytes, "big") % R_MOD
            if f != 0:
                self.append_message(label, challenge_bytes)
                return f

    def round_1(self, a_1, b_1, c_1):  # transcript.py:77-86
        self.append_point(b"a_1", a_1)
        self.append_point(b"b_1", b_1)
        self.append_point(b"c_1", c_1)
        beta = self.get_and_append_challenge(b"beta")
        gamma = self.get_and_append_challenge(b"gamma")
        return beta, gamma

    def round_2(self, z_1):  # transcript.py:88-97
        self.append_point(b"z_1", z_1)
        alpha = self.get_and_append_challenge(b"alpha")
        fft_cofactor = self.get_and_append_challenge(b"fft_cofactor")
        return alpha, fft_cofactor

    def round_3(self, t_lo_1, t_mid_1, t_hi_1):  # transcript.py:99-105
        self.append_point(b"t_lo_1", t_lo_1)
        self.append_point(b"t_mid_1", t_mid_1)
        self.append_point(b"t_hi_1", t_hi_1)
        return self.get_and_append_challenge(b"zeta")

    def round_4(self, a_eval, b_eval, c_eval, s1_eval, s2_eval, z_shifted_eval):  # transcript.py:107-116
        self.append_scalar(b"a_eval", a_eval)
        self.append_scalar(b"b_eval", b_eval)
        self.append_scalar(b"c_eval", c_eval)
        self.append_scalar(b"s1_eval", s1_eval)
        self.append_scalar(b"s2_eval", s2_eval)
        self.append_scalar(b"z_shifted_eval", z_shifted_eval)
        return self.get_and_append_challenge(b"v")

    def round_5(self, W_z_1, W_zw_1):  # transcript.py:118-123
        self.append_point(b"W_z_1", W_z_1)
        self.append_point(b"W_zw_1", W_zw_1)
        return self.get_and_append_challenge(b"u")
