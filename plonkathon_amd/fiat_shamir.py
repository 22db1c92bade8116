"""Fiat-Shamir transcript and proof messages — /root/reference/transcript.py with the Merlin/STROBE/
Keccak work done natively by libplonk_hip.so's host transcript (plonk_transcript_*).

Same message dataclasses, labels, round order and encodings (32-byte big-endian scalars; a G1 point
is two messages x then y) as transcript.py:8-123.
"""
import ctypes
from dataclasses import dataclass

from . import _lib
from ._lib import check
from .field import Scalar


@dataclass
class Message1:  # transcript.py:8-15
    a_1: object
    b_1: object
    c_1: object


@dataclass
class Message2:  # transcript.py:18-21
    z_1: object


@dataclass
class Message3:  # transcript.py:24-31
    t_lo_1: object
    t_mid_1: object
    t_hi_1: object


@dataclass
class Message4:  # transcript.py:34-47
    a_eval: Scalar
    b_eval: Scalar
    c_eval: Scalar
    s1_eval: Scalar
    s2_eval: Scalar
    z_shifted_eval: Scalar


@dataclass
class Message5:  # transcript.py:50-55
    W_z_1: object
    W_zw_1: object


class Transcript:
    def __init__(self, label: bytes):
        self._L = _lib.lib()
        self._h = ctypes.c_void_p()
        check(self._L.plonk_transcript_new(label, len(label), ctypes.byref(self._h)))

    def __del__(self):
        try:
            if self._h:
                self._L.plonk_transcript_free(self._h)
                self._h = None
        except Exception:
            pass

    # merlin.MerlinTranscript interface
    def append_message(self, label: bytes, message: bytes) -> None:
        check(self._L.plonk_transcript_append_message(self._h, label, len(label), message, len(message)))

    def challenge_bytes(self, label: bytes, n: int) -> bytes:
        out = ctypes.create_string_buffer(n)
        check(self._L.plonk_transcript_challenge_bytes(self._h, label, len(label), out, n))
        return out.raw

    # transcript.py:59-75
    def append(self, label: bytes, item: bytes) -> None:
        self.append_message(label, item)

    def append_scalar(self, label: bytes, item: Scalar):
        self.append_message(label, item.n.to_bytes(32, "big"))

    def append_point(self, label: bytes, item):
        self.append_message(label, item[0].n.to_bytes(32, "big"))  # item is None -> TypeError, as upstream
        self.append_message(label, item[1].n.to_bytes(32, "big"))

    def get_and_append_challenge(self, label: bytes) -> Scalar:
        out = ctypes.create_string_buffer(32)
        check(self._L.plonk_transcript_challenge_scalar(self._h, label, len(label), out))
        return Scalar(int.from_bytes(out.raw, "little"))

    # transcript.py:77-123
    def round_1(self, message: Message1):
        self.append_point(b"a_1", message.a_1)
        self.append_point(b"b_1", message.b_1)
        self.append_point(b"c_1", message.c_1)
        beta = self.get_and_append_challenge(b"beta")
        gamma = self.get_and_append_challenge(b"gamma")
        return beta, gamma

    def round_2(self, message: Message2):
        self.append_point(b"z_1", message.z_1)
        alpha = self.get_and_append_challenge(b"alpha")
        fft_cofactor = self.get_and_append_challenge(b"fft_cofactor")
        return alpha, fft_cofactor

    def round_3(self, message: Message3) -> Scalar:
        self.append_point(b"t_lo_1", message.t_lo_1)
        self.append_point(b"t_mid_1", message.t_mid_1)
        self.append_point(b"t_hi_1", message.t_hi_1)
        return self.get_and_append_challenge(b"zeta")

    def round_4(self, message: Message4) -> Scalar:
        self.append_scalar(b"a_eval", message.a_eval)
        self.append_scalar(b"b_eval", message.b_eval)
        self.append_scalar(b"c_eval", message.c_eval)
        self.append_scalar(b"s1_eval", message.s1_eval)
        self.append_scalar(b"s2_eval", message.s2_eval)
        self.append_scalar(b"z_shifted_eval", message.z_shifted_eval)
        return self.get_and_append_challenge(b"v")

    def round_5(self, message: Message5) -> Scalar:
        self.append_point(b"W_z_1", message.W_z_1)
        self.append_point(b"W_zw_1", message.W_zw_1)
        return self.get_and_append_challenge(b"u")
