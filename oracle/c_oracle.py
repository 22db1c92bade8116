"""ctypes wrapper of oracle/c/liboracle_c.so.  (oracle: test infrastructure only)"""
import ctypes
import os
import subprocess

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c")
_lib = None


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(_DIR, "liboracle_c.so")
        if not os.path.exists(so):
            subprocess.run(["make", "-s", "-C", _DIR], check=True)
        _lib = ctypes.CDLL(so)
    return _lib


def fr_ntt(values, inverse=False):
    """poly.py:113-148 on a list of canonical ints (length a power of two)."""
    n = len(values)
    log_n = n.bit_length() - 1
    assert 1 << log_n == n
    buf = (ctypes.c_uint64 * (4 * n))()
    raw = b"".join(int(v).to_bytes(32, "little") for v in values)
    ctypes.memmove(buf, raw, 32 * n)
    rc = lib().oracle_fr_ntt(buf, ctypes.c_uint(log_n), ctypes.c_int(1 if inverse else 0))
    assert rc == 0
    out = bytes(buf)
    return [int.from_bytes(out[32 * i : 32 * i + 32], "little") for i in range(n)]


def fr_ntt_bytes(raw, inverse=False, field="bn254"):
    """The same transform on canonical 32-byte little-endian elements back to back (sizes where Python ints are too slow).
    field = "bls12_381": the BLS12-381 scalar field, generator 7 (no counterpart in the reference; see bn254_oracle.c)."""
    n = len(raw) // 32
    log_n = n.bit_length() - 1
    assert 1 << log_n == n and len(raw) == 32 * n
    buf = (ctypes.c_uint64 * (4 * n)).from_buffer_copy(raw)
    fn = {"bn254": lib().oracle_fr_ntt, "bls12_381": lib().oracle_bls_fr_ntt}[field]
    rc = fn(buf, ctypes.c_uint(log_n), ctypes.c_int(1 if inverse else 0))
    assert rc == 0
    return bytes(buf)


BLS12_381_FR_MODULUS = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def g1_lincomb(points, scalars):
    """curve.py:38-49 on affine int tuples (None = identity) and canonical scalars."""
    n = len(points)
    pts = b"".join((b"\0" * 64) if p is None else (int(p[0]).to_bytes(32, "little") + int(p[1]).to_bytes(32, "little")) for p in points)
    sc = b"".join(int(s).to_bytes(32, "little") for s in scalars)
    pb = (ctypes.c_uint64 * (8 * n)).from_buffer_copy(pts)
    sb = (ctypes.c_uint64 * (4 * n)).from_buffer_copy(sc)
    out = (ctypes.c_uint64 * 8)()
    ident = ctypes.c_int(0)
    rc = lib().oracle_g1_lincomb(pb, sb, ctypes.c_size_t(n), out, ctypes.byref(ident))
    assert rc == 0
    if ident.value:
        return None
    raw = bytes(out)
    return (int.from_bytes(raw[:32], "little"), int.from_bytes(raw[32:], "little"))
