"""The standalone NTT / iNTT / coset-NTT over the BLS12-381 scalar field, on the MI355X (csrc/ntt_bls.hip behind plonk_bls_fr_*).

The reference is BN254 throughout (/root/reference/curve.py:2, 10-11); this module exists because BASELINE.json's
north_star quotes a standalone NTT metric on this field.  It is the reference's transform (poly.py:113-148: natural order
in and out, the inverse includes 1/N) with the modulus and generator swapped: w = 7^((r-1)/N), so w_{2^32} is the
ROOT_OF_UNITY constant of the `bls12_381` crate.  Sizes: 2^8 .. 2^13 (one launch) and 2^14 .. 2^26 (two).
"""
from ._lib import check
from .backend import DeviceBuffer, get_context

MODULUS = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
GENERATOR = 7
TWO_ADICITY = 32


def root_of_unity(order: int) -> int:
    """curve.py:14-16 for this field."""
    assert order & (order - 1) == 0 and order <= 1 << TWO_ADICITY
    return pow(GENERATOR, (MODULUS - 1) // order, MODULUS)


def upload(raw: bytes) -> DeviceBuffer:
    """Canonical 32-byte little-endian elements -> device (Montgomery form); a value >= r raises."""
    assert len(raw) % 32 == 0
    ctx = get_context()
    buf = ctx.alloc(len(raw) // 32)
    check(ctx.L.plonk_bls_fr_upload(ctx.handle, buf.ptr, bytes(raw), len(raw) // 32))
    return buf


def download(buf: DeviceBuffer, count=None, offset=0) -> bytes:
    import ctypes

    ctx = get_context()
    count = buf.n - offset if count is None else count
    out = ctypes.create_string_buffer(32 * count)
    check(ctx.L.plonk_bls_fr_download(ctx.handle, out, buf.at(offset), count))
    return out.raw


def ntt(buf: DeviceBuffer, log_n: int, inverse=False, batch=1, out: DeviceBuffer = None) -> DeviceBuffer:
    """`batch` transforms of 2^log_n points laid out back to back; out may be buf (in place)."""
    ctx = get_context()
    assert buf.n >= batch << log_n
    out = ctx.alloc(batch << log_n) if out is None else out
    check(ctx.L.plonk_bls_fr_ntt(ctx.handle, buf.ptr, out.ptr, log_n, 1 if inverse else 0, batch))
    return out


def coset_extend(buf: DeviceBuffer, log_n: int, offset: int, batch=1) -> DeviceBuffer:
    """poly.py:156-163 over this field: `batch` vectors of 2^log_n Lagrange values -> their 4 * 2^log_n values on offset * <w_4n>."""
    ctx = get_context()
    assert buf.n >= batch << log_n
    out = ctx.alloc(batch << (log_n + 2))
    check(ctx.L.plonk_bls_fr_coset_extend(ctx.handle, buf.ptr, out.ptr, log_n, int(offset).to_bytes(32, "little"), batch))
    return out


def coset_to_coeffs(buf: DeviceBuffer, log_m: int, offset: int, batch=1) -> DeviceBuffer:
    """poly.py:169-177 over this field: 2^log_m values on the coset -> 2^log_m coefficients."""
    ctx = get_context()
    assert buf.n >= batch << log_m
    out = ctx.alloc(batch << log_m)
    check(ctx.L.plonk_bls_fr_coset_to_coeffs(ctx.handle, buf.ptr, out.ptr, log_m, int(offset).to_bytes(32, "little"), batch))
    return out


def ntt_ints(values, inverse=False):
    """Convenience: one transform of a list of ints."""
    n = len(values)
    log_n = n.bit_length() - 1
    assert 1 << log_n == n
    raw = download(ntt(upload(b"".join(int(v).to_bytes(32, "little") for v in values)), log_n, inverse))
    return [int.from_bytes(raw[32 * i : 32 * i + 32], "little") for i in range(n)]
