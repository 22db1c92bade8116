"""`Setup` (trusted-setup loader + KZG commit), `ec_lincomb`, `ec_mul` — the reference's
/root/reference/setup.py:16-77 and /root/reference/curve.py:30-44 on the GPU.

`Setup.from_file` keeps the reference's parsing contract (byte 60 = log2(#powers), G1 from byte 80,
byte-wise scan for the G2 generator, setup.py:23-63) but hands the G1 section's bytes to the
device as they are: a .ptau stores coordinates little-endian in Montgomery form with R = 2^256, and
the library (R = 2^261) rescales them on the GPU by five modular doublings instead of dividing the
factor out on the host as setup.py:39-40 does.  `commit` = inverse NTT + fixed-base Pippenger MSM
(setup.py:66-72).
"""
import ctypes
from dataclasses import dataclass
from typing import Optional

from . import _lib
from ._lib import check
from .backend import get_context
from .field import Fq, Q_MOD, R_MOD, Scalar, le32
from .polynomial import Basis, Polynomial, _log2_exact

SETUP_FILE_G1_STARTPOS = 80  # setup.py:11
SETUP_FILE_POWERS_POS = 60  # setup.py:12

G1 = (Fq(1), Fq(2))
Z1 = None
# py_ecc.bn128.G2: the published generator of the BN254 twist subgroup, (x.c0, x.c1), (y.c0, y.c1)
_G2_X_C0 = 10857046999023057135944570762232829481370756359578518086990519993285655852781
_G2_COORDS = ((_G2_X_C0, 11559732032986387107991004021392285783925812861821192530917403151452391805634),
              (8495653923123431417604973247489272438418190587263600148770280649306958101930,
               4082367875863433681332203403145435568316851327593401208105741076214120093531))
_MONT_R_Q = (1 << 256) % Q_MOD


class Fq2:
    """Minimal container for a G2 coordinate (only equality and `.coeffs` are needed here)."""

    def __init__(self, coeffs):
        self.coeffs = tuple(Fq(c) for c in coeffs)

    def __eq__(self, other):
        return isinstance(other, Fq2) and self.coeffs == other.coeffs

    def __repr__(self):
        return repr(self.coeffs)


G2 = None  # set below, once Fq2 exists


def pairing_check(pairs) -> bool:
    """True iff prod e(P_i, Q_i) == 1 for `pairs` = [(G1 point or None, G2 point)]: the question every
    `b.pairing(...) == b.pairing(...)` of the reference's verifier asks (TESTING_verifier_DO_NOT_OPEN.py:148-160,
    237-262), answered by plonk_pairing_check on the host CPU."""
    pairs = list(pairs)
    g1 = b"".join(bytes(64) if p is None else le32(int(p[0])) + le32(int(p[1])) for p, _ in pairs)
    flags = bytes(1 if p is None else 0 for p, _ in pairs)
    g2 = b"".join(le32(int(q[0].coeffs[0])) + le32(int(q[0].coeffs[1])) + le32(int(q[1].coeffs[0])) + le32(int(q[1].coeffs[1]))
                  for _, q in pairs)
    ok = ctypes.c_int(0)
    check(_lib.lib().plonk_pairing_check(g1, flags, g2, len(pairs), ctypes.byref(ok)))
    return bool(ok.value)


def _decode_points(xy: bytes, flags: bytes):
    out = []
    for i, f in enumerate(flags):
        if f:
            out.append(None)  # py_ecc identity (utils.py:13-14)
        else:
            out.append((Fq(int.from_bytes(xy[64 * i : 64 * i + 32], "little")),
                        Fq(int.from_bytes(xy[64 * i + 32 : 64 * i + 64], "little"))))
    return out


class _DeviceBases:
    """Owns a `plonk_srs*` (bases + window / lookup tables in HBM)."""

    def __init__(self, ctx, handle, n, owner=None):
        self.ctx, self.handle, self.n = ctx, handle, n
        self._owner = owner  # a Lagrange view is owned by (and keeps alive) its parent
        self._views = {}

    def lagrange(self, log_n):
        """The Lagrange-basis view [L_i(tau)]_1, i < 2^log_n (plonk_srs_lagrange; built once, cached on the device)."""
        v = self._views.get(log_n)
        if v is None:
            h = ctypes.c_void_p()
            check(self.ctx.L.plonk_srs_lagrange(self.ctx.handle, self.handle, log_n, ctypes.byref(h)))
            v = self._views[log_n] = _DeviceBases(self.ctx, h, 1 << log_n, owner=self)
        return v

    @property
    def lookup_bits(self):
        """Bits of the table its MSMs run on — teeth of the comb, or window bits of a window table (0: bucket method); built at
        the first MSM."""
        out = ctypes.c_uint(0)
        check(self.ctx.L.plonk_srs_lookup_bits(self.handle, ctypes.byref(out)))
        return out.value

    def lookup_info(self):
        """{bits, bytes, build_s, sharers, layout, additions_per_base, top_bits, top_group} of the table attached to these bases (one
        table per device, base set and layout, shared by every context / stream / prover that loaded the same SRS).  A comb with top
        tables (top_group = g > 0 bases per joint table) performs additions_per_base * (n + ceil(ceil(n / g) / additions_per_base))
        mixed additions per MSM of n scalars: `additions(n)` below."""
        bits, nbytes, secs, sharers = ctypes.c_uint(0), ctypes.c_size_t(0), ctypes.c_double(0), ctypes.c_int(0)
        check(self.ctx.L.plonk_srs_lookup_info(self.handle, ctypes.byref(bits), ctypes.byref(nbytes), ctypes.byref(secs),
                                               ctypes.byref(sharers)))
        kind, adds = ctypes.c_uint(0), ctypes.c_uint(0)
        check(self.ctx.L.plonk_srs_lookup_layout(self.handle, ctypes.byref(kind), ctypes.byref(adds)))
        tb, tg = ctypes.c_uint(0), ctypes.c_uint(0)
        check(self.ctx.L.plonk_srs_lookup_top(self.handle, ctypes.byref(tb), ctypes.byref(tg)))
        return {"bits": bits.value, "bytes": nbytes.value, "build_s": secs.value, "sharers": sharers.value,
                "layout": {0: None, 1: "comb", 2: "windows"}[kind.value], "additions_per_base": adds.value,
                "top_bits": tb.value, "top_group": tg.value}

    @staticmethod
    def table_additions(info, n):
        """Mixed additions of one MSM of n scalars on the table `info` (lookup_info) describes."""
        a, g = info["additions_per_base"], info.get("top_group", 0)
        return a * (n + (-(-(-(-n // g)) // a) if g else 0))

    def __del__(self):
        try:
            if self._owner is not None:  # a view: freed with its parent
                self.handle = None
                return
            self._views = {}
            if self.handle and self.ctx.handle:
                self.ctx.L.plonk_srs_free(self.ctx.handle, self.handle)
                self.handle = None
        except Exception:
            pass


def _msm(bases: _DeviceBases, scalars_ptr, n, batch, stride):
    ctx = bases.ctx
    xy = ctypes.create_string_buffer(64 * batch)
    flags = ctypes.create_string_buffer(batch)
    check(ctx.L.plonk_g1_msm(ctx.handle, bases.handle, scalars_ptr, n, batch, stride, xy, flags))
    return _decode_points(xy.raw, flags.raw[:batch])


@dataclass
class VerificationKey:
    """verifier.py:9-34, with the verification the reference leaves blank (verifier.py:40-92) following its complete
    test verifier, TESTING_verifier_DO_NOT_OPEN.py:39-277.  Group arithmetic = `ec_lincomb` on the GPU, the evaluation of
    PI = `Polynomial.barycentric_eval` on the GPU, challenges = the native transcript, pairings = `pairing_check`
    (host CPU).  Off the prover hot path."""

    group_order: int
    Qm: object
    Ql: object
    Qr: object
    Qo: object
    Qc: object
    S1: object
    S2: object
    S3: object
    X_2: object
    w: Scalar

    # TESTING_verifier_DO_NOT_OPEN.py:266-277 / verifier.py:95-105
    def compute_challenges(self, proof):
        from .fiat_shamir import Transcript

        transcript = Transcript(b"plonk")
        beta, gamma = transcript.round_1(proof.msg_1)
        alpha, _fft_cofactor = transcript.round_2(proof.msg_2)
        zeta = transcript.round_3(proof.msg_3)
        v = transcript.round_4(proof.msg_4)
        u = transcript.round_5(proof.msg_5)
        return beta, gamma, alpha, zeta, v, u

    def _common(self, group_order, zeta, public):
        ZH_ev = zeta**group_order - 1
        L0_ev = ZH_ev / (group_order * (zeta - 1))
        PI = Polynomial([Scalar(-x) for x in public] + [Scalar(0) for _ in range(group_order - len(public))], Basis.LAGRANGE)
        return ZH_ev, L0_ev, PI.barycentric_eval(zeta)

    # TESTING_verifier_DO_NOT_OPEN.py:39-163: one pairing check
    def verify_proof(self, group_order: int, pf, public=[]) -> bool:
        beta, gamma, alpha, zeta, v, u = self.compute_challenges(pf)
        proof = pf.flatten()
        root_of_unity = Scalar.root_of_unity(group_order)
        ZH_ev, L0_ev, PI_ev = self._common(group_order, zeta, public)
        a, b_, c = proof["a_eval"], proof["b_eval"], proof["c_eval"]
        s1, s2, zw = proof["s1_eval"], proof["s2_eval"], proof["z_shifted_eval"]
        r0 = PI_ev - L0_ev * alpha**2 - alpha * (a + beta * s1 + gamma) * (b_ + beta * s2 + gamma) * (c + gamma) * zw
        D_pt = ec_lincomb([
            (self.Qm, a * b_), (self.Ql, a), (self.Qr, b_), (self.Qo, c), (self.Qc, 1),
            (proof["z_1"], (a + beta * zeta + gamma) * (b_ + beta * 2 * zeta + gamma) * (c + beta * 3 * zeta + gamma) * alpha
             + L0_ev * alpha**2 + u),
            (self.S3, -(a + beta * s1 + gamma) * (b_ + beta * s2 + gamma) * alpha * beta * zw),
            (proof["t_lo_1"], -ZH_ev), (proof["t_mid_1"], -ZH_ev * zeta**group_order),
            (proof["t_hi_1"], -ZH_ev * zeta ** (group_order * 2)),
        ])
        F_pt = ec_lincomb([(D_pt, 1), (proof["a_1"], v), (proof["b_1"], v**2), (proof["c_1"], v**3), (self.S1, v**4), (self.S2, v**5)])
        E_pt = ec_mul(G1, -r0 + v * a + v**2 * b_ + v**3 * c + v**4 * s1 + v**5 * s2 + u * zw)
        lhs = ec_lincomb([(proof["W_z_1"], 1), (proof["W_zw_1"], u)])
        rhs = ec_lincomb([(proof["W_z_1"], zeta), (proof["W_zw_1"], u * zeta * root_of_unity), (F_pt, 1), (E_pt, -1)])
        # e(lhs, [x]_2) == e(rhs, [1]_2)
        return pairing_check([(lhs, self.X_2), (_g1_neg(rhs), G2)])

    # TESTING_verifier_DO_NOT_OPEN.py:166-264: the two opening checks separately.  e(Y, [1]_2) == e(W, [x]_2 - z [1]_2) is
    # asked as e(Y + z W, [1]_2) e(-W, [x]_2) == 1, so no arithmetic in G2 is needed.
    def verify_proof_unoptimized(self, group_order: int, pf, public=[]) -> bool:
        beta, gamma, alpha, zeta, v, _ = self.compute_challenges(pf)
        proof = pf.flatten()
        root_of_unity = Scalar.root_of_unity(group_order)
        ZH_ev, L0_ev, PI_ev = self._common(group_order, zeta, public)
        a, b_, c = proof["a_eval"], proof["b_eval"], proof["c_eval"]
        s1, s2, zw = proof["s1_eval"], proof["s2_eval"], proof["z_shifted_eval"]
        R_pt = ec_lincomb([
            (self.Qm, a * b_), (self.Ql, a), (self.Qr, b_), (self.Qo, c), (G1, PI_ev), (self.Qc, 1),
            (proof["z_1"], (a + beta * zeta + gamma) * (b_ + beta * 2 * zeta + gamma) * (c + beta * 3 * zeta + gamma) * alpha),
            (self.S3, -(a + beta * s1 + gamma) * (b_ + beta * s2 + gamma) * beta * alpha * zw),
            (G1, -(a + beta * s1 + gamma) * (b_ + beta * s2 + gamma) * (c + gamma) * alpha * zw),
            (proof["z_1"], L0_ev * alpha**2), (G1, -L0_ev * alpha**2),
            (proof["t_lo_1"], -ZH_ev), (proof["t_mid_1"], -ZH_ev * zeta**group_order),
            (proof["t_hi_1"], -ZH_ev * zeta ** (group_order * 2)),
        ])
        Y1 = ec_lincomb([
            (R_pt, 1), (proof["a_1"], v), (G1, -v * a), (proof["b_1"], v**2), (G1, -(v**2) * b_), (proof["c_1"], v**3),
            (G1, -(v**3) * c), (self.S1, v**4), (G1, -(v**4) * s1), (self.S2, v**5), (G1, -(v**5) * s2),
            (proof["W_z_1"], zeta),
        ])
        if not pairing_check([(Y1, G2), (_g1_neg(proof["W_z_1"]), self.X_2)]):
            return False
        Y2 = ec_lincomb([(proof["z_1"], 1), (G1, -zw), (proof["W_zw_1"], zeta * root_of_unity)])
        return pairing_check([(Y2, G2), (_g1_neg(proof["W_zw_1"]), self.X_2)])


def _g1_neg(pt):
    return None if pt is None else (pt[0], Fq(-int(pt[1])))


G2 = (Fq2(_G2_COORDS[0]), Fq2(_G2_COORDS[1]))


# Setup.commit and the Lagrange-basis SRS.  Up to 2^LAGRANGE_SRS_EAGER_LOG the view is built by the FIRST commit of a size (n MSMs
# of size n: a few milliseconds at the prover's sizes).  Above it the view is an inverse DFT over the group (csrc/g1_ntt.hip: log n
# stages of n / 2 scalar multiplications, tens of milliseconds at 2^16, about a second at 2^20) plus 2 n 128 B of staging, and a fixed
# SRS then gets a comb table of its own charged to the budget: that pays only when a size is committed repeatedly, so the first
# commit of such a size takes the reference's route (ifft + one MSM, setup.py:66-72) and the SECOND builds the view.  Beyond
# 2^LAGRANGE_SRS_MAX_LOG no view is built at all (the largest size the GPU tests cover the group transform at is 2^16).
LAGRANGE_SRS_EAGER_LOG = 12
LAGRANGE_SRS_MAX_LOG = 20


class Setup:
    def __init__(self, powers_of_x=None, X2=None, _g1_mont_bytes: Optional[bytes] = None):
        self._powers = powers_of_x
        self.X2 = X2
        if _g1_mont_bytes is None:
            # built from explicit affine points: re-encode in the .ptau layout
            _g1_mont_bytes = b"".join(
                le32(p[0].n * _MONT_R_Q % Q_MOD) + le32(p[1].n * _MONT_R_Q % Q_MOD) for p in powers_of_x
            )
        self._raw = _g1_mont_bytes
        self._n = len(_g1_mont_bytes) // 64
        self._dev = {}  # per-context device copies (bases + window table)
        self._commits = {}  # log2(size) -> Lagrange-value commits seen (a large size builds its view on the second)

    # setup.py:23-63
    @classmethod
    def from_file(cls, filename):
        with open(filename, "rb") as f:
            contents = f.read()
        powers = 2 ** contents[SETUP_FILE_POWERS_POS]
        g1_end = SETUP_FILE_G1_STARTPOS + 64 * powers
        raw = contents[SETUP_FILE_G1_STARTPOS:g1_end]
        assert len(raw) == 64 * powers
        factor = int.from_bytes(raw[:32], "little")  # = R mod q, because G1[0] is the generator (1, 2)
        assert factor * pow(G1[0].n, -1, Q_MOD) % Q_MOD == _MONT_R_Q, "unexpected .ptau encoding"
        inv_factor = pow(factor, -1, Q_MOD)
        target = (factor * _G2_X_C0 % Q_MOD).to_bytes(32, "little")
        pos = contents.find(target, g1_end)  # setup.py:45-51
        assert pos >= 0, "G2 generator not found in the setup file"
        enc = contents[pos + 32 * 4 : pos + 32 * 8]
        xv = [int.from_bytes(enc[i : i + 32], "little") * inv_factor % Q_MOD for i in range(0, 128, 32)]
        X2 = (Fq2(xv[:2]), Fq2(xv[2:]))
        return cls(None, X2, raw)

    @property
    def powers_of_x(self):
        """list[(Fq, Fq)] — decoded lazily from the .ptau bytes (setup.py:39-41)."""
        if self._powers is None:
            inv_factor = pow(_MONT_R_Q, -1, Q_MOD)
            vals = [int.from_bytes(self._raw[i : i + 32], "little") for i in range(0, len(self._raw), 32)]
            assert max(vals) < Q_MOD  # setup.py:36
            self._powers = [(Fq(vals[2 * i] * inv_factor), Fq(vals[2 * i + 1] * inv_factor)) for i in range(self._n)]
        return self._powers

    def device_bases(self, ctx=None) -> _DeviceBases:
        ctx = ctx or get_context()
        dev = self._dev.get(id(ctx))
        if dev is None:
            h = ctypes.c_void_p()
            check(ctx.L.plonk_srs_load_ptau(ctx.handle, self._raw, self._n, ctypes.byref(h)))
            dev = self._dev[id(ctx)] = _DeviceBases(ctx, h, self._n)
        return dev

    # setup.py:66-72
    def commit(self, values: Polynomial):
        """KZG commitment of Lagrange values.  The reference runs an ifft and then a lincomb with powers_of_x; here the
        SRS itself is taken to the Lagrange basis once per size (an inverse DFT over the group, on the device) and the
        commitment is a single MSM of the values — the same group element."""
        assert values.basis == Basis.LAGRANGE
        n = len(values)
        assert n <= self._n  # setup.py:70
        bases, log_n = self.device_bases(), _log2_exact(n)
        seen = self._commits.get(log_n, 0)
        self._commits[log_n] = seen + 1
        if log_n not in bases._views and (log_n > LAGRANGE_SRS_MAX_LOG or (log_n > LAGRANGE_SRS_EAGER_LOG and seen == 0)):
            return self.commit_coeffs(values.ifft())
        lag = bases.lagrange(log_n)
        return _msm(lag, values.device().ptr, n, 1, n)[0]

    def commit_many(self, polys):
        """[commit(p) for p in polys] (Lagrange values) or [commit_coeffs(p) ...] (MONOMIAL) for polynomials of one size and
        basis as ONE batched MSM call: a lone 2^11 commitment is latency-bound (64 workgroups and a host synchronisation), so
        the three commitments of round 1 or round 3 (prover.py:105-107, 221-223) cost the time of one.  The scalar vectors
        are read in place when they are consecutive views of one buffer, gathered into one buffer otherwise."""
        polys = list(polys)
        n, basis = len(polys[0]), polys[0].basis
        assert all(len(p) == n and p.basis == basis for p in polys) and n <= self._n
        bases = self.device_bases()
        if basis == Basis.LAGRANGE:
            log_n = _log2_exact(n)
            if log_n > LAGRANGE_SRS_MAX_LOG and log_n not in bases._views:
                return [self.commit(p) for p in polys]
            bases = bases.lagrange(log_n)
        ctx = bases.ctx
        ptrs = [p.device().ptr.value for p in polys]
        if all(ptrs[k] == ptrs[0] + 32 * n * k for k in range(len(ptrs))):
            return _msm(bases, polys[0].device().ptr, n, len(polys), n)
        stacked = ctx.alloc(n * len(polys))
        for k, p in enumerate(polys):
            check(ctx.L.plonk_mem_d2d(ctx.handle, stacked.at(n * k), p.device().ptr, 32 * n))
        return _msm(bases, stacked.ptr, n, len(polys), n)

    def commit_coeffs(self, coeffs: Polynomial):
        """KZG commitment of a polynomial already in MONOMIAL basis (skips the ifft)."""
        assert coeffs.basis == Basis.MONOMIAL
        assert len(coeffs) <= self._n
        return _msm(self.device_bases(), coeffs.device().ptr, len(coeffs), 1, len(coeffs))[0]

    # setup.py:75-77
    def verification_key(self, pk) -> VerificationKey:
        polys = (pk.QM, pk.QL, pk.QR, pk.QO, pk.QC, pk.S1, pk.S2, pk.S3)
        c = [self.commit(p) for p in polys]
        return VerificationKey(pk.group_order, c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], self.X2,
                               Scalar.root_of_unity(pk.group_order))


def g1_compress(points):
    """Affine points (None = identity) -> 32 bytes each: x big-endian as append_point writes it (transcript.py:62-67)
    with the two spare top bits carrying y's root (10 smaller, 11 larger) or infinity (01).  Batched on the device
    (plonk_g1_compress)."""
    points = list(points)
    ctx = get_context()
    xy = b"".join((le32(0) + le32(0)) if p is None else (le32(int(p[0])) + le32(int(p[1]))) for p in points)
    out = ctypes.create_string_buffer(32 * max(len(points), 1))
    check(ctx.L.plonk_g1_compress(ctx.handle, xy, len(points), out))
    return out.raw[: 32 * len(points)]


def g1_decompress(blob):
    """The inverse: bytes -> list of affine points (None = identity); ValueError on a malformed or off-curve encoding."""
    assert len(blob) % 32 == 0
    n = len(blob) // 32
    ctx = get_context()
    xy = ctypes.create_string_buffer(64 * max(n, 1))
    status = ctypes.create_string_buffer(max(n, 1))
    check(ctx.L.plonk_g1_decompress(ctx.handle, bytes(blob), n, xy, status))
    out = []
    for i in range(n):
        st = status.raw[i]
        if st:
            raise ValueError("compressed point %d: %s" % (i, "malformed encoding" if st == 1 else "not on the curve"))
        x = int.from_bytes(xy.raw[64 * i : 64 * i + 32], "little")
        y = int.from_bytes(xy.raw[64 * i + 32 : 64 * i + 64], "little")
        out.append(None if (blob[32 * i] & 0xC0) == 0x40 else (Fq(x), Fq(y)))
    return out


def ec_mul(pt, coeff):  # curve.py:30-33
    return ec_lincomb([(pt, coeff)])


def ec_lincomb(pairs):
    """curve.py:38-44: sum_i coeff_i * pt_i for arbitrary affine points (None = identity)."""
    pairs = list(pairs)
    if not pairs:
        raise ValueError("max() arg is an empty sequence")  # what curve.py:93 raises
    ctx = get_context()
    xy = b"".join(
        (le32(0) + le32(0)) if p is None else (le32(int(p[0])) + le32(int(p[1]))) for p, _ in pairs
    )
    h = ctypes.c_void_p()
    check(ctx.L.plonk_srs_load_affine(ctx.handle, xy, len(pairs), ctypes.byref(h)))
    bases = _DeviceBases(ctx, h, len(pairs))
    scalars = ctx.upload_ints([int(n) % R_MOD for _, n in pairs])  # curve.py:41
    return _msm(bases, scalars.ptr, len(pairs), 1, len(pairs))[0]


def _is_g1(x):
    """A BN254 G1 point as this package (and py_ecc) represents it: None or a pair of Fq — a generic 2-tuple is NOT one, so that
    `lincomb` / `multisubset` over any other group take the host fold the reference performs (curve.py:59-111)."""
    return x is None or (isinstance(x, tuple) and len(x) == 2 and isinstance(x[0], Fq) and isinstance(x[1], Fq))


def multisubset(numbers, subsets, adder=None, zero=0):
    """curve.py:59-87's generic entry: for every subset (an iterable of indices) the sum of numbers[i] under `adder`, from `zero`.
    G1 points under the curve's own addition (adder None) are summed on the GPU, one MSM with 0 / 1 coefficients per subset; any
    other group — the reference's self-test runs this on plain integers (test.py's K8 vector) — is folded on the host."""
    numbers = list(numbers)
    if adder is None and numbers and all(_is_g1(x) for x in numbers):
        return [ec_lincomb([(numbers[i], 1) for i in sub]) if sub else None for sub in subsets]
    add = adder if adder is not None else (lambda x, y: x + y)
    out = []
    for sub in subsets:
        acc = zero
        for i in sorted(sub):
            acc = add(acc, numbers[i])
        out.append(acc)
    return out


def lincomb(numbers, factors, adder=None, zero=0):
    """curve.py:91-111's generic entry: numbers[0] * factors[0] + numbers[1] * factors[1] + ... under `adder`, from `zero`.
    G1 points under the curve's own addition go to the GPU (`ec_lincomb`); for any other group the sum is formed on the host by a
    left-to-right binary method written for this file: per bit of the longest factor one doubling of the running sum and one
    addition per number whose factor has that bit set (the reference partitions the numbers into power-set tables instead;
    both compute the same group element, which is all its K8 self-test compares)."""
    numbers, factors = list(numbers), [int(f) for f in factors]
    if adder is None and numbers and all(_is_g1(x) for x in numbers):
        return ec_lincomb(zip(numbers, factors))
    add = adder if adder is not None else (lambda x, y: x + y)
    assert all(f >= 0 for f in factors)
    acc = zero
    for bit in range(max(f.bit_length() for f in factors) - 1, -1, -1):  # (an empty list raises ValueError, as curve.py:93 does)
        acc = add(acc, acc)
        for x, f in zip(numbers, factors):
            if (f >> bit) & 1:
                acc = add(acc, x)
    return acc

