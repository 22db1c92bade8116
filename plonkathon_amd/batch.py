"""`BatchProver` — many proofs of one circuit in lock-step, entirely GPU-resident.

The throughput form of /root/reference/prover.py's `Prover`: same constructor arguments
(`setup`, `program`), `prove(witness)` for one proof and `prove_batch(witnesses)` for many; returns
the same `Proof` objects.  All five rounds and the Fiat-Shamir transcript run on the device
(plonk_prover_* in include/plonk_hip.h), with one host synchronisation per batch.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import check
from .backend import get_context
from .circuit import Program
from .field import Fq, R_MOD, Scalar
from .fiat_shamir import Message1, Message2, Message3, Message4, Message5
from .kzg import Setup
from .plonk import Proof
from .polynomial import _log2_exact


class ProofError(AssertionError):
    """The witness does not satisfy the circuit (the reference's in-prover asserts fail)."""


def _le(vals):
    return b"".join(int(v).to_bytes(32, "little") for v in vals)


try:  # host-side marshalling helper (csrc/pyext/pypack.c, built by __graft_entry__.build()); same bytes either way
    from ._pypack import pack_dicts_le32 as _pack_witnesses
except ImportError:  # pragma: no cover - pure-Python equivalent of the packer (not a compute fallback)
    def _pack_witnesses(witnesses, keys, modulus):
        return b"".join([(int(w[k]) % modulus).to_bytes(32, "little") for w in witnesses for k in keys])


class BatchProver:
    def __init__(self, setup: Setup, program: Program, ctx=None, lagrange_commits=False):
        """`ctx`: the Context (HIP stream) to run on; several BatchProvers on distinct contexts of one
        GPU overlap each other's latency-bound kernels (transcript, inversions) with MSM work.
        `lagrange_commits`: commit a_1, b_1, c_1, z_1 from Lagrange values over the Lagrange-basis SRS
        (PLONK_PROVER_LAGRANGE_COMMITS) instead of from coefficient forms; same proofs."""
        self.group_order = program.group_order
        self.setup = setup
        self.program = program
        self.ctx = ctx or get_context()
        self._public_vars = program.get_public_assignments()
        self._wires = [w.as_list() for w in program.wires()]
        n = self.group_order
        # wire cell -> variable index; one extra all-zero slot serves the empty cells and the padding rows
        variables, self._cell_index = program.wiring_table()
        self._vars = list(variables)
        pos = {v: i for i, v in enumerate(self._vars)}
        L, R, M, O, C = program.gate_columns()
        sigma = program.permutation_columns()
        sel = _le(M) + _le(L) + _le(R) + _le(O) + _le(C) + _le(sigma[1]) + _le(sigma[2]) + _le(sigma[3])
        self._bases = setup.device_bases(self.ctx)
        self._h = ctypes.c_void_p()
        check(self.ctx.L.plonk_prover_create(self.ctx.handle, self._bases.handle, _log2_exact(n), sel,
                                             len(self._public_vars), ctypes.byref(self._h)))
        self._resident = 0
        if lagrange_commits:
            check(self.ctx.L.plonk_prover_set_options(self._h, 1))
        # the wiring goes to the device once; a batch is then only the variables' values (V x 32 B per proof)
        self._getter = None
        if self._vars:
            cells = np.ascontiguousarray(self._cell_index, dtype=np.uint32)
            pubs = np.ascontiguousarray([pos[v] for v in self._public_vars], dtype=np.uint32)
            check(self.ctx.L.plonk_prover_set_wiring(self._h, cells.ctypes.data, pubs.ctypes.data if len(pubs) else None,
                                                     len(self._vars)))
            self._getter = True
            self._var_keys = tuple(self._vars)

    def __del__(self):
        try:
            if self._h and self.ctx.handle:
                self.ctx.L.plonk_prover_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # ---- inputs ---------------------------------------------------------------------------------
    def wire_columns(self, witness):
        """A, B, C value columns of round 1 (prover.py:94-103): witness[None] = 0, zero padded."""
        n = self.group_order
        cols = [[0] * n, [0] * n, [0] * n]
        get = witness.get
        for i, (wl, wr, wo) in enumerate(self._wires):
            cols[0][i] = (get(wl, 0) if wl is not None else 0) % R_MOD
            cols[1][i] = (get(wr, 0) if wr is not None else 0) % R_MOD
            cols[2][i] = (get(wo, 0) if wo is not None else 0) % R_MOD
            if (wl is not None and wl not in witness) or (wr is not None and wr not in witness) or (
                wo is not None and wo not in witness
            ):
                raise KeyError([w for w in (wl, wr, wo) if w is not None and w not in witness][0])
        return cols

    def upload(self, witnesses):
        """Stage a batch of witnesses in HBM: each variable's value is encoded once (V x 32 bytes per proof) and
        the wire columns A, B, C + public inputs are gathered from them on the device (prover.py:94-103, 57-62).
        A KeyError names a missing variable."""
        B = len(witnesses)
        if self._getter is None:
            return self._upload_columns(witnesses)
        enc = _pack_witnesses(witnesses, self._var_keys, R_MOD)
        check(self.ctx.L.plonk_prover_upload_variables(self._h, enc, B))
        self._resident = B

    def upload_values(self, blob, B):
        """The same staging for callers that produce witnesses natively: `blob` = [B][V] canonical 32-byte little-endian
        values in the order of `self.variables` (64 KiB per proof at 2^11; no Python work per value)."""
        if len(blob) != 32 * B * len(self._vars):
            raise ValueError("upload_values: expected %d bytes" % (32 * B * len(self._vars)))
        check(self.ctx.L.plonk_prover_upload_variables(self._h, blob, B))
        self._resident = B

    def upload_values_async(self, pinned, B):
        """upload_values without a host wait: `pinned` = a Context.host_alloc buffer holding [B][V] canonical 32-byte
        values, which must stay untouched until this batch has been downloaded.  The copy runs on the context's copy
        stream and overlaps the kernels of the compute stream; a non-canonical value shows up as status bit 3."""
        if len(pinned) < 32 * B * len(self._vars):
            raise ValueError("upload_values_async: expected %d bytes" % (32 * B * len(self._vars)))
        check(self.ctx.L.plonk_prover_upload_variables_async(self._h, ctypes.addressof(pinned), B))
        self._resident = B

    @property
    def variables(self):
        """Variable names in the column order `upload_values` expects."""
        return tuple(self._vars)

    def _upload_columns(self, witnesses):
        """The [3][B][n] column form of the same upload (circuits without variables; plonk_prover_upload_witness)."""
        B = len(witnesses)
        n, V = self.group_order, len(self._vars)
        abc = np.empty((3, B, n, 4), dtype=np.uint64)  # 32-byte little-endian elements as 4 x u64
        flat_index = self._cell_index.ravel()
        zero = bytes(32)
        for b, w in enumerate(witnesses):
            enc = b"".join([(w[v] % R_MOD).to_bytes(32, "little") for v in self._vars]) + zero
            table = np.frombuffer(enc, dtype=np.uint64).reshape(V + 1, 4)
            abc[:, b] = np.take(table, flat_index, axis=0).reshape(3, n, 4)
        abc = abc.tobytes()
        pub = b"".join(_le([w[v] % R_MOD for v in self._public_vars]) for w in witnesses)
        check(self.ctx.L.plonk_prover_upload_witness(self._h, abc, pub if self._public_vars else None, B))
        self._resident = B

    def upload_raw(self, abc_bytes, pub_bytes, B):
        check(self.ctx.L.plonk_prover_upload_witness(self._h, abc_bytes, pub_bytes, B))
        self._resident = B

    # ---- proving --------------------------------------------------------------------------------
    def run(self, B=None):
        """Enqueue the five rounds for the resident witnesses (asynchronous)."""
        B = self._resident if B is None else B
        check(self.ctx.L.plonk_prover_run(self._h, B))

    def download_raw(self, B=None):
        B = self._resident if B is None else B
        out = ctypes.create_string_buffer(768 * B)
        status = ctypes.create_string_buffer(B)
        check(self.ctx.L.plonk_prover_download(self._h, B, out, status))
        return out.raw, status.raw[:B]

    def download_compressed(self, B=None):
        """The resident batch's proofs as 480-byte records (Proof.to_bytes' form: nine compressed commitments, six
        big-endian evaluations), packed on the device, + the status bytes of download_raw."""
        B = self._resident if B is None else B
        out = ctypes.create_string_buffer(480 * B)
        status = ctypes.create_string_buffer(B)
        check(self.ctx.L.plonk_prover_download_compressed(self._h, B, out, status))
        return out.raw, status.raw[:B]

    def download(self, B=None):
        raw, status = self.download_raw(B)
        proofs = []
        for b, st in enumerate(status):
            if st & 8:
                raise ProofError("proof %d: an uploaded witness value is not a canonical Fr value (>= r)" % b)
            if st & 4:
                raise ProofError("proof %d: witness does not satisfy the gate constraints "
                                 "(prover.py:108-116, checked row by row; it is what the quotient-degree assert of prover.py:205-208 detects)" % b)
            if st & 2:
                raise ProofError("proof %d: permutation accumulator does not close to 1 (prover.py:132)" % b)
            if st & 1:
                raise ProofError("proof %d: a commitment is the identity; the reference's transcript "
                                 "cannot absorb it (transcript.py:65-67)" % b)
            proofs.append(self.decode(raw[768 * b : 768 * (b + 1)]))
        return proofs

    @staticmethod
    def decode(blob):
        def pt(i):
            o = 64 * i
            return (Fq(int.from_bytes(blob[o : o + 32], "little")), Fq(int.from_bytes(blob[o + 32 : o + 64], "little")))

        def sc(i):
            o = 576 + 32 * i
            return Scalar(int.from_bytes(blob[o : o + 32], "little"))

        return Proof(
            Message1(pt(0), pt(1), pt(2)),
            Message2(pt(3)),
            Message3(pt(4), pt(5), pt(6)),
            Message4(sc(0), sc(1), sc(2), sc(3), sc(4), sc(5)),
            Message5(pt(7), pt(8)),
        )

    def prove_batch(self, witnesses):
        self.upload(witnesses)
        self.run()
        return self.download()

    def prove(self, witness) -> Proof:  # prover.py:51-84
        return self.prove_batch([witness])[0]

    def challenges(self, b=0):
        out = ctypes.create_string_buffer(192)
        check(self.ctx.L.plonk_prover_challenges(self._h, b, out))
        names = ("beta", "gamma", "alpha", "fft_cofactor", "zeta", "v")
        return {k: Scalar(int.from_bytes(out.raw[32 * i : 32 * i + 32], "little")) for i, k in enumerate(names)}
