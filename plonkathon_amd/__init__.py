"""plonkathon_amd — MI355X (gfx950) backend for the prover hot path of 0xPARC/plonkathon.

Python host layer with the reference's own API surface (`Scalar`, `Polynomial`, `Basis`, `Setup`,
`Program`, `Prover`, `Proof`, `Transcript`, `ec_lincomb`), every vector operation executed by
hand-written HIP kernels in libplonk_hip.so through the C-ABI of include/plonk_hip.h.
There is no CPU fallback: without the library and a GPU the compute entry points raise.
"""
from .field import Fq, Scalar, primitive_root  # noqa: F401
from .polynomial import Basis, Polynomial  # noqa: F401
from .kzg import G1, G2, Z1, Setup, VerificationKey, ec_lincomb, ec_mul, g1_compress, g1_decompress, lincomb, multisubset, pairing_check  # noqa: F401
from .circuit import AssemblyEqn, Cell, Column, CommonPreprocessedInput, GateWires, Program  # noqa: F401
from .fiat_shamir import Message1, Message2, Message3, Message4, Message5, Transcript  # noqa: F401
from .plonk import Proof, Prover  # noqa: F401
from .batch import BatchProver, ProofError  # noqa: F401
from .backend import Context, get_context, set_context  # noqa: F401

__all__ = [
    "Scalar", "Fq", "Basis", "Polynomial", "Setup", "VerificationKey", "ec_lincomb", "ec_mul", "lincomb", "multisubset", "g1_compress", "g1_decompress", "pairing_check", "G1", "G2", "Z1",
    "Program", "CommonPreprocessedInput", "AssemblyEqn", "GateWires", "Column", "Cell", "Transcript", "Message1", "Message2",
    "Message3", "Message4", "Message5", "Prover", "BatchProver", "ProofError", "Proof", "Context", "get_context", "set_context",
]
