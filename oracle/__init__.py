"""oracle/ — CPU restatement of plonkathon's prover hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this
package, and only as the checker.  Nothing under plonkathon_amd/ imports it; the product
path is HIP-only and fails loudly when libplonk_hip.so is missing.

Parity status: PINNED.  Every function here is checked (tests/test_oracle_golden.py) against
the reference's own known-answer data: test.py:18-33 (K1 commit KAT, K2 omega_8), the three
zkrepl/snarkjs verification keys test/main.plonk.vkey{,-58,-59}.json (K3-K5), the golden proof
test/proof.pickle (K6, all 9 G1 points + 6 Fr evaluations, which also pins the un-vendored
Merlin transcript and py_ecc field/curve semantics), poseidon_hash(1,2) (K7), and against
vectors produced by importing the reference's own poly.py / curve.py / compiler / setup.py /
transcript.py in the build container (tools/gen_golden.py -> tests/golden/*.json).

One exception, stated where it lives (oracle/c/bn254_oracle.c, oracle_bls_fr_ntt): the BLS12-381 scalar-field transform has NO
reference counterpart (the reference is BN254 throughout) — parity unpinned against the reference, pinned by definition only
(DFT sum in Python integers, the bls12_381 crate's published root of unity, the committed vectors of tools/gen_bls_vectors.py;
tests/test_oracle_c.py).

Third-party arithmetic that is NOT under /root/reference and is restated from its published
algorithm: py-ecc 6.0.0 (bn128 FQ + affine G1 add/double/multiply; pyproject.toml:11,
poetry.lock:362) and merlin @805d0678 (Merlin v1.0 over STROBE-128 / Keccak-f[1600];
pyproject.toml:12, poetry.lock:255-269).

All elements are plain Python ints in canonical form (0 <= x < modulus); G1 points are affine
(x, y) int tuples with None as the identity (py_ecc convention, utils.py:13-14).
"""
