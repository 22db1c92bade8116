"""PLONK verification (optimised form) on the CPU.  (oracle: test infrastructure only)

Restates /root/reference/TESTING_verifier_DO_NOT_OPEN.py:39-163 (`verify_proof`: challenges, Z_H(zeta),
L_0(zeta), PI(zeta), r0, D, F, E and the single pairing check) and `compute_challenges` (:266-277) over
the oracle's own field / curve / transcript / pairing code.  Its purpose is an INDEPENDENT acceptance test
for proofs the GPU produces at sizes where no reference proof exists (group_order 2^10, 2^11): the proof
must verify against a verification key built from the circuit, not merely equal another prover's output.
The golden proof test/proof.pickle (K6) must verify, tampered proofs must not.
"""
from . import g1, pairing
from .field import R_MOD, inv, root_of_unity
from .fr_poly import Basis, Polynomial
from .strobe_merlin import Transcript


class VerificationKey:
    """verifier.py:9-34 — commitments to the circuit polynomials, [x]_2 and omega."""

    def __init__(self, group_order, Qm, Ql, Qr, Qo, Qc, S1, S2, S3, X_2, w):
        self.group_order = group_order
        self.Qm, self.Ql, self.Qr, self.Qo, self.Qc = Qm, Ql, Qr, Qo, Qc
        self.S1, self.S2, self.S3 = S1, S2, S3
        self.X_2, self.w = X_2, w

    @classmethod
    def from_setup(cls, setup, pk):
        """Setup.verification_key (setup.py:75-77, contract pinned by K3-K5)."""
        c = [setup.commit(p) for p in (pk.QM, pk.QL, pk.QR, pk.QO, pk.QC, pk.S1, pk.S2, pk.S3)]
        x2 = (pairing.FQ2(list(setup.X2[0])), pairing.FQ2(list(setup.X2[1])))
        return cls(pk.group_order, *c, x2, root_of_unity(pk.group_order))

    def compute_challenges(self, proof):  # TESTING_verifier:266-277
        t = Transcript(b"plonk")
        beta, gamma = t.round_1(proof["a_1"], proof["b_1"], proof["c_1"])
        alpha, _cof = t.round_2(proof["z_1"])
        zeta = t.round_3(proof["t_lo_1"], proof["t_mid_1"], proof["t_hi_1"])
        v = t.round_4(*[proof[k] for k in ("a_eval", "b_eval", "c_eval", "s1_eval", "s2_eval", "z_shifted_eval")])
        u = t.round_5(proof["W_z_1"], proof["W_zw_1"])
        return beta, gamma, alpha, zeta, v, u

    def verify_proof(self, group_order, proof, public=()) -> bool:
        """`proof` is the Proof.flatten() dict (G1 = affine int tuples, Fr = ints).  TESTING_verifier:39-163."""
        n = group_order
        beta, gamma, alpha, zeta, v, u = self.compute_challenges(proof)
        a, b, c = proof["a_eval"], proof["b_eval"], proof["c_eval"]
        s1, s2, zw = proof["s1_eval"], proof["s2_eval"], proof["z_shifted_eval"]
        w = root_of_unity(n)
        ZH_ev = (pow(zeta, n, R_MOD) - 1) % R_MOD
        L0_ev = ZH_ev * inv(n * (zeta - 1)) % R_MOD
        PI = Polynomial([-x for x in public] + [0] * (n - len(public)), Basis.LAGRANGE)
        PI_ev = PI.barycentric_eval(zeta)
        r0 = (PI_ev - L0_ev * alpha * alpha
              - alpha * (a + beta * s1 + gamma) * (b + beta * s2 + gamma) * (c + gamma) * zw) % R_MOD
        zn = pow(zeta, n, R_MOD)
        D_pt = g1.ec_lincomb([
            (self.Qm, a * b), (self.Ql, a), (self.Qr, b), (self.Qo, c), (self.Qc, 1),
            (proof["z_1"], (a + beta * zeta + gamma) * (b + beta * 2 * zeta + gamma) * (c + beta * 3 * zeta + gamma) * alpha
             + L0_ev * alpha * alpha + u),
            (self.S3, -(a + beta * s1 + gamma) * (b + beta * s2 + gamma) * alpha * beta * zw),
            (proof["t_lo_1"], -ZH_ev), (proof["t_mid_1"], -ZH_ev * zn), (proof["t_hi_1"], -ZH_ev * zn * zn),
        ])
        F_pt = g1.ec_lincomb([(D_pt, 1), (proof["a_1"], v), (proof["b_1"], v ** 2), (proof["c_1"], v ** 3),
                              (self.S1, v ** 4), (self.S2, v ** 5)])
        E_pt = g1.ec_mul(g1.G1, -r0 + v * a + v ** 2 * b + v ** 3 * c + v ** 4 * s1 + v ** 5 * s2 + u * zw)
        lhs = pairing.pairing(self.X_2, g1.ec_lincomb([(proof["W_z_1"], 1), (proof["W_zw_1"], u)]))
        rhs = pairing.pairing(pairing.G2, g1.ec_lincomb([
            (proof["W_z_1"], zeta), (proof["W_zw_1"], u * zeta * w), (F_pt, 1), (E_pt, -1)]))
        return lhs == rhs
