// g1.h — BN254 G1 (y^2 = x^3 + 3 over Fq) group law for the MSM kernels.
//
// Replaces py_ecc.bn128 `add` / `double` / `multiply` as called from the reference's
// `ec_lincomb` (/root/reference/curve.py:38-44).  py_ecc works on affine points with one field
// inversion per addition; the group law is canonical, so any coordinate system gives the same
// group element, and parity is defined on the unique affine representative produced at the very
// end (g1_to_affine) — SURVEY.md Appendix B.
//
// Coordinates: bases are affine (x, y), 64 B, the .ptau layout (setup.py:29-41); the identity is
// encoded as (0, 0), which is not on the curve (b = 3).  Accumulators are extended Jacobian
// "XYZZ" (X, Y, ZZ, ZZZ) with x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; identity <=> ZZ == 0.
// Formulas: EFD short-Weierstrass xyzz add-2008-s (12M+2S), madd-2008-s (8M+2S), dbl-2008-s-1
// (6M+4S for a = 0 ... counted as implemented below).  All exceptional cases (identity operands,
// P == Q, P == -Q) are handled explicitly so `ec_lincomb` is correct for duplicate and cancelling
// inputs as well as for the SRS.
#pragma once
#include "fp.h"

struct alignas(16) G1Affine { Fq x, y; };
struct alignas(16) G1Xyzz { Fq x, y, zz, zzz; };

PLONK_HD bool g1_affine_is_identity(const G1Affine& p) { return fp_is_zero(p.x) && fp_is_zero(p.y); }
PLONK_HD bool g1_is_identity(const G1Xyzz& p) { return fp_is_zero(p.zz); }

PLONK_HD G1Xyzz g1_xyzz_identity() {
    G1Xyzz r;
    r.x = fp_zero<FqParams>(); r.y = fp_zero<FqParams>(); r.zz = fp_zero<FqParams>(); r.zzz = fp_zero<FqParams>();
    return r;
}
PLONK_HD G1Affine g1_affine_identity() {
    G1Affine r;
    r.x = fp_zero<FqParams>(); r.y = fp_zero<FqParams>();
    return r;
}
PLONK_HD G1Xyzz g1_xyzz_from_affine(const G1Affine& p) {
    if (g1_affine_is_identity(p)) return g1_xyzz_identity();
    G1Xyzz r;
    r.x = p.x; r.y = p.y; r.zz = fp_one<FqParams>(); r.zzz = fp_one<FqParams>();
    return r;
}
PLONK_HD G1Affine g1_affine_neg(const G1Affine& p) {
    G1Affine r;
    r.x = p.x; r.y = fp_neg(p.y);
    return r;
}

// acc = 2*acc   (dbl-2008-s-1, a = 0)
PLONK_HD void g1_dbl(G1Xyzz& p) {
    if (g1_is_identity(p)) return;
    Fq u = fp_dbl(p.y);
    Fq v = fp_sqr(u);
    Fq w = fp_mul(u, v);
    Fq s = fp_mul(p.x, v);
    Fq m = fp_mul3(fp_sqr(p.x));
    Fq x3 = fp_sub(fp_sqr(m), fp_dbl(s));
    Fq y3 = fp_sub(fp_mul(m, fp_sub(s, x3)), fp_mul(w, p.y));
    p.zz = fp_mul(v, p.zz);
    p.zzz = fp_mul(w, p.zzz);
    p.x = x3;
    p.y = y3;
}

// acc = 2*(affine q)   (mdbl-2008-s-1)
PLONK_HD G1Xyzz g1_dbl_affine(const G1Affine& q) {
    G1Xyzz r;
    Fq u = fp_dbl(q.y);
    r.zz = fp_sqr(u);
    r.zzz = fp_mul(u, r.zz);
    Fq s = fp_mul(q.x, r.zz);
    Fq m = fp_mul3(fp_sqr(q.x));
    r.x = fp_sub(fp_sqr(m), fp_dbl(s));
    r.y = fp_sub(fp_mul(m, fp_sub(s, r.x)), fp_mul(r.zzz, q.y));
    return r;
}

// the rare exits of g1_madd: acc == q (double it) or acc == -q (identity)
PLONK_HD_NOINLINE G1Xyzz g1_madd_equal_x(const G1Affine q, bool same) { return same ? g1_dbl_affine(q) : g1_xyzz_identity(); }

// acc += affine q   (madd-2008-s).  OUTLINE_RARE: the acc == +-q exit as a call — inlined, that branch costs a kernel whose
// loop body is this addition 120 .. 1 000 spilled registers (msm_lookup_fill_kernel: 2 200 B of scratch per lane in round 3);
// as a call it costs a 144-byte frame that only the rare exit touches.  The kernels that had no scratch keep the inline form.
template <bool OUTLINE_RARE = false> PLONK_HD void g1_madd(G1Xyzz& p, const G1Affine& q) {
    if (g1_affine_is_identity(q)) return;
    if (g1_is_identity(p)) {
        p.x = q.x; p.y = q.y; p.zz = fp_one<FqParams>(); p.zzz = fp_one<FqParams>();
        return;
    }
    Fq u2 = fp_mul(q.x, p.zz);
    Fq s2 = fp_mul(q.y, p.zzz);
    Fq pp_ = fp_sub(u2, p.x);
    Fq r = fp_sub(s2, p.y);
    if (fp_is_zero(pp_)) {
        if constexpr (OUTLINE_RARE) p = g1_madd_equal_x(q, fp_is_zero(r));
        else if (fp_is_zero(r)) p = g1_dbl_affine(q);   // same point
        else p = g1_xyzz_identity();                    // opposite points
        return;
    }
    Fq pp = fp_sqr(pp_);
    Fq ppp = fp_mul(pp_, pp);
    Fq qq = fp_mul(p.x, pp);
    Fq x3 = fp_sub(fp_sub(fp_sqr(r), ppp), fp_dbl(qq));
    Fq y3 = fp_sub(fp_mul(r, fp_sub(qq, x3)), fp_mul(p.y, ppp));
    p.zz = fp_mul(p.zz, pp);
    p.zzz = fp_mul(p.zzz, ppp);
    p.x = x3;
    p.y = y3;
}

// acc += xyzz q   (add-2008-s)
PLONK_HD void g1_add(G1Xyzz& p, const G1Xyzz& q) {
    if (g1_is_identity(q)) return;
    if (g1_is_identity(p)) { p = q; return; }
    Fq u1 = fp_mul(p.x, q.zz);
    Fq u2 = fp_mul(q.x, p.zz);
    Fq s1 = fp_mul(p.y, q.zzz);
    Fq s2 = fp_mul(q.y, p.zzz);
    Fq pp_ = fp_sub(u2, u1);
    Fq r = fp_sub(s2, s1);
    if (fp_is_zero(pp_)) {
        if (fp_is_zero(r)) g1_dbl(p);
        else p = g1_xyzz_identity();
        return;
    }
    Fq pp = fp_sqr(pp_);
    Fq ppp = fp_mul(pp_, pp);
    Fq qq = fp_mul(u1, pp);
    Fq x3 = fp_sub(fp_sub(fp_sqr(r), ppp), fp_dbl(qq));
    Fq y3 = fp_sub(fp_mul(r, fp_sub(qq, x3)), fp_mul(s1, ppp));
    p.zz = fp_mul(fp_mul(p.zz, q.zz), pp);
    p.zzz = fp_mul(fp_mul(p.zzz, q.zzz), ppp);
    p.x = x3;
    p.y = y3;
}

// unique affine representative (Montgomery-form coordinates); identity -> (0, 0)
PLONK_HD G1Affine g1_to_affine(const G1Xyzz& p) {
    if (g1_is_identity(p)) return g1_affine_identity();
    Fq t = fp_inv(fp_mul(p.zz, p.zzz));
    Fq izz = fp_mul(t, p.zzz);
    Fq izzz = fp_mul(t, p.zz);
    G1Affine r;
    r.x = fp_mul(p.x, izz);
    r.y = fp_mul(p.y, izzz);
    return r;
}

// ------------------------------------------------------------------------------------------------
// Lazy-limb accumulator for the MSM inner loop (fpl.h): same madd-2008-s formulas on signed limbs, no canonical
// reductions, one carry sweep per addition.  Invariants between calls (all four normalised): x in (-7m, 5m),
// y, zz, zzz in (-m, 2m).
#include "fpl.h"

typedef FpL<FqParams> FqL;
struct G1XyzzL {
    FqL x, y, zz, zzz;
    bool inf;
};

PLONK_HD G1XyzzL g1l_identity() {
    G1XyzzL r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.x.l[i] = r.y.l[i] = r.zz.l[i] = r.zzz.l[i] = 0;
    r.inf = true;
    return r;
}

PLONK_HD G1Xyzz g1l_to_xyzz(const G1XyzzL& p) {
    if (p.inf) return g1_xyzz_identity();
    G1Xyzz r;
    r.x = fpl_to_fp(p.x);
    r.y = fpl_to_fp(p.y);
    r.zz = fpl_to_fp(p.zz);
    r.zzz = fpl_to_fp(p.zzz);
    return r;
}

PLONK_HD G1XyzzL g1l_from_xyzz(const G1Xyzz& p) {
    G1XyzzL r;
    r.inf = g1_is_identity(p);
    r.x = fpl_from_fp(p.x);
    r.y = fpl_from_fp(p.y);
    r.zz = fpl_from_fp(p.zz);
    r.zzz = fpl_from_fp(p.zzz);
    return r;
}

// acc += (x2, y2), canonical Montgomery coordinates.  Returns false WITHOUT touching acc when the step is
// exceptional — identity base (0, 0), or P == +-Q (detected by a cheap filter with a 2^-25 false-positive
// rate) — and the caller resolves it with the general packed formulas (msm.hip defers it to the bucket
// reduction).  No calls, no packed arithmetic: minimal live registers.  neg_y adds (x2, -y2) instead: on signed limbs
// the negation is 18 bit operations on y2, against ~40 instructions for a packed m - y and its select.
//
// Bounds (|value| / m, see fpl.h): U2, S2, PP, Q, PPP, R^2, ZZ3, ZZZ3, Y3 are products, in (-1, 2); P = U2 - X1 in
// (-6, 9); R = S2 - Y1 in (-3, 3); X3 = R^2 - PPP - 2Q in (-7, 5); D = Q - X3 in (-6, 9).  Products: P^2 <= 81,
// X1 PP <= 14, P PP <= 18, R^2 <= 9, R D + Y1 PPP <= 27 + 4 — all below the 128 of fpl_mul.  Limbs: every
// multiplicand is a normalised value or the difference of two, within (-2^29, 2^29).
PLONK_HD bool g1l_madd_fast(G1XyzzL& p, const Fq& x2p, const Fq& y2p, bool neg_y = false) {
    if (fp_is_zero(x2p) && fp_is_zero(y2p)) return false;
    const FqL x2 = fpl_from_fp(x2p), y2 = fpl_cneg(fpl_from_fp(y2p), neg_y);   // y2 in (-m, m)
    if (p.inf) {
        p.x = x2;
        p.y = fpl_norm(y2);  // a negated y2 has limbs in (-2^29, 0]: the invariant wants them normalised (R = S2 - Y1 is squared)
        p.zz = fpl_one<FqParams>();
        p.zzz = p.zz;
        p.inf = false;
        return true;
    }
    const FqL u2 = fpl_mul(x2, p.zz);
    const FqL pp_ = fpl_sub(u2, p.x);                                  // P = U2 - X1
    if (fpl_maybe_zero_mod<FqParams, -5, 8>(pp_)) return false;        // (2^-25 false-positive rate: the general path copes)
    const FqL s2 = fpl_mul(y2, p.zzz);
    const FqL rr = fpl_sub(s2, p.y);                                   // R = S2 - Y1
    const FqL pp = fpl_sqr(pp_);
    const FqL q = fpl_mul(p.x, pp);                                    // (X1 dead after this)
    p.zz = fpl_mul(p.zz, pp);
    const FqL ppp = fpl_mul(pp_, pp);                                  // (P, PP dead after this)
    p.zzz = fpl_mul(p.zzz, ppp);
    const FqL r2 = fpl_sqr(rr);
    p.x = fpl_norm(fpl_sub(fpl_sub(r2, ppp), fpl_add(q, q)));          // X3 = R^2 - PPP - 2Q: the one carry sweep
    const FqL d = fpl_sub(q, p.x);                                     // Q - X3
    // Y3 = R (Q - X3) - Y1 PPP as one sum of products with a single reduction
    p.y = fpl_mul_add(rr, d, fpl_neg(p.y), ppp);
    return true;
}

// acc += q, both on lazy limbs (add-2008-s: 14 products under 13 reductions, ~2 700 instructions against ~4 600 for the
// packed g1_add) — the tree reductions and the Horner chains of the table MSMs (msm_comb.h), where a general addition is the
// step every other lane or wave waits for.  Operands: normalised limbs, |x| <= 7 m, |y|, |zz|, |zzz| <= 4 m — the invariants
// of g1l_madd_fast's accumulator as well as a "piece" in [0, 4m) unpacked as it is (g1l_from_piece).  Result within the
// accumulator invariants: x in (-7m, 5m), y, zz, zzz in (-m, 2m).  Returns false WITHOUT touching p when the operands are
// equal or opposite (an exact, call-free test: fpl_is_zero_mod_in): the caller takes the packed formulas (g1l_add below) or,
// where a call would cost the kernel its registers, hands the whole MSM to the general-formula kernel.
// Bounds (|value| / m): U1, U2 <= 7 * 4, S1, S2, ZZ1 ZZ2, ZZZ1 ZZZ2 <= 16; P, R in (-3, 3); P^2 <= 9, P PP <= 6, U1 PP <= 4,
// (ZZ1 ZZ2) PP <= 4; X3 = R^2 - PPP - 2Q in (-7, 5); D = Q - X3 in (-6, 9); R D + S1 PPP <= 27 + 4 — all below 128.
// OPPOSITE_OK: p == -q is resolved here (the sum is the identity: one more exact test, reached only on equal x) and only p == q
// still returns false.  For the last steps of an MSM whose true result is the identity — every scalar zero (a zero selector
// polynomial), or cancelling terms: the comb recoding turns a zero scalar into r, so the cancellation happens in the final
// Horner / butterfly addition and would otherwise send the whole MSM to the general-formula kernel.
template <bool OPPOSITE_OK = false> PLONK_HD bool g1l_add_fast(G1XyzzL& p, const G1XyzzL& q) {
    if (q.inf) return true;
    if (p.inf) {
        p = q;
        return true;
    }
    const FqL u1 = fpl_mul(p.x, q.zz), u2 = fpl_mul(q.x, p.zz);
    const FqL pp_ = fpl_sub(u2, u1);
    if (fpl_is_zero_mod_in<FqParams, -3, 3>(pp_)) {
        if constexpr (OPPOSITE_OK) {
            const FqL t1 = fpl_mul(p.y, q.zzz), t2 = fpl_mul(q.y, p.zzz);
            if (fpl_is_zero_mod_in<FqParams, -3, 3>(fpl_sub(t2, t1))) return false;  // equal points: a doubling, the caller's business
            p = g1l_identity();  // equal x, different y: opposite points
            return true;
        }
        return false;
    }
    const FqL s1 = fpl_mul(p.y, q.zzz), s2 = fpl_mul(q.y, p.zzz);
    const FqL rr = fpl_sub(s2, s1);
    const FqL pp = fpl_sqr(pp_);
    const FqL qq = fpl_mul(u1, pp);
    const FqL ppp = fpl_mul(pp_, pp);
    p.zz = fpl_mul(fpl_mul(p.zz, q.zz), pp);
    p.zzz = fpl_mul(fpl_mul(p.zzz, q.zzz), ppp);
    const FqL r2 = fpl_sqr(rr);
    p.x = fpl_norm(fpl_sub(fpl_sub(r2, ppp), fpl_add(qq, qq)));
    const FqL d = fpl_sub(qq, p.x);
    p.y = fpl_mul_add(rr, d, fpl_neg(s1), ppp);
    return true;
}

// acc = 2 acc on lazy limbs (dbl-2008-s-1, a = 0: 9 products, three of them squarings, under 8 reductions).  Operand and
// result ranges as g1l_add_fast.  Bounds: U = 2Y <= 8, U^2 <= 64, U V <= 16, X V <= 14, X^2 <= 49, M = 3 X^2 in (-3, 6),
// M^2 <= 36, X3 = M^2 - 2S in (-5, 4), M (S - X3) + W Y <= 6 * 7 + 2 * 4.  (No exceptional case: G1 has no point of order two.)
PLONK_HD void g1l_dbl(G1XyzzL& p) {
    if (p.inf) return;
    const FqL u = fpl_norm(fpl_add(p.y, p.y));
    const FqL v = fpl_sqr(u);
    const FqL w = fpl_mul(u, v);
    const FqL s = fpl_mul(p.x, v);
    const FqL xx = fpl_sqr(p.x);
    const FqL m = fpl_norm(fpl_add(fpl_add(xx, xx), xx));
    p.zz = fpl_mul(v, p.zz);
    p.zzz = fpl_mul(w, p.zzz);
    const FqL m2 = fpl_sqr(m);
    p.x = fpl_norm(fpl_sub(m2, fpl_add(s, s)));
    const FqL d = fpl_sub(s, p.x);
    p.y = fpl_mul_add(m, d, fpl_neg(w), p.y);
}

// Piece form of a lazy accumulator for msm_accumulate_kernel: four 256-bit words in [0, 4m) (not canonical;
// g1_piece_load canonicalises), identity = all zero.  Costs ~150 instructions, no multiplication, so flushing at a
// bucket boundary stays cheap.
//   K = the multiple of m added first to make the value positive; value + K m must stay below 13 m < 2^258.
template <unsigned K> PLONK_HD void fpl_pack_lt4m(const FqL& a, uint32_t out[8]) {
    const FqL t = fpl_add_km_norm<FqParams, K>(a);
    uint32_t u[9], w[9];
#pragma unroll
    for (int i = 0; i < 9; i++) u[i] = (uint32_t)t.l[i];
    fp29_pack(u, w);                        // limbs 0..8 -> words 0..7 (bits < 256) ...
    w[8] = u[8] >> 24;                      // ... bit 256 and up (limb 8 starts at bit 232)
#pragma unroll
    for (int sh = 3; sh >= 2; sh--) {       // subtract 8m, then 4m, when that does not go negative
        if ((K + 2) <= (1u << sh)) continue;   // value < (K + 2) m: nothing to take off
        uint32_t s[9], borrow = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const uint32_t mi = i < 8 ? ((FqParams::mod(i) << sh) | (i ? FqParams::mod(i - 1) >> (32 - sh) : 0))
                                      : (FqParams::mod(7) >> (32 - sh));
            s[i] = fp_sbb(w[i], mi, borrow);
        }
        if (!borrow) {
#pragma unroll
            for (int i = 0; i < 9; i++) w[i] = s[i];
        }
    }
    FPL_CHECK(w[8] == 0, "fpl_pack_lt4m: the piece does not fit 256 bits");
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = w[i];
}

PLONK_HD G1Xyzz g1l_to_piece(const G1XyzzL& p) {
    if (p.inf) return g1_xyzz_identity();
    G1Xyzz r;
    fpl_pack_lt4m<8>(p.x, r.x.v);           // (-7m, 5m) + 8m -> (m, 13m) -> [0, 4m)
    fpl_pack_lt4m<1>(p.y, r.y.v);           // (-m, 2m) + m -> (0, 3m)
    fpl_pack_lt4m<1>(p.zz, r.zz.v);
    fpl_pack_lt4m<1>(p.zzz, r.zzz.v);
    return r;
}

// The same for a sum that may itself be a piece taken over unchanged (g1l_add_fast with an identity accumulator copies its
// operand): x within (-8m, 5m), y, zz, zzz within (-3m, 5m) -> four words in [0, 4m).  Idempotent on pieces: unpacking and
// packing a piece any number of times stays inside [0, 4m) (g1l_to_piece would add m to y, zz, zzz every time).
PLONK_HD G1Xyzz g1l_to_piece_wide(const G1XyzzL& p) {
    if (p.inf) return g1_xyzz_identity();
    G1Xyzz r;
    fpl_pack_lt4m<8>(p.x, r.x.v);
    fpl_pack_lt4m<3>(p.y, r.y.v);           // (-3m, 5m) + 3m -> (0, 8m) -> [0, 4m)
    fpl_pack_lt4m<3>(p.zz, r.zz.v);
    fpl_pack_lt4m<3>(p.zzz, r.zzz.v);
    return r;
}

// canonical XYZZ from a stored piece (either form: canonical pieces pass through unchanged)
PLONK_HD G1Xyzz g1_piece_load(const G1Xyzz* src) {
    G1Xyzz r;
    r.x = fp_load(&src->x);
    r.y = fp_load(&src->y);
    r.zz = fp_load(&src->zz);
    r.zzz = fp_load(&src->zzz);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        fp_reduce_once<FqParams>(r.x.v);
        fp_reduce_once<FqParams>(r.y.v);
        fp_reduce_once<FqParams>(r.zz.v);
        fp_reduce_once<FqParams>(r.zzz.v);
    }
    return r;
}


// lazy limbs of a stored piece as it is (no canonicalisation: a piece in [0, 4m) is a valid operand of g1l_add_fast / g1l_dbl);
// the identity is the all-zero piece
PLONK_HD G1XyzzL g1l_from_piece(const G1Xyzz* src) {
    G1XyzzL r;
    const Fq x = fp_load(&src->x), y = fp_load(&src->y), zz = fp_load(&src->zz), zzz = fp_load(&src->zzz);
    r.inf = fp_is_zero(zz);
    r.x = fpl_from_fp(x);
    r.y = fpl_from_fp(y);
    r.zz = fpl_from_fp(zz);
    r.zzz = fpl_from_fp(zzz);
    return r;
}

// the exceptional exit of g1l_add: both operands through the packed general formulas (identity, P == +-Q)
PLONK_HD_NOINLINE G1XyzzL g1l_add_slow(const G1XyzzL p, const G1XyzzL q) {
    G1Xyzz a = g1l_to_xyzz(p);
    g1_add(a, g1l_to_xyzz(q));
    return g1l_from_xyzz(a);
}

// acc += q on lazy limbs, every case
PLONK_HD void g1l_add(G1XyzzL& p, const G1XyzzL& q) {
    if (!g1l_add_fast(p, q)) p = g1l_add_slow(p, q);
}
