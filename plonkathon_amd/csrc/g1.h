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

// acc += affine q   (madd-2008-s)
PLONK_HD void g1_madd(G1Xyzz& p, const G1Affine& q) {
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
        if (fp_is_zero(r)) p = g1_dbl_affine(q);   // same point
        else p = g1_xyzz_identity();               // opposite points
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
// Lazy-limb accumulator for the MSM inner loop (fpl.h): same madd-2008-s formulas, no canonical
// reductions.  Invariants between calls (all limbs normalised): x < 8m, y < 4m, zz < 2m, zzz < 2m.
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
// reduction).  No calls, no packed arithmetic: minimal live registers.
PLONK_HD bool g1l_madd_fast(G1XyzzL& p, const Fq& x2p, const Fq& y2p) {
    if (fp_is_zero(x2p) && fp_is_zero(y2p)) return false;
    const FqL x2 = fpl_from_fp(x2p), y2 = fpl_from_fp(y2p);
    if (p.inf) {
        p.x = x2;
        p.y = y2;
        p.zz = fpl_one<FqParams>();
        p.zzz = p.zz;
        p.inf = false;
        return true;
    }
    const FqL u2 = fpl_mul(x2, p.zz);                                  // < 2m
    const FqL pp_ = fpl_norm(fpl_sub<FqParams, 8>(u2, p.x));           // U2 - X1 + 8m   in (0, 10m)
    {   // cheap filter for P == 0 (mod m): limb 0 must match limb 0 of some j*m, j < 16
        bool maybe = false;
#pragma unroll
        for (unsigned j = 0; j < 16; j++)
            maybe |= pp_.l[0] == (uint32_t)(((uint64_t)j * fp29_mod_limb<FqParams>(0)) & FP29_MASK);
        if (maybe) return false;  // (2^-25 false-positive rate: the general path copes)
    }
    const FqL s2 = fpl_mul(y2, p.zzz);                                 // < 2m
    const FqL rr = fpl_norm(fpl_sub<FqParams, 4>(s2, p.y));            // S2 - Y1 + 4m   in (0, 6m)
    const FqL pp = fpl_sqr(pp_);                                       // < 2m
    const FqL q = fpl_mul(p.x, pp);                                    // < 2m   (X1 dead after this)
    p.zz = fpl_mul(p.zz, pp);
    const FqL ppp = fpl_mul(pp_, pp);                                  // < 2m   (P, PP dead after this)
    p.zzz = fpl_mul(p.zzz, ppp);
    const FqL r2 = fpl_sqr(rr);                                        // < 2m
    // X3 = R^2 - PPP - 2Q  ->  R^2 + (2m - PPP) + (4m - 2Q)  in (0, 8m)
    p.x = fpl_norm(fpl_sub<FqParams, 4>(fpl_sub<FqParams, 2>(r2, ppp), fpl_add(q, q)));
    const FqL d = fpl_norm(fpl_sub<FqParams, 8>(q, p.x));              // Q - X3 + 8m    in (0, 10m)
    FqL zero;
#pragma unroll
    for (int i = 0; i < 9; i++) zero.l[i] = 0;
    const FqL ny1 = fpl_norm(fpl_sub<FqParams, 4>(zero, p.y));         // 4m - Y1        in (0, 4m]
    // Y3 = R (Q - X3) - Y1 PPP as one sum of products with a single reduction: 6*10 + 4*2 <= 128
    p.y = fpl_mul_add(rr, d, ny1, ppp);                                // < 2m
    return true;
}

// Piece form of a lazy accumulator for msm_accumulate_kernel: four 256-bit words, x and y < 4m, zz and zzz
// < 2m (not canonical; g1_piece_load canonicalises), identity = all zero.  Costs ~100 instructions, no
// multiplication, so flushing at a bucket boundary stays cheap.
PLONK_HD void fpl_pack_lt4m(const FqL& a, bool sub4m, uint32_t out[8]) {
    uint32_t w[9];
    fp29_pack(a.l, w);                      // limbs 0..8 -> words 0..7 (bits < 256) ...
    w[8] = a.l[8] >> 24;                    // ... bit 256 and up (limb 8 starts at bit 232)
    if (sub4m) {                            // value < 8m: subtract 4m when that does not go negative
        uint32_t t[9], borrow = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            // 4m as 9 words
            const uint32_t mi = i < 8 ? ((FqParams::mod(i) << 2) | (i ? FqParams::mod(i - 1) >> 30 : 0)) : (FqParams::mod(7) >> 30);
            t[i] = fp_sbb(w[i], mi, borrow);
        }
        if (!borrow) {
#pragma unroll
            for (int i = 0; i < 9; i++) w[i] = t[i];
        }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = w[i];
}

PLONK_HD G1Xyzz g1l_to_piece(const G1XyzzL& p) {
    if (p.inf) return g1_xyzz_identity();
    G1Xyzz r;
    fpl_pack_lt4m(p.x, true, r.x.v);
    fpl_pack_lt4m(p.y, false, r.y.v);
    fpl_pack_lt4m(p.zz, false, r.zz.v);
    fpl_pack_lt4m(p.zzz, false, r.zzz.v);
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
    }
    fp_reduce_once<FqParams>(r.zz.v);
    fp_reduce_once<FqParams>(r.zzz.v);
    return r;
}
