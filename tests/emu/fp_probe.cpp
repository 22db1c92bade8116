// Host-side probe of the device field/curve primitives (fp.h / g1.h compiled with g++ via PLONK_EMU).
// TEST INFRASTRUCTURE ONLY: lets tests/test_emu_primitives.py compare the primitives with Python ints.
#include "fp.h"
#include "g1.h"
#include "bls12_381_constants.h"

template <class P> static void binop(int op, const uint32_t* a, const uint32_t* b, uint32_t* out) {
    Fp<P> x, y, r;
    memcpy(x.v, a, 32);
    memcpy(y.v, b, 32);
    switch (op) {
        case 0: r = fp_add(x, y); break;
        case 1: r = fp_sub(x, y); break;
        case 2: r = fp_mul(x, y); break;
        case 3: r = fp_inv(x); break;
        case 4: r = fp_to_mont(x); break;
        case 5: r = fp_from_mont(x); break;
        case 6: r = fp_neg(x); break;
        case 7: r = fp_sqr(x); break;
        case 8: r = fp_inv_fermat(x); break;
        default: r = fp_zero<P>();
    }
    memcpy(out, r.v, 32);
}
extern "C" void probe_fr(int op, const uint32_t* a, const uint32_t* b, uint32_t* out) { binop<FrParams>(op, a, b, out); }
extern "C" void probe_fq(int op, const uint32_t* a, const uint32_t* b, uint32_t* out) { binop<FqParams>(op, a, b, out); }
extern "C" void probe_bls_fr(int op, const uint32_t* a, const uint32_t* b, uint32_t* out) { binop<BlsFrParams>(op, a, b, out); }

// points: affine in = 16 words (x,y Montgomery; (0,0) = identity); xyzz = 32 words
extern "C" void probe_g1(int op, const uint32_t* p, const uint32_t* q, uint32_t* out) {
    G1Xyzz acc;
    G1Affine a, b2;
    memcpy(&a, p, 64);
    memcpy(&b2, q, 64);
    switch (op) {
        case 0:  // affine + affine through xyzz mixed add
            acc = g1_xyzz_from_affine(a);
            g1_madd(acc, b2);
            break;
        case 1:  // xyzz(a) + xyzz(b) full add
            acc = g1_xyzz_from_affine(a);
            g1_add(acc, g1_xyzz_from_affine(b2));
            break;
        case 2:  // double
            acc = g1_xyzz_from_affine(a);
            g1_dbl(acc);
            break;
        case 3: {  // (a+a) + b with non-trivial ZZ on the accumulator
            acc = g1_xyzz_from_affine(a);
            g1_dbl(acc);
            g1_madd(acc, b2);
            break;
        }
        case 4: {  // (2a) + (2b) full add with both ZZ non-trivial
            acc = g1_xyzz_from_affine(a);
            g1_dbl(acc);
            G1Xyzz o = g1_xyzz_from_affine(b2);
            g1_dbl(o);
            g1_add(acc, o);
            break;
        }
        default: acc = g1_xyzz_identity();
    }
    G1Affine r = g1_to_affine(acc);
    memcpy(out, &r, 64);
}

// acc = sum_i (+/-) pts[i] through the LAZY accumulator of g1.h / fpl.h; affine Montgomery coords out
extern "C" void probe_g1l_chain(const uint32_t* pts, const int* neg, int n, uint32_t* out) {
    G1XyzzL acc = g1l_identity();
    for (int i = 0; i < n; i++) {
        G1Affine a;
        memcpy(&a, pts + 16 * i, 64);
        if (!g1l_madd_fast(acc, a.x, a.y, neg[i] != 0)) {  // exceptional step: the general packed formulas
            if (neg[i]) a.y = fp_neg(a.y);
            G1Xyzz t = g1l_to_xyzz(acc);
            g1_madd(t, a);
            acc = g1l_from_xyzz(t);
        }
    }
    // through the stored "piece" form and back, as msm_accumulate_kernel / msm_bucket_reduce_kernel do
    const G1Xyzz piece = g1l_to_piece(acc);
    G1Affine r = g1_to_affine(g1_piece_load(&piece));
    G1Affine direct = g1_to_affine(g1l_to_xyzz(acc));
    if (memcmp(&r, &direct, 64) != 0) memset(&r, 0xff, 64);  // the two routes must agree
    memcpy(out, &r, 64);
}

// raw signed-limb probe of fpl.h: op 0 = mul(a, b), 1 = sqr(a), 2 = mul_add(a, b, c, d), 3 = norm(a), 4 = to_fp(a)
// (8 packed words in out[0..7]), 5 = pack_lt4m<8>(a) (8 words), 6 = pack_lt4m<1>(a) (8 words); limbs as int32[9]
extern "C" void probe_fpl(int op, const int32_t* a, const int32_t* b, const int32_t* c, const int32_t* d, int32_t* out) {
    FqL x, y, z, w, r = fpl_zero<FqParams>();
    memcpy(x.l, a, 36);
    memcpy(y.l, b, 36);
    memcpy(z.l, c, 36);
    memcpy(w.l, d, 36);
    switch (op) {
        case 0: r = fpl_mul(x, y); break;
        case 1: r = fpl_sqr(x); break;
        case 2: r = fpl_mul_add(x, y, z, w); break;
        case 3: r = fpl_norm(x); break;
        case 4: { Fq f = fpl_to_fp(x); memcpy(r.l, f.v, 32); break; }
        case 5: fpl_pack_lt4m<8>(x, (uint32_t*)r.l); break;
        case 6: fpl_pack_lt4m<1>(x, (uint32_t*)r.l); break;
        default: break;
    }
    memcpy(out, r.l, 36);
}

// fpl_mul_shoup: a = 9 signed limbs, wt = a constant in canonical Montgomery form (8 packed words);
// out[0..8] = a * w (limbs), out[9..17] = w, out[18..26] = wp = floor(w 2^261 / m)
template <class P> static void shoup_probe(const int32_t* a, const uint32_t* wt, int32_t* out) {
    FpL<P> x;
    memcpy(x.l, a, 36);
    Fp<P> c;
    memcpy(c.v, wt, 32);
    uint32_t ninv[9];
    fpl_ninv261<P>(ninv);
    const FpLS<P> s = fpl_shoup_from_mont(c, ninv);
    const FpL<P> r = fpl_mul_shoup(x, s);
    memcpy(out, r.l, 36);
    memcpy(out + 9, s.w, 36);
    memcpy(out + 18, s.wp, 36);
}
extern "C" void probe_fpl_shoup(const int32_t* a, const uint32_t* wt, int32_t* out) { shoup_probe<FrParams>(a, wt, out); }
extern "C" void probe_fpl_shoup_bls(const int32_t* a, const uint32_t* wt, int32_t* out) { shoup_probe<BlsFrParams>(a, wt, out); }
