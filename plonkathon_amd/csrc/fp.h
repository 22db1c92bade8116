// fp.h — 254-bit prime-field arithmetic in Montgomery form (R = 2^256), 8 x u32 limbs.
//
// Replaces py_ecc's `FQ.__add__/__sub__/__mul__/__truediv__/__pow__` as used through `Scalar`
// (/root/reference/curve.py:10-11) and `b.FQ` (the G1 coordinates in curve.py:38-44): every
// device-resident field element lives in Montgomery form, canonical (< m), little-endian limbs,
// 32 bytes — the same layout as one coordinate in a snarkjs .ptau file (setup.py:29-41), so SRS
// bytes upload without conversion.
//
// Both BN254 moduli are 254-bit, so 2m < 2^255: sums never carry out of the top limb and the CIOS
// accumulator needs a single extra word.  Multiplication is operand-scanning CIOS on 32-bit limbs;
// hipcc lowers each `(u64)a*b + c` to one v_mad_u64_u32.  Not a dense contraction: no MFMA.
#pragma once
#include "hip_compat.h"
#include "bn254_constants.h"

template <class P>
struct alignas(16) Fp {
    uint32_t v[8];
};
using Fr = Fp<FrParams>;
using Fq = Fp<FqParams>;

template <class P> PLONK_HD Fp<P> fp_zero() {
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = 0;
    return r;
}
template <class P> PLONK_HD Fp<P> fp_one() {
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = P::one(i);
    return r;
}
template <class P> PLONK_HD bool fp_is_zero(const Fp<P>& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i];
    return o == 0;
}
template <class P> PLONK_HD bool fp_eq(const Fp<P>& a, const Fp<P>& b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i] ^ b.v[i];
    return o == 0;
}

// r = t - m if t >= m else t      (t < 2m)
template <class P> PLONK_HD void fp_reduce_once(uint32_t t[8]) {
    uint32_t d[8];
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t x = (uint64_t)t[i] - P::mod(i) - br;
        d[i] = (uint32_t)x;
        br = (x >> 32) & 1;
    }
    if (!br) {
#pragma unroll
        for (int i = 0; i < 8; i++) t[i] = d[i];
    }
}

template <class P> PLONK_HD Fp<P> fp_add(const Fp<P>& a, const Fp<P>& b) {
    Fp<P> r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)a.v[i] + b.v[i];
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    fp_reduce_once<P>(r.v);
    return r;
}

template <class P> PLONK_HD Fp<P> fp_sub(const Fp<P>& a, const Fp<P>& b) {
    Fp<P> r;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t x = (uint64_t)a.v[i] - b.v[i] - br;
        r.v[i] = (uint32_t)x;
        br = (x >> 32) & 1;
    }
    if (br) {
        uint64_t c = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            c += (uint64_t)r.v[i] + P::mod(i);
            r.v[i] = (uint32_t)c;
            c >>= 32;
        }
    }
    return r;
}

template <class P> PLONK_HD Fp<P> fp_neg(const Fp<P>& a) {
    if (fp_is_zero(a)) return a;
    Fp<P> r;
    uint64_t br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t x = (uint64_t)P::mod(i) - a.v[i] - br;
        r.v[i] = (uint32_t)x;
        br = (x >> 32) & 1;
    }
    return r;
}

template <class P> PLONK_HD Fp<P> fp_dbl(const Fp<P>& a) { return fp_add(a, a); }

// Montgomery product a*b*R^-1 mod m.  Inputs < m, output < m.
template <class P> PLONK_HD Fp<P> fp_mul(const Fp<P>& a, const Fp<P>& b) {
    uint32_t t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
        const uint32_t bi = b.v[i];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c += (uint64_t)a.v[j] * bi + t[j];
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        t[8] += (uint32_t)c;  // t < 2m + (2^32-1) m: one extra word suffices, no further carry
        const uint32_t q = t[0] * P::NINV;
        c = ((uint64_t)q * P::mod(0) + t[0]) >> 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            c += (uint64_t)q * P::mod(j) + t[j];
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += t[8];
        t[7] = (uint32_t)c;
        t[8] = (uint32_t)(c >> 32);
    }
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
    fp_reduce_once<P>(r.v);
    return r;
}

template <class P> PLONK_HD Fp<P> fp_sqr(const Fp<P>& a) { return fp_mul(a, a); }

// canonical integer (< m, plain limbs) -> Montgomery form, and back
template <class P> PLONK_HD Fp<P> fp_to_mont(const Fp<P>& a) {
    Fp<P> r2;
#pragma unroll
    for (int i = 0; i < 8; i++) r2.v[i] = P::r2(i);
    return fp_mul(a, r2);
}
template <class P> PLONK_HD Fp<P> fp_from_mont(const Fp<P>& a) {
    Fp<P> one = fp_zero<P>();
    one.v[0] = 1;
    return fp_mul(a, one);
}

// a^e for a 256-bit exponent given as 8 x u32 limbs (left-to-right square and multiply).
template <class P> PLONK_HD Fp<P> fp_pow_limbs(const Fp<P>& a, const uint32_t e[8]) {
    Fp<P> r = fp_one<P>();
    bool started = false;
    for (int i = 7; i >= 0; i--) {
        for (int b = 31; b >= 0; b--) {
            if (started) r = fp_sqr(r);
            if ((e[i] >> b) & 1) {
                r = started ? fp_mul(r, a) : a;
                started = true;
            }
        }
    }
    return r;
}

template <class P> PLONK_HD Fp<P> fp_pow_u64(const Fp<P>& a, uint64_t e) {
    uint32_t l[8] = {(uint32_t)e, (uint32_t)(e >> 32), 0, 0, 0, 0, 0, 0};
    return fp_pow_limbs(a, l);
}

// Fermat inverse a^(m-2).  0 -> 0, which is exactly py_ecc's `x / 0 == 0`.
template <class P> PLONK_HD Fp<P> fp_inv(const Fp<P>& a) {
    uint32_t e[8];
#pragma unroll
    for (int i = 0; i < 8; i++) e[i] = P::mod_minus_2(i);
    return fp_pow_limbs(a, e);
}

// small-constant multiples used by the curve formulas
template <class P> PLONK_HD Fp<P> fp_mul3(const Fp<P>& a) { return fp_add(fp_dbl(a), a); }

// 32-byte loads/stores as two 16-byte vector accesses (global_load_dwordx4 / ds_read_b128)
struct alignas(16) u32x4 { uint32_t x, y, z, w; };
template <class P> PLONK_HD Fp<P> fp_load(const Fp<P>* p) {
    const u32x4* q = reinterpret_cast<const u32x4*>(p);
    u32x4 lo = q[0], hi = q[1];
    Fp<P> r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}
template <class P> PLONK_HD void fp_store(Fp<P>* p, const Fp<P>& a) {
    u32x4* q = reinterpret_cast<u32x4*>(p);
    q[0] = u32x4{a.v[0], a.v[1], a.v[2], a.v[3]};
    q[1] = u32x4{a.v[4], a.v[5], a.v[6], a.v[7]};
}
