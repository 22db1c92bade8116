// fpl.h — "lazy" field elements for the inner loops: 9 x 29-bit limbs kept unpacked in registers,
// values allowed to range over [0, 2^k m) instead of being canonical.
//
// Why: with R = 2^261 a Montgomery product a*b*R^-1 stays below 2m as long as (a/m)(b/m) < 128, so
// the long chains of a Pippenger mixed addition or an NTT butterfly never need a conditional
// subtraction; additions are 9 independent 32-bit adds (3 spare bits per limb), subtractions add a
// multiple of m spread over the limbs first, and a 24-op carry sweep (`fpl_norm`) restores 29-bit
// limbs before the next multiplication.  Compared with the packed canonical type of fp.h this drops
// the unpack / pack / compare-subtract around every operation: a mixed addition falls from ~3750 to
// ~2400 VALU instructions.
//
// Conventions: "normalised" = limbs 0..7 < 2^29 (limb 8 holds the rest, < 2^29 for values < 2^261).
// fpl_mul / fpl_sqr need normalised inputs with (a/m)(b/m) <= 128 and return a normalised value < 2m.
#pragma once
#include "fp.h"

template <class P>
struct FpL {
    uint32_t l[9];
};

template <class P> PLONK_HD FpL<P> fpl_from_fp(const Fp<P>& a) {
    FpL<P> r;
    fp29_unpack(a.v, r.l);
    return r;
}

// k * m as 9 limbs where every limb below the top is >= 2^30, so `x + spread(k) - y` never underflows
// a limb for normalised-ish y (limbs < 2^30).  Built from the plain limbs c_i of k*m by moving
// 2^30 * 2^(29 i) = 2 * 2^(29 (i+1)) from each limb to its lower neighbour.
template <class P> PLONK_HD constexpr uint32_t fpl_spread_limb(unsigned k, int i) {
    // plain limbs of k*m (k <= 16): accumulate k * mod29 with carries
    uint64_t carry = 0;
    uint32_t c = 0;
    for (int j = 0; j <= i; j++) {
        uint64_t v = (uint64_t)k * fp29_mod_limb<P>(j) + carry;
        if (j < 8) {
            c = (uint32_t)(v & FP29_MASK);
            carry = v >> 29;
        } else {
            c = (uint32_t)v;
        }
    }
    // borrow scheme: limb i gains 2^30 (i < 8) and loses 2 (i > 0)
    uint32_t out = c;
    if (i < 8) out += 1u << 30;
    if (i > 0) out -= 2u;
    return out;
}

template <class P> PLONK_HD FpL<P> fpl_add(const FpL<P>& a, const FpL<P>& b) {
    FpL<P> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}

// a - b + K*m, limb-wise; K (compile-time) must be >= the bound of b in units of m.  Limbs < 2^31.4.
template <class P, unsigned K> PLONK_HD FpL<P> fpl_sub(const FpL<P>& a, const FpL<P>& b) {
    FpL<P> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + fpl_spread_limb<P>(K, i) - b.l[i];
    return r;
}

// carry sweep: limbs 0..7 back below 2^29 (input limbs < 2^32)
template <class P> PLONK_HD FpL<P> fpl_norm(const FpL<P>& a) {
    FpL<P> r;
    uint32_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t v = a.l[i] + carry;
        r.l[i] = v & FP29_MASK;
        carry = v >> 29;
    }
    r.l[8] = a.l[8] + carry;
    return r;
}

template <class P> PLONK_HD FpL<P> fpl_mul(const FpL<P>& a, const FpL<P>& b) {
    const uint32_t ninv = P::NINV & FP29_MASK;
    uint32_t q[9];
    FpL<P> r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)q[i] * fp29_mod_limb<P>(k - i);
        q[k] = ((uint32_t)acc * ninv) & FP29_MASK;
        acc += (uint64_t)q[k] * fp29_mod_limb<P>(0);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (uint64_t)q[i] * fp29_mod_limb<P>(k - i);
        r.l[k - 9] = (uint32_t)acc & FP29_MASK;
        acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    return r;
}

// (a*b + c*d) R^-1 with ONE Montgomery reduction: 162 product + 81 reduction multiplier instructions instead
// of 2 x 162.  Needs normalised inputs with (a/m)(b/m) + (c/m)(d/m) <= 128; a column holds at most 27
// products < 2^58 plus a carry, which still fits the 64-bit accumulator.  Returns a normalised value < 2m.
template <class P> PLONK_HD FpL<P> fpl_mul_add(const FpL<P>& a, const FpL<P>& b, const FpL<P>& c, const FpL<P>& d) {
    const uint32_t ninv = P::NINV & FP29_MASK;
    uint32_t q[9];
    FpL<P> r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)c.l[i] * d.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)q[i] * fp29_mod_limb<P>(k - i);
        q[k] = ((uint32_t)acc * ninv) & FP29_MASK;
        acc += (uint64_t)q[k] * fp29_mod_limb<P>(0);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (uint64_t)c.l[i] * d.l[k - i];
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (uint64_t)q[i] * fp29_mod_limb<P>(k - i);
        r.l[k - 9] = (uint32_t)acc & FP29_MASK;
        acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    return r;
}

template <class P> PLONK_HD FpL<P> fpl_sqr(const FpL<P>& a) {
    const uint32_t ninv = P::NINV & FP29_MASK;
    uint32_t q[9], a2[9];
#pragma unroll
    for (int i = 0; i < 9; i++) a2[i] = a.l[i] << 1;
    FpL<P> r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) {
            const int j = k - i;
            if (i > 8 || j > 8 || i > j) continue;
            acc += (i == j) ? (uint64_t)a.l[i] * a.l[i] : (uint64_t)a2[i] * a.l[j];
        }
        if (k < 9) {
#pragma unroll
            for (int i = 0; i < k; i++) acc += (uint64_t)q[i] * fp29_mod_limb<P>(k - i);
            q[k] = ((uint32_t)acc * ninv) & FP29_MASK;
            acc += (uint64_t)q[k] * fp29_mod_limb<P>(0);
        } else {
#pragma unroll
            for (int i = k - 8; i < 9; i++) acc += (uint64_t)q[i] * fp29_mod_limb<P>(k - i);
            r.l[k - 9] = (uint32_t)acc & FP29_MASK;
        }
        acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    return r;
}

// R mod m as limbs (the Montgomery form of 1): multiplying by it maps any value < 128 m to < 2m.
template <class P> PLONK_HD FpL<P> fpl_one() {
    Fp<P> o = fp_one<P>();
    return fpl_from_fp(o);
}

// normalised value of any size < 128 m -> canonical packed element
template <class P> PLONK_HD Fp<P> fpl_to_fp(const FpL<P>& a) {
    FpL<P> t = fpl_mul(a, fpl_one<P>());  // < 2m, normalised
    Fp<P> out;
    fp29_pack(t.l, out.v);
    fp_reduce_once<P>(out.v);
    return out;
}

template <class P> PLONK_HD_NOINLINE bool fpl_is_zero_mod_slow(const FpL<P>& a) { return fp_is_zero(fpl_to_fp(a)); }

// exact test "a == 0 (mod m)" for a normalised value known to be < 16 m (rarely true: cheap filter first)
template <class P> PLONK_HD bool fpl_is_zero_mod(const FpL<P>& a) {
    // a == j*m for some j in 0..15  =>  limb 0 equals limb 0 of j*m
    bool maybe = false;
#pragma unroll
    for (unsigned j = 0; j < 16; j++) {
        const uint32_t l0 = (uint32_t)(((uint64_t)j * fp29_mod_limb<P>(0)) & FP29_MASK);
        maybe |= a.l[0] == l0;
    }
    if (!maybe) return false;
    return fpl_is_zero_mod_slow(a);
}
