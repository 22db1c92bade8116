// fpl.h — "lazy" field elements for the inner loops: 9 SIGNED limbs of 29 bits kept unpacked in registers,
// values allowed to range over (-2^k m, 2^k m) instead of being canonical.
//
// Why: with R = 2^261 a Montgomery product a*b*R^-1 stays inside (-m, 2m) as long as |a/m| |b/m| <= 128, so the long
// chains of a Pippenger mixed addition never need a conditional subtraction; additions and subtractions are 9
// independent 32-bit operations (3 spare bits per limb), and because the limbs — and the multiply-adds, v_mad_i64_i32 —
// are SIGNED, a difference needs neither an added multiple of m nor a carry sweep before it is multiplied: the
// difference of two normalised values has limbs in (-2^29, 2^29), as good as normalised ones.  On gfx950 every VALU
// instruction of this code, multiplier or not, issues at ~4.5 cycles per wave (tools/ubench/ubench3.hip), so the
// instructions AROUND the multiplications count as much as the multiplications: round 1's unsigned form paid five
// carry sweeps (24 instructions each) and five spread-constant additions per mixed addition, this form pays one sweep.
//
// Conventions.  "normalised": limbs 0..7 in [0, 2^29), limb 8 signed (it carries the sign of the value).
//   fpl_mul / fpl_sqr / fpl_mul_add take operands whose limbs are within (-2^30, 2^30) — a normalised value, or the sum
//   or difference of two — PROVIDED the column sums stay inside 64 bits: at most one operand of a product may exceed
//   2^29 in limb magnitude (9 * 2^59 + 9 * 2^58 < 2^63); fpl_sqr and both products of fpl_mul_add need limbs within
//   (-2^29, 2^29) (normalised values or their differences).  Value bound: sum of |a/m| |b/m| <= 128.
//   They return a normalised value in (-m, 2m).
#pragma once
#include <math.h>
#include "fp.h"

template <class P>
struct FpL {
    int32_t l[9];
};

// Range checks of the emulator build (tests/emu, -DPLONK_EMU: the same sources compiled for the host by the CPU test-suite):
// every operation below asserts the operand bounds its header documents — no int32 limb overflow, column sums inside 64
// bits, |a||b| <= 128 m^2 — on the data the emulated kernels actually see (random, golden and adversarial inputs of the NTT,
// MSM and prover tests).  They compile to nothing in the product.
#ifdef PLONK_EMU
#include <stdio.h>
#include <stdlib.h>
#define FPL_CHECK(cond, what)                                                       \
    do {                                                                            \
        if (!(cond)) {                                                              \
            fprintf(stderr, "fpl.h range check failed: %s (%s:%d)\n", what, __FILE__, __LINE__); \
            abort();                                                                \
        }                                                                           \
    } while (0)
template <class L> inline long double fpl_dbg_value(const L& a) {  // the value, good to 64 bits: enough for bounds
    long double v = 0;
    for (int i = 8; i >= 0; i--) v = v * 536870912.0L + (long double)a.l[i];
    return v;
}
template <class L> inline long long fpl_dbg_maxlimb(const L& a) {
    long long m = 0;
    for (int i = 0; i < 9; i++) {
        long long v = a.l[i] < 0 ? -(long long)a.l[i] : (long long)a.l[i];
        if (v > m) m = v;
    }
    return m;
}
template <class P, class L> inline void fpl_dbg_check_product(const L& a, const L& b, const L* c, const L* d) {
    // column sums: 9 products per pair of operands + 9 reduction products < 2^58 + carries, inside (-2^63, 2^63)
    long double col = 9.0L * (long double)fpl_dbg_maxlimb(a) * (long double)fpl_dbg_maxlimb(b) + 9.0L * 288230376151711744.0L + 1099511627776.0L;
    long double mm = 0;
    for (int i = 8; i >= 0; i--) mm = mm * 536870912.0L + (long double)fp29_mod_limb<P>(i);
    long double val = fabsl(fpl_dbg_value(a)) * fabsl(fpl_dbg_value(b));
    if (c) {
        col += 9.0L * (long double)fpl_dbg_maxlimb(*c) * (long double)fpl_dbg_maxlimb(*d);
        val += fabsl(fpl_dbg_value(*c)) * fabsl(fpl_dbg_value(*d));
    }
    if (!(col < 9223372036854775808.0L))
        fprintf(stderr, "fpl.h: max |limb| a %lld b %lld c %lld d %lld\n", fpl_dbg_maxlimb(a), fpl_dbg_maxlimb(b), c ? fpl_dbg_maxlimb(*c) : 0LL, d ? fpl_dbg_maxlimb(*d) : 0LL);
    FPL_CHECK(col < 9223372036854775808.0L, "column sum of a product exceeds 64 bits");
    // (-m, 2m) needs |a||b| <= R m: 169 m^2 for BN254 (the documented 128 leaves slack), 70 m^2 for BLS12-381 Fr
    const long double r_over_m = ldexpl(1.0L, 261) / mm;
    if (!(val <= (r_over_m < 128.0L ? r_over_m : 128.0L) * mm * mm))
        fprintf(stderr, "fpl.h: |a| = %.3Lf m, |b| = %.3Lf m, R / m = %.2Lf\n", fabsl(fpl_dbg_value(a)) / mm, fabsl(fpl_dbg_value(b)) / mm, r_over_m);
    FPL_CHECK(val <= (r_over_m < 128.0L ? r_over_m : 128.0L) * mm * mm, "|a||b| exceeds min(128, R / m) m^2");
}
#define FPL_CHECK_I32(expr64, what) FPL_CHECK((expr64) >= -2147483648LL && (expr64) <= 2147483647LL, what)
#else
#define FPL_CHECK(cond, what) ((void)0)
#define FPL_CHECK_I32(expr64, what) ((void)0)
#endif

// Hides from the compiler that a limb is known to be non-negative.  Without it LLVM multiplies a signed limb by a
// masked one as sext x zext — a v_mad_u64_u32 plus a correction v_mad_u64_u32 with the sign mask — instead of one
// v_mad_i64_i32 (24 extra multiplier instructions and 48 moves per mixed addition when measured).  No instruction.
// PLONK_CHAIN (hip_compat.h) keeps each column sum a chain that starts from the carry.
#if defined(__HIP_DEVICE_COMPILE__)
#define FPL_ANY_SIGN(x) asm("" : "+v"(x))
// the same for a wave-uniform value, which is moved to (and stays in) a scalar register
#define FPL_ANY_SIGN_UNIFORM(x)                           \
    do {                                                  \
        (x) = __builtin_amdgcn_readfirstlane((int)(x));   \
        asm("" : "+s"(x));                                \
    } while (0)
#else
#define FPL_ANY_SIGN(x) ((void)0)
#define FPL_ANY_SIGN_UNIFORM(x) ((void)0)
#endif

template <class P> PLONK_HD FpL<P> fpl_from_fp(const Fp<P>& a) {
    uint32_t u[9];
    fp29_unpack(a.v, u);
    FpL<P> r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        r.l[i] = (int32_t)u[i];
        FPL_ANY_SIGN(r.l[i]);
    }
    return r;
}

// the same for a wave-uniform element (a kernel argument): the limbs stay in scalar registers
template <class P> PLONK_HD FpL<P> fpl_from_fp_uniform(const Fp<P>& a) {
    uint32_t u[9];
    fp29_unpack(a.v, u);
    FpL<P> r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        r.l[i] = (int32_t)u[i];
        FPL_ANY_SIGN_UNIFORM(r.l[i]);
    }
    return r;
}

template <class P> PLONK_HD FpL<P> fpl_zero() {
    FpL<P> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = 0;
    return r;
}

template <class P> PLONK_HD FpL<P> fpl_add(const FpL<P>& a, const FpL<P>& b) {
    FpL<P> r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        FPL_CHECK_I32((long long)a.l[i] + b.l[i], "fpl_add overflows a limb");
        r.l[i] = a.l[i] + b.l[i];
    }
    return r;
}

template <class P> PLONK_HD FpL<P> fpl_sub(const FpL<P>& a, const FpL<P>& b) {
    FpL<P> r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        FPL_CHECK_I32((long long)a.l[i] - b.l[i], "fpl_sub overflows a limb");
        r.l[i] = a.l[i] - b.l[i];
    }
    return r;
}

template <class P> PLONK_HD FpL<P> fpl_neg(const FpL<P>& a) {
    FpL<P> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = -a.l[i];
    return r;
}

// neg ? -a : a without a branch: (x ^ s) - s with s = 0 or -1
template <class P> PLONK_HD FpL<P> fpl_cneg(const FpL<P>& a, bool neg) {
    const int32_t s = neg ? -1 : 0;
    FpL<P> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = (a.l[i] ^ s) - s;
    return r;
}

// carry sweep: limbs 0..7 back into [0, 2^29), the sign moves to limb 8 (input limbs: any int32)
template <class P> PLONK_HD FpL<P> fpl_norm(const FpL<P>& a) {
    FpL<P> r;
    int32_t carry = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        FPL_CHECK_I32((long long)a.l[i] + carry, "fpl_norm overflows a limb");
        const int32_t v = a.l[i] + carry;
        r.l[i] = v & (int32_t)FP29_MASK;
        FPL_ANY_SIGN(r.l[i]);
        carry = v >> 29;  // arithmetic: floor division
    }
    FPL_CHECK_I32((long long)a.l[8] + carry, "fpl_norm overflows the top limb");
    r.l[8] = a.l[8] + carry;
    return r;
}

template <class P> PLONK_HD FpL<P> fpl_mul(const FpL<P>& a, const FpL<P>& b) {
#ifdef PLONK_EMU
    fpl_dbg_check_product<P>(a, b, (const FpL<P>*)nullptr, (const FpL<P>*)nullptr);
#endif
    const uint32_t ninv = P::NINV & FP29_MASK;
    uint32_t q[9];
    FpL<P> r;
    int64_t acc = 0;
    PLONK_CHAIN_BEGIN();
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) {
            acc += (int64_t)a.l[i] * b.l[k - i];
            PLONK_CHAIN(acc);
        }
#pragma unroll
        for (int i = 0; i < k; i++) {
            acc += (int64_t)((uint64_t)q[i] * fp29_mod_limb<P>(k - i));
            PLONK_CHAIN(acc);
        }
        q[k] = ((uint32_t)acc * ninv) & FP29_MASK;
        acc += (int64_t)((uint64_t)q[k] * fp29_mod_limb<P>(0));
        acc >>= 29;  // exact: the low 29 bits are zero
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) {
            acc += (int64_t)a.l[i] * b.l[k - i];
            PLONK_CHAIN(acc);
        }
#pragma unroll
        for (int i = k - 8; i < 9; i++) {
            acc += (int64_t)((uint64_t)q[i] * fp29_mod_limb<P>(k - i));
            PLONK_CHAIN(acc);
        }
        r.l[k - 9] = (int32_t)((uint32_t)acc & FP29_MASK);
        FPL_ANY_SIGN(r.l[k - 9]);
        acc >>= 29;
    }
    r.l[8] = (int32_t)acc;
    PLONK_CHAIN_END(r.l[8]);
    return r;
}

// (a*b + c*d) R^-1 with ONE Montgomery reduction: 162 product + 81 reduction multiplier instructions instead of
// 2 x 171.  All four operands with limbs within (-2^29, 2^29): a column holds at most 18 products below 2^58 plus 9
// reduction products, inside 64 bits.  Returns a normalised value in (-m, 2m).
template <class P> PLONK_HD FpL<P> fpl_mul_add(const FpL<P>& a, const FpL<P>& b, const FpL<P>& c, const FpL<P>& d) {
#ifdef PLONK_EMU
    fpl_dbg_check_product<P>(a, b, &c, &d);
#endif
    const uint32_t ninv = P::NINV & FP29_MASK;
    uint32_t q[9];
    FpL<P> r;
    int64_t acc = 0;
    PLONK_CHAIN_BEGIN();
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) {
            acc += (int64_t)a.l[i] * b.l[k - i];
            PLONK_CHAIN(acc);
        }
#pragma unroll
        for (int i = 0; i <= k; i++) {
            acc += (int64_t)c.l[i] * d.l[k - i];
            PLONK_CHAIN(acc);
        }
#pragma unroll
        for (int i = 0; i < k; i++) {
            acc += (int64_t)((uint64_t)q[i] * fp29_mod_limb<P>(k - i));
            PLONK_CHAIN(acc);
        }
        q[k] = ((uint32_t)acc * ninv) & FP29_MASK;
        acc += (int64_t)((uint64_t)q[k] * fp29_mod_limb<P>(0));
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) {
            acc += (int64_t)a.l[i] * b.l[k - i];
            PLONK_CHAIN(acc);
        }
#pragma unroll
        for (int i = k - 8; i < 9; i++) {
            acc += (int64_t)c.l[i] * d.l[k - i];
            PLONK_CHAIN(acc);
        }
#pragma unroll
        for (int i = k - 8; i < 9; i++) {
            acc += (int64_t)((uint64_t)q[i] * fp29_mod_limb<P>(k - i));
            PLONK_CHAIN(acc);
        }
        r.l[k - 9] = (int32_t)((uint32_t)acc & FP29_MASK);
        FPL_ANY_SIGN(r.l[k - 9]);
        acc >>= 29;
    }
    r.l[8] = (int32_t)acc;
    PLONK_CHAIN_END(r.l[8]);
    return r;
}

// limbs within (-2^29, 2^29)
template <class P> PLONK_HD FpL<P> fpl_sqr(const FpL<P>& a) {
#ifdef PLONK_EMU
    fpl_dbg_check_product<P>(a, a, (const FpL<P>*)nullptr, (const FpL<P>*)nullptr);
#endif
    const uint32_t ninv = P::NINV & FP29_MASK;
    uint32_t q[9];
    int32_t a2[9];
#pragma unroll
    for (int i = 0; i < 9; i++) a2[i] = a.l[i] * 2;
    FpL<P> r;
    int64_t acc = 0;
    PLONK_CHAIN_BEGIN();
#pragma unroll
    for (int k = 0; k < 17; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) {
            const int j = k - i;
            if (i > 8 || j > 8 || i > j) continue;
            acc += (i == j) ? (int64_t)a.l[i] * a.l[i] : (int64_t)a2[i] * a.l[j];
            PLONK_CHAIN(acc);
        }
        if (k < 9) {
#pragma unroll
            for (int i = 0; i < k; i++) {
            acc += (int64_t)((uint64_t)q[i] * fp29_mod_limb<P>(k - i));
            PLONK_CHAIN(acc);
        }
            q[k] = ((uint32_t)acc * ninv) & FP29_MASK;
            acc += (int64_t)((uint64_t)q[k] * fp29_mod_limb<P>(0));
        } else {
#pragma unroll
            for (int i = k - 8; i < 9; i++) {
            acc += (int64_t)((uint64_t)q[i] * fp29_mod_limb<P>(k - i));
            PLONK_CHAIN(acc);
        }
            r.l[k - 9] = (int32_t)((uint32_t)acc & FP29_MASK);
            FPL_ANY_SIGN(r.l[k - 9]);
        FPL_ANY_SIGN(r.l[k - 9]);
        }
        acc >>= 29;
    }
    r.l[8] = (int32_t)acc;
    PLONK_CHAIN_END(r.l[8]);
    return r;
}

// R mod m as limbs (the Montgomery form of 1): multiplying by it maps any value within (-128 m, 128 m) into (-m, 2m).
template <class P> PLONK_HD FpL<P> fpl_one() {
    Fp<P> o = fp_one<P>();
    return fpl_from_fp(o);
}

// limb i of k*m (k small), 29-bit digits with the excess in limb 8
template <class P> PLONK_HD constexpr int32_t fpl_km_limb(unsigned k, int i) {
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
    return (int32_t)c;
}

// a + K*m, normalised: for a in (-K m, ...) the result is non-negative
template <class P, unsigned K> PLONK_HD FpL<P> fpl_add_km_norm(const FpL<P>& a) {
    FpL<P> t;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        FPL_CHECK_I32((long long)a.l[i] + fpl_km_limb<P>(K, i), "fpl_add_km_norm overflows a limb");
        t.l[i] = a.l[i] + fpl_km_limb<P>(K, i);
    }
    const FpL<P> r = fpl_norm(t);
    FPL_CHECK(r.l[8] >= 0, "fpl_add_km_norm: the value was below -K m");
    return r;
}

// any value within (-128 m, 128 m) (limbs within (-2^30, 2^30)) -> canonical packed element
template <class P> PLONK_HD Fp<P> fpl_to_fp(const FpL<P>& a) {
    const FpL<P> t = fpl_add_km_norm<P, 1>(fpl_mul(a, fpl_one<P>()));  // (-m, 2m) + m = (0, 3m), normalised
    uint32_t u[9];
#pragma unroll
    for (int i = 0; i < 9; i++) u[i] = (uint32_t)t.l[i];
    Fp<P> out;
    fp29_pack(u, out.v);
    fp_reduce_once<P>(out.v);
    fp_reduce_once<P>(out.v);
    return out;
}

template <class P> PLONK_HD_NOINLINE bool fpl_is_zero_mod_slow(const FpL<P>& a) { return fp_is_zero(fpl_to_fp(a)); }

// cheap filter for "a == 0 (mod m)" when a is known to lie within [JLO m, JHI m] (limb 0: any int32): a = j*m forces
// limb0 * (-1/m) == -j (mod 2^29), so one multiplication by the Montgomery constant recovers the candidate j and a range
// check replaces a comparison per j.  True with probability (JHI - JLO + 1) 2^-29 for a random a.
template <class P, int JLO, int JHI> PLONK_HD bool fpl_maybe_zero_mod(const FpL<P>& a) {
    const uint32_t minus_j = ((uint32_t)a.l[0] * (P::NINV & FP29_MASK)) & FP29_MASK;
    return ((minus_j + (uint32_t)JHI) & FP29_MASK) <= (uint32_t)(JHI - JLO);
}

// exact, inline and call-free: "a == 0 (mod m)" for a within [JLO m, JHI m] (limbs: any int32 whose sweep does not overflow).
// The filter above names the only candidate j; a - j m is then carry-swept and compared with zero (~45 instructions, reached
// with probability 2^-26).  For the reductions that must not contain a call (g1l_add_fast in the MSM kernels' trees).
template <class P, int JLO, int JHI> PLONK_HD bool fpl_is_zero_mod_in(const FpL<P>& a) {
    const uint32_t minus_j = ((uint32_t)a.l[0] * (P::NINV & FP29_MASK)) & FP29_MASK;
    const uint32_t t = (minus_j + (uint32_t)JHI) & FP29_MASK;
    if (t > (uint32_t)(JHI - JLO)) return false;
    const int64_t j = (int64_t)JHI - (int64_t)t;
    int64_t carry = 0;
    uint32_t nz = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int64_t v = (int64_t)a.l[i] - j * (int64_t)fp29_mod_limb<P>(i) + carry;
        if (i < 8) {
            nz |= (uint32_t)v & FP29_MASK;
            carry = v >> 29;  // arithmetic: floor
        } else {
            nz |= (uint32_t)v | (uint32_t)((uint64_t)v >> 32);
        }
    }
    return nz == 0;
}

// exact test, a within (-16 m, 16 m)
template <class P> PLONK_HD bool fpl_is_zero_mod(const FpL<P>& a) {
    if (!fpl_maybe_zero_mod<P, -15, 15>(a)) return false;
    return fpl_is_zero_mod_slow(a);
}

// ---- cheap range reduction for values that are not about to be multiplied ---------------------------------------------
// jm = table of j*m, j = -FPL_RS_J .. FPL_RS_J, 12 int32 per entry (9 used): limbs 0..7 in [0, 2^29), limb 8 signed.
// x: |value| <= FPL_RS_J - 1 times m; limbs 0..7 within (-2^30, 2^31 - 2^29), limb 8 small.  The multiple of m nearest to
// x is estimated from the top limb alone (the lower limbs contribute < 2^-19 of m even un-normalised) and subtracted limb
// by limb; one carry sweep.  Result: normalised, value within (-0.51 m, 0.51 m).  ~41 instructions against ~235 for a
// multiplication by one — what the butterfly outputs that carry no twiddle factor get in the limb-form NTT kernel.
#define FPL_RS_J 24
// BIAS = 1 subtracts (j - 1) m instead: the result lies within (0.49 m, 1.51 m) — what the last store wants, one
// conditional subtraction away from canonical (fpl_pack_positive).
// (in two halves, so that a caller with several values to reduce can request all the table entries before it needs the first:
// fpl_reduce_small_lookup issues the loads, fpl_reduce_small_apply subtracts)
struct FplJmEntry {
    u32x4 t0, t1;
    int32_t t8;
};
template <class P, int BIAS = 0> PLONK_HD FplJmEntry fpl_reduce_small_lookup(const FpL<P>& x, const int32_t* jm) {
    constexpr float inv_top = 1.0f / (float)(P::mod(7) >> 8);  // m >> 232
    int j = (int)rintf((float)x.l[8] * inv_top);
    FPL_CHECK(j > -FPL_RS_J && j < FPL_RS_J, "fpl_reduce_small: |value| beyond the table");
    j = j < -FPL_RS_J + BIAS ? -FPL_RS_J + BIAS : (j > FPL_RS_J ? FPL_RS_J : j);  // (memory safety only: the bound above keeps |j| <= FPL_RS_J - 1)
    const int32_t* t = jm + (j + FPL_RS_J - BIAS) * 12;
    FplJmEntry e;
    e.t0 = *reinterpret_cast<const u32x4*>(t);
    e.t1 = *reinterpret_cast<const u32x4*>(t + 4);
    e.t8 = t[8];
    return e;
}
template <class P, int BIAS = 0> PLONK_HD FpL<P> fpl_reduce_small_apply(const FpL<P>& x, const FplJmEntry& e) {
    const u32x4 t0 = e.t0, t1 = e.t1;
    const int32_t t8 = e.t8;
    FpL<P> r;
    r.l[0] = x.l[0] - (int32_t)t0.x; r.l[1] = x.l[1] - (int32_t)t0.y; r.l[2] = x.l[2] - (int32_t)t0.z; r.l[3] = x.l[3] - (int32_t)t0.w;
    r.l[4] = x.l[4] - (int32_t)t1.x; r.l[5] = x.l[5] - (int32_t)t1.y; r.l[6] = x.l[6] - (int32_t)t1.z; r.l[7] = x.l[7] - (int32_t)t1.w;
    r.l[8] = x.l[8] - t8;
#ifdef PLONK_EMU
    {
        const int32_t tt[9] = {(int32_t)t0.x, (int32_t)t0.y, (int32_t)t0.z, (int32_t)t0.w, (int32_t)t1.x, (int32_t)t1.y, (int32_t)t1.z, (int32_t)t1.w, t8};
        for (int i = 0; i < 9; i++) FPL_CHECK_I32((long long)x.l[i] - tt[i], "fpl_reduce_small overflows a limb");
        long double mm = 0;
        for (int i = 8; i >= 0; i--) mm = mm * 536870912.0L + (long double)fp29_mod_limb<P>(i);
        FPL_CHECK(fabsl(fpl_dbg_value(r) - (long double)BIAS * mm) < 0.51L * mm, "fpl_reduce_small: result outside (-0.51 m, 0.51 m) (+ BIAS m)");
    }
#endif
    return fpl_norm(r);
}
template <class P, int BIAS = 0> PLONK_HD FpL<P> fpl_reduce_small(const FpL<P>& x, const int32_t* jm) {
    return fpl_reduce_small_apply<P, BIAS>(x, fpl_reduce_small_lookup<P, BIAS>(x, jm));
}

// host side: entry j of that table
template <class P> inline void fpl_jm_entry(int j, int32_t out[12]) {
    long long carry = 0;
    for (int i = 0; i < 9; i++) {
        long long v = (long long)j * (long long)fp29_mod_limb<P>(i) + carry;
        if (i < 8) {
            out[i] = (int32_t)(v & (long long)FP29_MASK);
            carry = v >> 29;  // arithmetic: floor
        } else {
            out[i] = (int32_t)v;
        }
    }
    out[9] = out[10] = out[11] = 0;
}

// normalised value within (-m, 2m) -> canonical packed element
template <class P> PLONK_HD Fp<P> fpl_pack_canonical(const FpL<P>& a) {
    const FpL<P> t = fpl_add_km_norm<P, 1>(a);  // (0, 3m), limbs non-negative
    uint32_t u[9];
#pragma unroll
    for (int i = 0; i < 9; i++) u[i] = (uint32_t)t.l[i];
    Fp<P> out;
    fp29_pack(u, out.v);
    fp_reduce_once<P>(out.v);
    fp_reduce_once<P>(out.v);
    return out;
}

// ---- multiplication by a constant known in advance (the NTT's twiddle factors): Shoup / Barrett form ---------------------
// For a constant w < m with wp = floor(w 2^261 / m) precomputed, a * w mod m needs no reduction pass over a full product:
//   q = floor(a wp / 2^261)   the TOP half of one product (columns 7..16 of 17: two guard columns bound the error to 1)
//   r = a w - q m             the BOTTOM halves of two products (mod 2^261)
// 53 + 45 + 45 = 143 multiply-adds and 19 column steps against fpl_mul's 171 and 17, and no serial q_k = acc * (-1/m)
// chain.  The data stays in Montgomery form (a = x R): a w = (x w) R, so w is the PLAIN value of the constant.
//   a: limbs within fpl_mul's operand range (|limb| < 1.27 * 2^30), |value| < 128 m.
//   result: normalised, value within (-(1 + A) m, (2 + A) m) with A = |a| / R   [exact r in (-A m, (1 + A) m); q off by at
//   most one either way].  BN254 (R / m = 169): (-1.8 m, 2.8 m) for any |a| < 128 m.  BLS12-381 Fr (R / m = 70.7): the same
//   interval as long as |a| < 56 m — the NTT's multiplicands are sums of at most eight such results, |a| < 18.1 m.
// wp from the Montgomery form wt = w R mod m of the constant (what the root tables hold): w R = wp m + wt, hence
// wp = wt * (-1/m) mod 2^261 — one bottom-half product with the 261-bit constant fpl_ninv261.
template <class P> struct FpLS {
    int32_t w[9];   // w, 29-bit limbs
    int32_t wp[9];  // floor(w 2^261 / m), 29-bit limbs
};

// bottom half of a product of two 9-limb non-negative numbers (host / table construction)
PLONK_HD void fpl_limbs_mul_low(const uint32_t a[9], const uint32_t b[9], uint32_t out[9]) {
    uint64_t acc = 0;
    for (int k = 0; k < 9; k++) {
        for (int i = 0; i <= k; i++) {
            const uint64_t t = (uint64_t)a[i] * b[k - i];  // < 2^58: nine of them and a carry stay inside 64 bits
            acc += t;
        }
        out[k] = (uint32_t)acc & FP29_MASK;
        acc >>= 29;
    }
}

// -1/m mod 2^261 as nine 29-bit limbs: Newton / Hensel lifting from 1/m = 1 (mod 2), ten doublings
template <class P> PLONK_HD void fpl_ninv261(uint32_t out[9]) {
    uint32_t m[9], inv[9] = {1, 0, 0, 0, 0, 0, 0, 0, 0}, t[9], u[9];
    for (int i = 0; i < 9; i++) m[i] = fp29_mod_limb<P>(i);
    for (int it = 0; it < 10; it++) {
        fpl_limbs_mul_low(m, inv, t);  // m * inv = 1 (mod 2^bits)
        uint32_t borrow = 0;           // u = 2 - t (mod 2^261)
        for (int i = 0; i < 9; i++) {
            const int64_t v = (int64_t)(i == 0 ? 2 : 0) - (int64_t)t[i] - (int64_t)borrow;
            u[i] = (uint32_t)((uint64_t)v & FP29_MASK);
            borrow = v < 0 ? 1u : 0u;
        }
        fpl_limbs_mul_low(inv, u, t);
        for (int i = 0; i < 9; i++) inv[i] = t[i];
    }
    uint32_t borrow = 0;  // out = -inv (mod 2^261)
    for (int i = 0; i < 9; i++) {
        const int64_t v = -(int64_t)inv[i] - (int64_t)borrow;
        out[i] = (uint32_t)((uint64_t)v & FP29_MASK);
        borrow = v < 0 ? 1u : 0u;
    }
}

// the Shoup pair of a constant given in canonical Montgomery form
template <class P> PLONK_HD FpLS<P> fpl_shoup_from_mont(const Fp<P>& wt, const uint32_t ninv261[9]) {
    FpLS<P> r;
    uint32_t t[9], wp[9];
    fp29_unpack(wt.v, t);
    fpl_limbs_mul_low(t, ninv261, wp);
    FpL<P> one_plain = fpl_zero<P>();
    one_plain.l[0] = 1;
    const Fp<P> plain = fpl_pack_canonical(fpl_mul(fpl_from_fp(wt), one_plain));  // w = wt / R mod m, canonical
    fp29_unpack(plain.v, t);
    for (int i = 0; i < 9; i++) {
        r.w[i] = (int32_t)t[i];
        r.wp[i] = (int32_t)wp[i];
    }
    return r;
}

// -(limb j of the modulus) as a multiplier.  A limb that is a power of two or 2^29 minus one (BLS12-381 Fr: 0x1 and 0x1ffffff8)
// is hidden from the compiler, which otherwise "strength-reduces" the multiply-add into a sign extension and 64-bit shifts /
// subtractions: three to four VALU instructions (and their s_nop hazards) instead of one, on a chip where v_mad_i64_i32
// issues as fast as an addition — the BLS12-381 kernels were 8 % longer than the BN254 ones for having two such limbs.
PLONK_HD constexpr bool fpl_limb_is_trivial(uint32_t c) { return (c & (c - 1)) == 0 || (((1u << 29) - c) & ((1u << 29) - c - 1)) == 0; }
template <class P> PLONK_HD int32_t fpl_neg_mod_limb(int j) {
    int32_t c = -(int32_t)fp29_mod_limb<P>(j);
#if defined(__HIP_DEVICE_COMPILE__)
    if (fpl_limb_is_trivial(fp29_mod_limb<P>(j))) asm("" : "+s"(c));
#endif
    return c;
}

// FENCE: keep the scheduler from starting the second half (and its loads of w) before the first is done — for kernels at
// their register limit
template <class P, bool FENCE = false> PLONK_HD FpL<P> fpl_mul_shoup(const FpL<P>& a, const FpLS<P>& c) {
#ifdef PLONK_EMU
    {
        long double mm = 0;
        for (int i = 8; i >= 0; i--) mm = mm * 536870912.0L + (long double)fp29_mod_limb<P>(i);
        FPL_CHECK(fabsl(fpl_dbg_value(a)) < 128.0L * mm, "fpl_mul_shoup: |a| exceeds 128 m");
        FPL_CHECK(9.0L * (long double)fpl_dbg_maxlimb(a) * 536870912.0L + 9.0L * 288230376151711744.0L + 1099511627776.0L < 9223372036854775808.0L,
                  "fpl_mul_shoup: column sum exceeds 64 bits");
    }
#endif
    int32_t q[9];
    int64_t acc = 0;
    PLONK_CHAIN_BEGIN();
    // q = floor(a * wp / 2^261): columns 7 and 8 only feed the carry
#pragma unroll
    for (int k = 7; k < 17; k++) {
#pragma unroll
        for (int i = (k > 8 ? k - 8 : 0); i <= (k < 8 ? k : 8); i++) {
            acc += (int64_t)a.l[i] * c.wp[k - i];
            PLONK_CHAIN(acc);
        }
        if (k >= 9) {
            q[k - 9] = (int32_t)((uint32_t)acc & FP29_MASK);
            FPL_ANY_SIGN(q[k - 9]);
        }
        acc >>= 29;  // arithmetic: floor
    }
    q[8] = (int32_t)acc;
    if (FENCE) PLONK_SCHED_FENCE();
    // r = a * w - q * m  (mod 2^261)
    FpL<P> r;
    acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) {
            acc += (int64_t)a.l[i] * c.w[k - i];
            PLONK_CHAIN(acc);
        }
#pragma unroll
        for (int i = 0; i <= k; i++) {
            acc += (int64_t)q[i] * fpl_neg_mod_limb<P>(k - i);
            PLONK_CHAIN(acc);
        }
        if (k < 8) {
            r.l[k] = (int32_t)((uint32_t)acc & FP29_MASK);
            FPL_ANY_SIGN(r.l[k]);
            acc >>= 29;
        }
    }
    r.l[8] = (int32_t)((uint32_t)acc << 3) >> 3;  // |r| < 2^256: limb 8 is the low 29 bits of the column, sign-extended
    PLONK_CHAIN_END(r.l[8]);
#ifdef PLONK_EMU
    {
        long double mm = 0;
        for (int i = 8; i >= 0; i--) mm = mm * 536870912.0L + (long double)fp29_mod_limb<P>(i);
        const long double v = fpl_dbg_value(r), A = fabsl(fpl_dbg_value(a)) / ldexpl(1.0L, 261);  // |a| / R
        FPL_CHECK(v > -(1.0L + A) * mm - 1.0L && v < (2.0L + A) * mm + 1.0L, "fpl_mul_shoup: result outside (-(1 + |a|/R) m, (2 + |a|/R) m)");
        FPL_CHECK(v > -1.8L * mm && v < 2.8L * mm, "fpl_mul_shoup: result outside (-1.8 m, 2.8 m)");
    }
#endif
    return r;
}

// normalised value within (0, 2m), limbs non-negative (fpl_reduce_small<P, 1>'s result) -> canonical packed element
// canonical = false leaves the value as it is, in (0, 2m): a packed but redundant residue (what the column pass of a
// two-pass NTT hands to the row pass, which unpacks it again)
template <class P> PLONK_HD Fp<P> fpl_pack_positive(const FpL<P>& a, bool canonical = true) {
    FPL_CHECK(a.l[8] >= 0, "fpl_pack_positive: negative value");
    uint32_t u[9];
#pragma unroll
    for (int i = 0; i < 9; i++) u[i] = (uint32_t)a.l[i];
    Fp<P> out;
    fp29_pack(u, out.v);
    if (canonical) fp_reduce_once<P>(out.v);
    return out;
}
