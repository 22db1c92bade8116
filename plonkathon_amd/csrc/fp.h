// fp.h — 254-bit prime-field arithmetic in Montgomery form, R = 2^261.
//
// Replaces py_ecc's `FQ.__add__/__sub__/__mul__/__truediv__/__pow__` as used through `Scalar`
// (/root/reference/curve.py:10-11) and `b.FQ` (the G1 coordinates in curve.py:38-44).  Every
// device-resident field element is a Montgomery residue, canonical (< m), stored as 8 x u32
// little-endian limbs (32 bytes).  A snarkjs .ptau stores coordinates the same way with R = 2^256
// (setup.py:29-41), so SRS bytes need only a x2^5 on upload (five modular doublings).
//
// Multiplication: product scanning on 9 x 29-bit limbs with lazy carries.  Measured on gfx950
// (profiles/r01_ubench.json) v_mad_u64_u32 issues at ~5 cycles per wave64 and plain VALU ops at ~2, and
// clang pads carry-flag chains with s_nop; a 32-bit-limb CIOS therefore spends more cycles on carry
// adds and register-pair moves (430 ops) than on its 136 multiply-adds.  With 29-bit limbs every
// partial product is < 2^58 and a column holds at most 18 of them plus a carry (< 2^63), so a column is
// a pure chain of v_mad_u64_u32 into one 64-bit accumulator — no carry flags, no zero-extension moves:
// 161 mad + 9 mul_lo + ~160 cheap ops instead of 566.  Not a dense contraction: no MFMA.
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

// 32-bit add/sub with carry.  clang lowers the builtins to v_add_co / v_addc_co chains (one VALU op per
// limb); the portable u64 form compiles to v_lshl_add_u64 plus register-pair moves, ~3x the instructions.
#if defined(__clang__)
PLONK_HD uint32_t fp_adc(uint32_t a, uint32_t b, uint32_t& c) { unsigned co; uint32_t r = __builtin_addc(a, b, c, &co); c = co; return r; }
PLONK_HD uint32_t fp_sbb(uint32_t a, uint32_t b, uint32_t& br) { unsigned bo; uint32_t r = __builtin_subc(a, b, br, &bo); br = bo; return r; }
#else
PLONK_HD uint32_t fp_adc(uint32_t a, uint32_t b, uint32_t& c) { uint64_t x = (uint64_t)a + b + c; c = (uint32_t)(x >> 32); return (uint32_t)x; }
PLONK_HD uint32_t fp_sbb(uint32_t a, uint32_t b, uint32_t& br) { uint64_t x = (uint64_t)a - b - br; br = (uint32_t)(x >> 32) & 1; return (uint32_t)x; }
#endif

// r = t - m if t >= m else t      (t < 2m)
template <class P> PLONK_HD void fp_reduce_once(uint32_t t[8]) {
    uint32_t d[8], br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d[i] = fp_sbb(t[i], P::mod(i), br);
#pragma unroll
    for (int i = 0; i < 8; i++) t[i] = br ? t[i] : d[i];
}

template <class P> PLONK_HD Fp<P> fp_add(const Fp<P>& a, const Fp<P>& b) {
    Fp<P> r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = fp_adc(a.v[i], b.v[i], c);
    fp_reduce_once<P>(r.v);
    return r;
}

template <class P> PLONK_HD Fp<P> fp_sub(const Fp<P>& a, const Fp<P>& b) {
    Fp<P> r;
    uint32_t d[8], br = 0, c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) d[i] = fp_sbb(a.v[i], b.v[i], br);
    const uint32_t mask = 0u - br;  // add m back when the difference went negative
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = fp_adc(d[i], P::mod(i) & mask, c);
    return r;
}

template <class P> PLONK_HD Fp<P> fp_neg(const Fp<P>& a) {
    uint32_t nz = 0, br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) nz |= a.v[i];
    const uint32_t mask = nz ? 0xffffffffu : 0u;  // -0 == 0
    Fp<P> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = fp_sbb(P::mod(i) & mask, a.v[i], br);
    return r;
}

template <class P> PLONK_HD Fp<P> fp_dbl(const Fp<P>& a) { return fp_add(a, a); }

#define FP29_MASK 0x1fffffffu

template <class P> PLONK_HD constexpr uint32_t fp29_mod_limb(int i) {
    const int bit = 29 * i, w = bit >> 5, s = bit & 31;
    const uint64_t lo = P::mod(w);
    const uint64_t hi = (w + 1 < 8) ? P::mod(w + 1) : 0;
    return (uint32_t)(((lo | (hi << 32)) >> s) & FP29_MASK);
}

// 8 x u32 -> 9 x 29-bit limbs
PLONK_HD void fp29_unpack(const uint32_t v[8], uint32_t l[9]) {
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, w = bit >> 5, s = bit & 31;
        uint32_t x = v[w] >> s;
        if (s > 3 && w + 1 < 8) x |= v[w + 1] << (32 - s);
        l[i] = x & FP29_MASK;
    }
}

// 9 normalised 29-bit limbs (value < 2^256) -> 8 x u32
PLONK_HD void fp29_pack(const uint32_t l[9], uint32_t v[8]) {
#pragma unroll
    for (int w = 0; w < 8; w++) {
        const int first = (32 * w) / 29, off = 32 * w - 29 * first;  // bit offset inside limb `first`
        uint32_t x = l[first] >> off;
        int have = 29 - off;
        if (first + 1 < 9) x |= l[first + 1] << have;
        have += 29;
        if (have < 32 && first + 2 < 9) x |= l[first + 2] << have;
        v[w] = x;
    }
}

// Montgomery product a*b*2^-261 mod m.  Inputs < m, output < m.
template <class P> PLONK_FP_CALL Fp<P> fp_mul(const Fp<P> a, const Fp<P> b) {
    uint32_t x[9], y[9], r[9];
    fp29_unpack(a.v, x);
    fp29_unpack(b.v, y);
    const uint32_t ninv = P::NINV & FP29_MASK;
    uint32_t q[9];
    uint64_t acc = 0;
    PLONK_CHAIN_BEGIN();
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) {
            acc += (uint64_t)x[i] * y[k - i];
            PLONK_CHAIN(acc);
        }
#pragma unroll
        for (int i = 0; i < k; i++) {
            acc += (uint64_t)q[i] * fp29_mod_limb<P>(k - i);
            PLONK_CHAIN(acc);
        }
        q[k] = ((uint32_t)acc * ninv) & FP29_MASK;
        acc += (uint64_t)q[k] * fp29_mod_limb<P>(0);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) {
            acc += (uint64_t)x[i] * y[k - i];
            PLONK_CHAIN(acc);
        }
#pragma unroll
        for (int i = k - 8; i < 9; i++) {
            acc += (uint64_t)q[i] * fp29_mod_limb<P>(k - i);
            PLONK_CHAIN(acc);
        }
        r[k - 9] = (uint32_t)acc & FP29_MASK;
        acc >>= 29;
    }
    r[8] = (uint32_t)acc;
    PLONK_CHAIN_END(r[8]);
    Fp<P> out;
    fp29_pack(r, out.v);
    fp_reduce_once<P>(out.v);
    return out;
}

// Squaring: the 81 cross products collapse to 45 (off-diagonal terms doubled).
template <class P> PLONK_FP_CALL Fp<P> fp_sqr(const Fp<P> a) {
    uint32_t x[9], x2[9], r[9];
    fp29_unpack(a.v, x);
#pragma unroll
    for (int i = 0; i < 9; i++) x2[i] = x[i] << 1;  // < 2^30: doubled partial products stay < 2^59
    const uint32_t ninv = P::NINV & FP29_MASK;
    uint32_t q[9];
    uint64_t acc = 0;
    PLONK_CHAIN_BEGIN();
#pragma unroll
    for (int k = 0; k < 17; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) {
            const int j = k - i;
            if (i > 8 || j > 8 || i > j) continue;
            acc += (i == j) ? (uint64_t)x[i] * x[i] : (uint64_t)x2[i] * x[j];
            PLONK_CHAIN(acc);
        }
        if (k < 9) {
#pragma unroll
            for (int i = 0; i < k; i++) {
            acc += (uint64_t)q[i] * fp29_mod_limb<P>(k - i);
            PLONK_CHAIN(acc);
        }
            q[k] = ((uint32_t)acc * ninv) & FP29_MASK;
            acc += (uint64_t)q[k] * fp29_mod_limb<P>(0);
        } else {
#pragma unroll
            for (int i = k - 8; i < 9; i++) {
            acc += (uint64_t)q[i] * fp29_mod_limb<P>(k - i);
            PLONK_CHAIN(acc);
        }
            r[k - 9] = (uint32_t)acc & FP29_MASK;
        }
        acc >>= 29;
    }
    r[8] = (uint32_t)acc;
    PLONK_CHAIN_END(r[8]);
    Fp<P> out;
    fp29_pack(r, out.v);
    fp_reduce_once<P>(out.v);
    return out;
}

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

// Fermat inverse a^(m-2) (~380 multiplications); kept as the cross-check for fp_inv in the tests.
template <class P> PLONK_HD Fp<P> fp_inv_fermat(const Fp<P>& a) {
    uint32_t e[8];
#pragma unroll
    for (int i = 0; i < 8; i++) e[i] = P::mod_minus_2(i);
    return fp_pow_limbs(a, e);
}

// f(0) .. f(N-1) with compile-time arguments: an array of field elements must never be indexed by a run-time value, or it
// moves from VGPRs to scratch memory — and `#pragma unroll` gives up on loops whose bodies hold two field multiplications
// (the wave NTT kernels' element arrays, the chunks of the batch inversions)
template <unsigned J> struct WaveIdx { static constexpr unsigned value = J; };
template <unsigned N, class F> PLONK_HD void wave_for(F f) {
    if constexpr (N > 0) {
        wave_for<N - 1>(f);
        f(WaveIdx<N - 1>{});
    }
}
template <unsigned N, class F> PLONK_HD void wave_for_down(F f) {  // f(N-1) .. f(0)
    if constexpr (N > 0) {
        f(WaveIdx<N - 1>{});
        wave_for_down<N - 1>(f);
    }
}

// ---- inversion by Bernstein-Yang division steps -------------------------------------------------------
// Values are 9 signed limbs of 30 bits.  Thirty division steps are run on the low limbs of (f, g) only and
// recorded as a 2x2 integer matrix t with [f'; g'] = t [f; g] / 2^30; the matrix is then applied once to
// the full (f, g) and, modulo m, to the pair (d, e) that tracks d*x = f, e*x = g (mod m).  When g reaches
// 0, f = +-1 and +-d is the inverse.  ~19 rounds of ~700 instructions against ~110 000 for Fermat: this is
// what the latency-bound kernels (affine conversion of commitments, grand product, batch inversion) wait
// on.  Variable time (public data only).  0 -> 0, which is exactly py_ecc's `x / 0 == 0`.
#define FP30_MASK 0x3fffffff

template <class P> PLONK_HD constexpr int32_t fp30_mod_limb(int i) {
    const int bit = 30 * i, j = bit >> 5, sh = bit & 31;
    const uint64_t two = (uint64_t)P::mod(j) | (j + 1 < 8 ? (uint64_t)P::mod(j + 1) << 32 : 0);
    return (int32_t)((two >> sh) & FP30_MASK);
}

struct Fp30Matrix { int32_t u, v, q, r; };

// 30 division steps on the low words; delta carries over between rounds
PLONK_HD Fp30Matrix fp30_divsteps(int32_t& delta, uint32_t f, uint32_t g) {
    uint32_t u = 1, v = 0, q = 0, r = 1;
    int32_t d = delta;
    for (int i = 0; i < 30; i++) {
        const uint32_t odd = 0u - (g & 1u);
        const uint32_t sw = (d > 0 ? ~0u : 0u) & odd;  // delta > 0 and g odd: (f, g) <- (g, -f)
        const uint32_t tf = f, tu = u, tv = v;
        f = sw ? g : f;
        g = sw ? 0u - tf : g;
        u = sw ? q : u;
        v = sw ? r : v;
        q = sw ? 0u - tu : q;
        r = sw ? 0u - tv : r;
        d = sw ? -d : d;
        g += f & odd;  // now even
        q += u & odd;
        r += v & odd;
        g >>= 1;
        u <<= 1;  // the f row absorbs the halving so the matrix stays integral
        v <<= 1;
        d += 1;
    }
    delta = d;
    return Fp30Matrix{(int32_t)u, (int32_t)v, (int32_t)q, (int32_t)r};
}

// (f, g) <- t (f, g) / 2^30, exact
PLONK_HD void fp30_update_fg(int32_t f[9], int32_t g[9], const Fp30Matrix& t) {
    int64_t cf = (int64_t)t.u * f[0] + (int64_t)t.v * g[0];
    int64_t cg = (int64_t)t.q * f[0] + (int64_t)t.r * g[0];
    cf >>= 30;
    cg >>= 30;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        cf += (int64_t)t.u * f[i] + (int64_t)t.v * g[i];
        cg += (int64_t)t.q * f[i] + (int64_t)t.r * g[i];
        f[i - 1] = (int32_t)cf & FP30_MASK;
        g[i - 1] = (int32_t)cg & FP30_MASK;
        cf >>= 30;
        cg >>= 30;
    }
    f[8] = (int32_t)cf;
    g[8] = (int32_t)cg;
}

// (d, e) <- t (d, e) / 2^30 mod m, keeping both in (-2m, m)
template <class P> PLONK_HD void fp30_update_de(int32_t d[9], int32_t e[9], const Fp30Matrix& t) {
    const uint32_t minv = (0u - P::NINV) & FP30_MASK;  // m^-1 mod 2^30
    const int32_t sd = d[8] >> 31, se = e[8] >> 31;
    int32_t md = (t.u & sd) + (t.v & se);  // + m for a negative d / e
    int32_t me = (t.q & sd) + (t.r & se);
    int64_t cd = (int64_t)t.u * d[0] + (int64_t)t.v * e[0];
    int64_t ce = (int64_t)t.q * d[0] + (int64_t)t.r * e[0];
    md -= (int32_t)((minv * (uint32_t)cd + (uint32_t)md) & FP30_MASK);  // low 30 bits of cd + md*m become 0
    me -= (int32_t)((minv * (uint32_t)ce + (uint32_t)me) & FP30_MASK);
    cd += (int64_t)fp30_mod_limb<P>(0) * md;
    ce += (int64_t)fp30_mod_limb<P>(0) * me;
    cd >>= 30;
    ce >>= 30;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        cd += (int64_t)t.u * d[i] + (int64_t)t.v * e[i] + (int64_t)fp30_mod_limb<P>(i) * md;
        ce += (int64_t)t.q * d[i] + (int64_t)t.r * e[i] + (int64_t)fp30_mod_limb<P>(i) * me;
        d[i - 1] = (int32_t)cd & FP30_MASK;
        e[i - 1] = (int32_t)ce & FP30_MASK;
        cd >>= 30;
        ce >>= 30;
    }
    d[8] = (int32_t)cd;
    e[8] = (int32_t)ce;
}

// a^-1 in Montgomery form (input a R, output a^-1 R)
template <class P> PLONK_HD Fp<P> fp_inv(const Fp<P>& a) {
    int32_t f[9], g[9], d[9], e[9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 30 * i, j = bit >> 5, sh = bit & 31;
        const uint64_t two = (uint64_t)a.v[j] | (j + 1 < 8 ? (uint64_t)a.v[j + 1] << 32 : 0);
        g[i] = (int32_t)((two >> sh) & FP30_MASK);
        f[i] = fp30_mod_limb<P>(i);
        d[i] = 0;
        e[i] = 0;
    }
    e[0] = 1;
    int32_t delta = 1;
    for (int round = 0; round < 26; round++) {  // <= 741 steps for 256-bit inputs (Bernstein-Yang, Thm 11.2)
        uint32_t nz = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) nz |= (uint32_t)g[i];
        if (!nz) break;
        const Fp30Matrix t = fp30_divsteps(delta, (uint32_t)f[0], (uint32_t)g[0]);
        fp30_update_fg(f, g, t);
        fp30_update_de<P>(d, e, t);
    }
    // (a R)^-1 = sign(f) d, in (-2m, 2m): add 2m, carry-propagate, reduce
    const bool neg = f[8] < 0;
    uint32_t l[9];
    int64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        acc += (int64_t)(neg ? -d[i] : d[i]) + 2 * (int64_t)fp30_mod_limb<P>(i);
        l[i] = (uint32_t)acc & FP30_MASK;
        acc >>= 30;
    }
    if constexpr (P::mod(7) >= 0x40000000u) {  // 4m >= 2^256 (BLS12-381 Fr): bring (0, 4m) into (0, 2m) before packing to 256 bits
        uint32_t t[9];
        int64_t bw = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            bw += (int64_t)l[i] - 2 * (int64_t)fp30_mod_limb<P>(i);
            t[i] = (uint32_t)bw & FP30_MASK;
            bw >>= 30;
        }
#pragma unroll
        for (int i = 0; i < 9; i++) l[i] = bw < 0 ? l[i] : t[i];
    }
    Fp<P> y;
#pragma unroll
    for (int j = 0; j < 8; j++) {  // 9 x 30 bits -> 8 x 32 bits (value < 4m < 2^256, or < 2m after the step above)
        const int bit = 32 * j, i = bit / 30, sh = bit % 30;
        uint64_t w = (uint64_t)l[i] >> sh;
        w |= (uint64_t)l[i + 1] << (30 - sh);
        if (i + 2 < 9) w |= (uint64_t)l[i + 2] << (60 - sh);
        y.v[j] = (uint32_t)w;
    }
    fp_reduce_once<P>(y.v);
    fp_reduce_once<P>(y.v);
    fp_reduce_once<P>(y.v);
    Fp<P> r3;
#pragma unroll
    for (int i = 0; i < 8; i++) r3.v[i] = P::r3(i);
    return fp_mul(y, r3);  // (a R)^-1 R^3 / R = a^-1 R
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
