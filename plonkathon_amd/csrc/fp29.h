// fp29.h — Montgomery multiplication on 9 x 29-bit limbs with lazy carries (product scanning).
//
// Motivation (profiles/r01_ubench.json): on gfx950 v_mad_u64_u32 issues at ~5 cycles per wave64 and
// plain VALU ops at ~2; the 32-bit CIOS in fp.h spends more cycles on carry adds and register-pair
// moves (430 ops) than on its 136 multiply-adds.  With 29-bit limbs every partial product is
// < 2^58, a column holds at most 18 of them plus a carry (< 2^63), so a column is a pure chain of
// v_mad_u64_u32 into one 64-bit accumulator: no carry flags, no zero-extension moves.
//
// Representation: the packed 8 x u32 form stays the storage format; values are Montgomery
// residues with R = 2^261 (9 * 29).  fp29_mul(a, b) = a * b * 2^-261 mod m, inputs/outputs < m.
#pragma once
#include "fp.h"

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
        // word w covers bits [32w, 32w+32): limbs floor(32w/29) .. floor((32w+31)/29)
        const int first = (32 * w) / 29, off = 32 * w - 29 * first;  // bit offset inside limb `first`
        uint32_t x = l[first] >> off;
        int have = 29 - off;
        if (first + 1 < 9) x |= l[first + 1] << have;
        have += 29;
        if (have < 32 && first + 2 < 9) x |= l[first + 2] << have;
        v[w] = x;
    }
}

template <class P> PLONK_HD Fp<P> fp29_mul(const Fp<P>& a, const Fp<P>& b) {
    uint32_t x[9], y[9], q[9], r[9];
    fp29_unpack(a.v, x);
    fp29_unpack(b.v, y);
    const uint32_t ninv = P::NINV & FP29_MASK;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)x[i] * y[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)q[i] * fp29_mod_limb<P>(k - i);
        q[k] = ((uint32_t)acc * ninv) & FP29_MASK;
        acc += (uint64_t)q[k] * fp29_mod_limb<P>(0);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (uint64_t)x[i] * y[k - i];
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (uint64_t)q[i] * fp29_mod_limb<P>(k - i);
        r[k - 9] = (uint32_t)acc & FP29_MASK;
        acc >>= 29;
    }
    r[8] = (uint32_t)acc;
    Fp<P> out;
    fp29_pack(r, out.v);
    fp_reduce_once<P>(out.v);
    return out;
}
