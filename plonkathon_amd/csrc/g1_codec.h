// g1_codec.h — the compressed G1 encoding (SURVEY.md 0 / north_star "compressed G1 bytes").
//
// The reference has ONE G1 byte encoding: x then y, each as a 32-byte big-endian integer (append_point,
// /root/reference/transcript.py:62-67).  The compressed form is derived from it: the same 32 big-endian bytes of x, with
// the two bits that a 254-bit modulus leaves free at the top of byte 0 carrying what y contributed:
//     10  y is the smaller of the two roots of x^3 + 3   (y <= (p - 1) / 2)
//     11  y is the larger root                            (y >  (p - 1) / 2)
//     01  the point at infinity (py_ecc's None); x = 0
//     00  never produced (rejected by the decoder)
// — the flag layout gnark-crypto uses for BN254, big-endian like the reference's own bytes.  A proof is then
// 9 x 32 + 6 x 32 = 480 bytes: the nine commitments of Proof.flatten() order (prover.py:18-35) compressed, then the six
// evaluations as 32-byte big-endian scalars (append_scalar's form, transcript.py:62-63).
#pragma once
#include "fp.h"

#define PLONK_G1C_SMALLEST 0x80u
#define PLONK_G1C_LARGEST 0xC0u
#define PLONK_G1C_INFINITY 0x40u

// canonical 8 x u32 little-endian words: a > (p - 1) / 2  <=>  2 a >= p + 1  <=>  2 a > p
template <class P> PLONK_HD bool g1c_is_larger_half(const uint32_t a[8]) {
    uint32_t carry = 0;
    bool gt = false, eq = true;  // compare 2a (257 bits) with p from the top word down
    uint32_t d[9];
    for (int i = 0; i < 8; i++) {
        d[i] = (a[i] << 1) | carry;
        carry = a[i] >> 31;
    }
    d[8] = carry;
    if (d[8]) return true;
    for (int i = 7; i >= 0 && eq; i--) {
        if (d[i] != P::mod(i)) {
            gt = d[i] > P::mod(i);
            eq = false;
        }
    }
    return gt;  // 2a == p is impossible (p odd)
}

// x, y canonical words (both zero = infinity) -> 32 bytes
template <class P> PLONK_HD void g1c_compress(const uint32_t x[8], const uint32_t y[8], uint8_t out[32]) {
    bool inf = true;
    for (int i = 0; i < 8; i++) inf = inf && x[i] == 0 && y[i] == 0;
    for (int i = 0; i < 8; i++) {
        const uint32_t w = inf ? 0u : x[7 - i];
        out[4 * i] = (uint8_t)(w >> 24);
        out[4 * i + 1] = (uint8_t)(w >> 16);
        out[4 * i + 2] = (uint8_t)(w >> 8);
        out[4 * i + 3] = (uint8_t)w;
    }
    out[0] |= inf ? PLONK_G1C_INFINITY : (g1c_is_larger_half<P>(y) ? PLONK_G1C_LARGEST : PLONK_G1C_SMALLEST);
}

// canonical Fr / Fq words -> 32 big-endian bytes (append_scalar's form)
PLONK_HD void g1c_be32(const uint32_t v[8], uint8_t out[32]) {
    for (int i = 0; i < 8; i++) {
        const uint32_t w = v[7 - i];
        out[4 * i] = (uint8_t)(w >> 24);
        out[4 * i + 1] = (uint8_t)(w >> 16);
        out[4 * i + 2] = (uint8_t)(w >> 8);
        out[4 * i + 3] = (uint8_t)w;
    }
}
