// transcript.h — Merlin v1.0 transcript (STROBE-128 over Keccak-f[1600]) and plonkathon's
// Fiat-Shamir layer, written host+device so the same code serves the host C-ABI
// (plonk_transcript_*) and the batched on-device transcript kernel.
//
// Replaces `merlin.MerlinTranscript` (third-party, un-vendored: pyproject.toml:12) as used by
// /root/reference/transcript.py:58-75: append_message(label, msg), challenge_bytes(label, n), and
// get_and_append_challenge (255 PRF bytes -> big-endian int -> mod r, retry on zero, re-append).
// Restated from the published Merlin / STROBE v1.0.2 specification; conformance is pinned by the
// merlin crate's public test vector and by the reference's golden proof (tests).
#pragma once
#include "fp.h"

#define STROBE_R 166
#define STROBE_FLAG_I 1
#define STROBE_FLAG_A 2
#define STROBE_FLAG_C 4
#define STROBE_FLAG_T 8
#define STROBE_FLAG_M 16
#define STROBE_FLAG_K 32

struct MerlinState {
    uint64_t st[25];  // Keccak state, lane (x,y) = st[x + 5y], bytes little-endian within a lane
    uint32_t pos, pos_begin, cur_flags, pad_;
};

PLONK_HD uint64_t keccak_rotl(uint64_t x, unsigned n) { return n ? ((x << n) | (x >> (64 - n))) : x; }

PLONK_HD uint64_t keccak_rc(int round) {
    constexpr uint64_t rc[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL,
        0x000000000000808BULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
        0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
        0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    return rc[round];
}

// Out of line (one copy per kernel image) but fully unrolled inside: the 25 lanes stay in registers for
// all 24 rounds (constant indices everywhere), instead of living in scratch memory.
PLONK_HD_NOINLINE void keccak_f1600(uint64_t st[25]) {
    constexpr unsigned rot[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    uint64_t a[25];
#pragma unroll
    for (int i = 0; i < 25; i++) a[i] = st[i];
    for (int round = 0; round < 24; round++) {
        uint64_t c[5], d[5], b[25];
#pragma unroll
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
#pragma unroll
        for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ keccak_rotl(c[(x + 1) % 5], 1);
#pragma unroll
        for (int i = 0; i < 25; i++) a[i] ^= d[i % 5];
#pragma unroll
        for (int x = 0; x < 5; x++)
#pragma unroll
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = keccak_rotl(a[x + 5 * y], rot[x + 5 * y]);
#pragma unroll
        for (int y = 0; y < 5; y++)
#pragma unroll
            for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= keccak_rc(round);
    }
#pragma unroll
    for (int i = 0; i < 25; i++) st[i] = a[i];
}

PLONK_HD uint8_t strobe_get(const MerlinState& s, unsigned i) { return (uint8_t)(s.st[i >> 3] >> (8 * (i & 7))); }
PLONK_HD void strobe_xor(MerlinState& s, unsigned i, uint8_t b) { s.st[i >> 3] ^= (uint64_t)b << (8 * (i & 7)); }
PLONK_HD void strobe_set0(MerlinState& s, unsigned i) { s.st[i >> 3] &= ~((uint64_t)0xff << (8 * (i & 7))); }

PLONK_HD void strobe_run_f(MerlinState& s) {
    strobe_xor(s, s.pos, (uint8_t)s.pos_begin);
    strobe_xor(s, s.pos + 1, 0x04);
    strobe_xor(s, STROBE_R + 1, 0x80);
    keccak_f1600(s.st);
    s.pos = 0;
    s.pos_begin = 0;
}

PLONK_HD void strobe_absorb(MerlinState& s, const uint8_t* data, size_t n) {
    for (size_t i = 0; i < n; i++) {
        strobe_xor(s, s.pos, data[i]);
        if (++s.pos == STROBE_R) strobe_run_f(s);
    }
}

PLONK_HD void strobe_squeeze(MerlinState& s, uint8_t* out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        out[i] = strobe_get(s, s.pos);
        strobe_set0(s, s.pos);
        if (++s.pos == STROBE_R) strobe_run_f(s);
    }
}

PLONK_HD void strobe_begin_op(MerlinState& s, uint32_t flags, bool more) {
    if (more) return;  // caller guarantees cur_flags == flags
    uint8_t hdr[2] = {(uint8_t)s.pos_begin, (uint8_t)flags};
    s.pos_begin = s.pos + 1;
    s.cur_flags = flags;
    strobe_absorb(s, hdr, 2);
    if ((flags & (STROBE_FLAG_C | STROBE_FLAG_K)) && s.pos != 0) strobe_run_f(s);
}

PLONK_HD void strobe_meta_ad(MerlinState& s, const uint8_t* d, size_t n, bool more) {
    strobe_begin_op(s, STROBE_FLAG_M | STROBE_FLAG_A, more);
    strobe_absorb(s, d, n);
}
PLONK_HD void strobe_ad(MerlinState& s, const uint8_t* d, size_t n, bool more) {
    strobe_begin_op(s, STROBE_FLAG_A, more);
    strobe_absorb(s, d, n);
}
PLONK_HD void strobe_prf(MerlinState& s, uint8_t* out, size_t n, bool more) {
    strobe_begin_op(s, STROBE_FLAG_I | STROBE_FLAG_A | STROBE_FLAG_C, more);
    strobe_squeeze(s, out, n);
}

PLONK_HD void merlin_append_message(MerlinState& s, const uint8_t* label, size_t label_len, const uint8_t* msg, size_t msg_len) {
    uint8_t len_le[4] = {(uint8_t)msg_len, (uint8_t)(msg_len >> 8), (uint8_t)(msg_len >> 16), (uint8_t)(msg_len >> 24)};
    strobe_meta_ad(s, label, label_len, false);
    strobe_meta_ad(s, len_le, 4, true);
    strobe_ad(s, msg, msg_len, false);
}

PLONK_HD void merlin_challenge_bytes(MerlinState& s, const uint8_t* label, size_t label_len, uint8_t* out, size_t n) {
    uint8_t len_le[4] = {(uint8_t)n, (uint8_t)(n >> 8), (uint8_t)(n >> 16), (uint8_t)(n >> 24)};
    strobe_meta_ad(s, label, label_len, false);
    strobe_meta_ad(s, len_le, 4, true);
    strobe_prf(s, out, n, false);
}

PLONK_HD void merlin_init(MerlinState& s, const uint8_t* label, size_t label_len) {
    for (int i = 0; i < 25; i++) s.st[i] = 0;
    const uint8_t init[18] = {1, STROBE_R + 2, 1, 0, 1, 96, 'S', 'T', 'R', 'O', 'B', 'E', 'v', '1', '.', '0', '.', '2'};
    for (unsigned i = 0; i < 18; i++) strobe_xor(s, i, init[i]);
    keccak_f1600(s.st);
    s.pos = 0;
    s.pos_begin = 0;
    s.cur_flags = 0;
    s.pad_ = 0;
    const uint8_t proto[11] = {'M', 'e', 'r', 'l', 'i', 'n', ' ', 'v', '1', '.', '0'};
    strobe_meta_ad(s, proto, 11, false);
    const uint8_t dom[7] = {'d', 'o', 'm', '-', 's', 'e', 'p'};
    merlin_append_message(s, dom, 7, label, label_len);
}

// big-endian bytes -> Fr (Montgomery), reducing mod r.  Horner in base 2^256: each 32-byte chunk is an
// integer < 2^256 < 6r, which one Montgomery multiplication by R^2 maps to its residue (fp_mul only needs
// a*b < R*m), so a 255-byte challenge costs 16 multiplications.
PLONK_HD Fr fr_from_be_bytes_mod(const uint8_t* b, size_t n) {
    Fr two256 = fp_zero<FrParams>();
    two256.v[0] = 1;
    two256 = fp_to_mont(two256);                       // R mod r ... times 2^256 below
    {   // 2^256 in Montgomery form = to_mont(2^256 mod r); build it as (2^128)^2
        Fr t = fp_zero<FrParams>();
        t.v[4] = 1;                                    // 2^128, canonical
        t = fp_to_mont(t);
        two256 = fp_mul(t, t);
    }
    Fr acc = fp_zero<FrParams>();
    size_t i = 0;
    while (i < n) {
        const size_t take = (i == 0 && (n % 32)) ? (n % 32) : 32;
        Fr chunk = fp_zero<FrParams>();                // little-endian limbs of the big-endian chunk
        for (size_t k = 0; k < take; k++) {
            const size_t pos = take - 1 - k;           // byte significance within the chunk
            chunk.v[pos >> 2] |= (uint32_t)b[i + k] << (8 * (pos & 3));
        }
        i += take;
        acc = fp_add(fp_mul(acc, two256), fp_to_mont(chunk));
    }
    return acc;
}

// transcript.py:69-75 — returns the challenge in Montgomery form
PLONK_HD Fr plonk_get_and_append_challenge(MerlinState& s, const uint8_t* label, size_t label_len) {
    uint8_t buf[255];
    for (;;) {
        merlin_challenge_bytes(s, label, label_len, buf, 255);
        Fr f = fr_from_be_bytes_mod(buf, 255);
        if (!fp_is_zero(f)) {
            merlin_append_message(s, label, label_len, buf, 255);
            return f;
        }
    }
}

// 32-byte big-endian encoding of a canonical field element given as LE limbs (transcript.py:62-67)
PLONK_HD void limbs_to_be32(const uint32_t v[8], uint8_t out[32]) {
    for (int i = 0; i < 8; i++) {
        uint32_t w = v[7 - i];
        out[4 * i] = (uint8_t)(w >> 24);
        out[4 * i + 1] = (uint8_t)(w >> 16);
        out[4 * i + 2] = (uint8_t)(w >> 8);
        out[4 * i + 3] = (uint8_t)w;
    }
}
