// prover.hip — the batched, GPU-resident five-round PLONK prover (plonk_prover_*).
//
// Reference behaviour replaced: Prover.prove / round_1..round_5 (/root/reference/prover.py:51-306,
// spec in SURVEY.md §3.2) for B independent proofs of one circuit run in lock-step, including the
// Fiat-Shamir transcript (transcript.py:77-123), which runs on the device (32 lanes per proof) so
// that a whole batch is one uninterrupted stream of kernel launches with no host round trip.
//
// Every committed polynomial is uniquely determined by (circuit, witness, challenges) — the
// reference adds no blinding (README.md:31-32) — so the schedule is free to differ from the
// reference's as long as the polynomials are the same:
//   * the coset offset used for the quotient is a FIXED generator g = 5 instead of the per-proof
//     `fft_cofactor` challenge (which is still drawn, to keep the transcript identical); the
//     circuit's selector / permutation polynomials are therefore extended ONCE per circuit;
//   * coset extensions start from the coefficient forms the commitments already produced;
//   * T1..T3 are committed straight from the quotient's coefficient slices (commit(fft(c)) = MSM(c));
//   * round 4 evaluates coefficient forms by a blocked Horner; round 5 builds the opening
//     polynomials in coefficient form (linear combination + synthetic division by X - z), so the
//     reference's ~15 further coset extensions and its 4n-point divisions never happen.
#include <string.h>

#include "plonk_internal.h"
#include "transcript.h"
#include "g1_codec.h"

#define NEVAL 7  // a, b, c, s1, s2, z_shifted, PI(zeta)
#define PI_SPARSE_MAX 8

struct ProofState {
    Fr beta, gamma, alpha, fft_cofactor, zeta, v;
    Fr evals[NEVAL];
    MerlinState transcript;
    uint32_t error;  // 1: a commitment was the identity (the reference's append_point(None) raises)
    uint32_t pad_[3];
};

struct ChallengeConsts { Fr c[8]; };  // c[j] = 2^(256 j) R^2 mod r (transcript_kernel)

enum { FX_QM = 0, FX_QL, FX_QR, FX_QO, FX_QC, FX_S1, FX_S2, FX_S3, FX_COUNT };
#define QCOSETS 3  // cosets of size n the lock-step prover evaluates the quotient on (deg t < 3n)

struct plonk_prover {
    plonk_ctx* ctx;
    plonk_srs* srs;
    unsigned log_n;
    size_t n, n_public;
    Fr g;                // fixed coset offset (Montgomery)
    // The quotient has degree < 3n, so THREE cosets of the n-th roots of unity determine it: x = g mu^r w^j, r < 3 (mu = the
    // 4n-th root of unity of prover.py:160), "coset-major" [r][j].  Every coset form below is [3][n] in that order.
    Fr* fixed_lag;       // [8][n]   Lagrange values
    Fr* fixed_coef;      // [8][n]   coefficient forms
    Fr* fixed_big;       // [8][3][n]  the circuit polynomials on the three cosets
    Fr* l0_big;          // [3][n]
    Fr* x_big;           // [3][n]   the points g mu^r w^j
    Fr* g_pow;           // [3][n]   (g mu^r)^i: the load-side scaling of the size-n transform that evaluates on coset r
    Fr* ginv_pow;        // [3][n]   (g mu^r)^-i / 2n: the store-side scaling of the inverse transform of coset r (its 1/n folded in)
    const Fr* roots;     // [n]      w^i (owned by ctx)
    Fr zh_inv[QCOSETS];  // 1 / (g^n * i^r - 1): Z_H is constant on a coset
    Fr comb_i, comb_g1, comb_g2;  // quotient_combine_kernel's constants: i = mu^n, 1 / g^n, 1 / g^2n
    // Public inputs are the only non-zero entries of the PI column (prover.py:57-62): with few of them PI's
    // coefficient and coset forms are cheaper from the Lagrange basis directly than through two transforms.
    bool sparse_pi;      // n_public <= PI_SPARSE_MAX
    Fr* li_big;          // [n_public][4n]  L_i on the coset: (w^i / n) Z_H(x_k) / (x_k - w^i)
    const Fr* roots_inv; // [n]             w^-i (owned by ctx)
    Fr* pub;             // [B][n_public]   public inputs of the resident batch (Montgomery)
    // per-batch buffers (capacity cap_b proofs)
    size_t cap_b;
    Fr *wit_lag;   // [4][B][n]  A, B, C, PI   Lagrange
    Fr *z_lag;     // [B][n]
    Fr *coef;      // [5][B][n]  Ac, Bc, Cc, PIc, Zc   (coefficient forms; Z last so rounds 1 and 2 fill it in order)
    Fr *big;       // [5][B][3][n] A, B, C, PI, Z on the three cosets
    Fr *quot;      // [B][4n]    quotient evaluations on the three cosets, then its 3n coefficients (in place; the last n unused)
    Fr *num, *den; // [B][n] scratch (round 2), reused as W_z numerator
    Fr *wz;        // [2][B][n]  W_z, W_zw coefficient forms
    struct LinWeights* lin_w;  // [B]   round-5 linearisation weights (own allocation: 480 B per proof)
    // wiring (plonk_prover_set_wiring): the wire cells are scattered from per-variable values on the device
    uint32_t* cell_index;      // [3][n]  variable index of each wire cell; n_vars = empty cell / padding row
    uint32_t* pub_index;       // [n_public]
    size_t n_vars;
    Fr* vars;                  // [B][n_vars] values of the resident batch (Montgomery)
    size_t vars_cap;           // elements
    plonk_srs* lag_srs;        // Lagrange-basis view of srs (PLONK_PROVER_LAGRANGE_COMMITS), owned by srs
    size_t resident_b;         // batch size of the witnesses currently resident (run / download must match it)
    unsigned long long* bad_input;  // device: index of the first uploaded value that was not below r, or ~0 (status bit 3)
    hipEvent_t ev_copied, ev_vars_read;  // async upload: the copy stream's H2D is done / the gather kernels have read `vars`
    bool vars_read_pending;
    Fq *commit_xy; // [9][B] x||y canonical
    uint8_t* commit_flags;  // [9][B]
    ProofState* state;      // [B]
    ChallengeConsts chal;   // 2^(256 j) R^2 mod r, for the challenge reduction in transcript_kernel
};

static inline dim3 grid1(size_t n, unsigned block = 256, size_t cap = 4096) {
    size_t g = (n + block - 1) / block;
    if (g > cap) g = cap;
    if (!g) g = 1;
    return dim3((unsigned)g);
}

// ------------------------------------------------------------------------------------------------
// witness upload helper: PI[b][i] = -public[b][i] for i < n_public, 0 otherwise (prover.py:57-62)
__global__ void pi_fill_kernel(const Fr* pub, size_t n_public, size_t n, size_t B, Fr* pi) {
    const size_t total = B * n;
    for (size_t gI = (size_t)blockIdx.x * blockDim.x + threadIdx.x; gI < total; gI += (size_t)gridDim.x * blockDim.x) {
        const size_t b = gI / n, i = gI - b * n;
        Fr v = fp_zero<FrParams>();
        if (i < n_public) v = fp_neg(fp_load(pub + b * n_public + i));
        fp_store(pi + gI, v);
    }
}

// prover.py:94-103 on the device: A[i], B[i], C[i] = witness[wires[i].L / R / O], witness[None] = 0, zero padded to n.
// vars = [B][V] variable values; cell[3][n] = variable index of each wire cell (V: empty); out = wit_lag [3][B][n].
__global__ void witness_scatter_kernel(const Fr* vars, const uint32_t* cell, size_t V, size_t n, size_t B, Fr* out) {
    const size_t total = 3 * B * n;
    for (size_t gI = (size_t)blockIdx.x * blockDim.x + threadIdx.x; gI < total; gI += (size_t)gridDim.x * blockDim.x) {
        const size_t j = gI / (B * n), r = gI - j * B * n, b = r / n, i = r - b * n;
        const uint32_t idx = cell[j * n + i];
        fp_store(out + gI, idx < V ? fp_load(vars + b * V + idx) : fp_zero<FrParams>());
    }
}
__global__ void public_gather_kernel(const Fr* vars, const uint32_t* pub_index, size_t V, size_t l, size_t B, Fr* pub) {
    const size_t total = B * l;
    for (size_t gI = (size_t)blockIdx.x * blockDim.x + threadIdx.x; gI < total; gI += (size_t)gridDim.x * blockDim.x) {
        const size_t b = gI / l, k = gI - b * l;
        fp_store(pub + gI, fp_load(vars + b * V + pub_index[k]));
    }
}

// Sparse public inputs.  PI = sum_{i < l} (-pub_i) L_i with L_i the Lagrange basis of the n-th roots of unity:
//   coefficient j of L_i is  w^(-ij) / n            (an inverse DFT of a unit vector)
//   L_i(x) = (w^i / n) (x^n - 1) / (x - w^i)        (li_big holds it on the coset points: [3][n] coset-major)
__global__ void pi_coeffs_kernel(const Fr* pub, size_t l, const Fr* roots_inv, size_t n, size_t B, Fr n_inv, Fr* pic) {
    const size_t total = B * n;
    for (size_t gI = (size_t)blockIdx.x * blockDim.x + threadIdx.x; gI < total; gI += (size_t)gridDim.x * blockDim.x) {
        const size_t b = gI / n, j = gI - b * n;
        Fr acc = fp_zero<FrParams>();
        for (size_t i = 0; i < l; i++) acc = fp_add(acc, fp_mul(fp_load(pub + b * l + i), fp_load(roots_inv + ((i * j) & (n - 1)))));
        fp_store(pic + gI, fp_neg(fp_mul(acc, n_inv)));
    }
}
__global__ void pi_coset_kernel(const Fr* pub, size_t l, const Fr* li_big, size_t n4, size_t B, Fr* pi_big) {
    const size_t total = B * n4;
    for (size_t gI = (size_t)blockIdx.x * blockDim.x + threadIdx.x; gI < total; gI += (size_t)gridDim.x * blockDim.x) {
        const size_t b = gI / n4, k = gI - b * n4;
        Fr acc = fp_zero<FrParams>();
        for (size_t i = 0; i < l; i++) acc = fp_add(acc, fp_mul(fp_load(pub + b * l + i), fp_load(li_big + i * n4 + k)));
        fp_store(pi_big + gI, fp_neg(acc));
    }
}
// li[i][k] = (w^i / n) zh[k / n] / (x_k - w^i), k = r n + j (Z_H is constant on coset r); one field inversion per entry, once per circuit
struct Zh4 { Fr v[4]; };
__global__ void li_coset_kernel(const Fr* xs, const Fr* roots, size_t n4, size_t n, size_t l, Zh4 zh, Fr n_inv, Fr* li) {
    const size_t total = l * n4;
    for (size_t gI = (size_t)blockIdx.x * blockDim.x + threadIdx.x; gI < total; gI += (size_t)gridDim.x * blockDim.x) {
        const size_t i = gI / n4, k = gI - i * n4;
        const Fr wi = fp_load(roots + i);
        const Fr num = fp_mul(fp_mul(wi, n_inv), zh.v[k / n]);
        fp_store(li + gI, fp_mul(num, fp_inv(fp_sub(fp_load(xs + k), wi))));  // x_k is never an n-th root of unity
    }
}

// ------------------------------------------------------------------------------------------------
// Transcript rounds on the device (transcript.py:77-123 + merlin): 32 lanes per proof, two proofs per
// 64-lane workgroup.  Lane i < 25 keeps Keccak lane st[i] in registers; a permutation round exchanges
// lanes through LDS (two barriers per round) instead of one thread grinding through all 25 lanes, the
// STROBE byte operations become "the lane that owns byte idx xors it", and the 255-byte challenge is
// reduced mod r by eight lanes in parallel.  Same byte stream as csrc/transcript.h (the host C-ABI
// transcript), which the tests pin against the merlin test vector and the golden proof.
// Control flow depends only on message lengths, which are the same for every proof, so the barriers
// are uniform; an identity commitment (flag set) is absorbed as zeros and reported through `error`.
#define TC_LANES 32
struct TcShared {
    uint64_t buf[2][25];
    uint8_t msg[256];
    Fr part[8];
};
struct TcState {
    uint64_t w;  // st[lane] for lane < 25
    uint32_t pos, pos_begin;
};

PLONK_DEV void tc_keccak(TcState& t, TcShared& sh, unsigned lane) {
    constexpr unsigned rot[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    const bool act = lane < 25;
    const unsigned i = act ? lane : 24, x = i % 5, y = i / 5;
    const unsigned r = rot[i], dst = y + 5 * ((2 * x + 3 * y) % 5);
    const unsigned ca = (x + 4) % 5, cb = (x + 1) % 5, n1 = (x + 1) % 5 + 5 * y, n2 = (x + 2) % 5 + 5 * y;
    uint64_t a = t.w;
    for (int round = 0; round < 24; round++) {
        if (act) sh.buf[0][i] = a;
        __syncthreads();
        uint64_t c0 = 0, c1 = 0;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            c0 ^= sh.buf[0][ca + 5 * k];
            c1 ^= sh.buf[0][cb + 5 * k];
        }
        a ^= c0 ^ keccak_rotl(c1, 1);
        if (act) sh.buf[1][dst] = keccak_rotl(a, r);
        __syncthreads();
        a = sh.buf[1][i] ^ (~sh.buf[1][n1] & sh.buf[1][n2]);
        if (i == 0) a ^= keccak_rc(round);
    }
    t.w = a;
}

PLONK_DEV void tc_xor_byte(TcState& t, unsigned lane, unsigned idx, uint8_t b) {
    if (lane == (idx >> 3)) t.w ^= (uint64_t)b << (8 * (idx & 7));
}
PLONK_DEV void tc_run_f(TcState& t, TcShared& sh, unsigned lane) {
    tc_xor_byte(t, lane, t.pos, (uint8_t)t.pos_begin);
    tc_xor_byte(t, lane, t.pos + 1, 0x04);
    tc_xor_byte(t, lane, STROBE_R + 1, 0x80);
    tc_keccak(t, sh, lane);
    t.pos = 0;
    t.pos_begin = 0;
}
PLONK_DEV void tc_absorb_byte(TcState& t, TcShared& sh, unsigned lane, uint8_t b) {
    tc_xor_byte(t, lane, t.pos, b);
    if (++t.pos == STROBE_R) tc_run_f(t, sh, lane);
}
// data: constant / global / LDS bytes readable by every lane of the group
PLONK_DEV void tc_absorb(TcState& t, TcShared& sh, unsigned lane, const uint8_t* data, unsigned n) {
    while (n) {
        const unsigned take = n < STROBE_R - t.pos ? n : STROBE_R - t.pos;
#pragma unroll
        for (unsigned j = 0; j < 8; j++) {
            const unsigned idx = 8 * lane + j;
            if (idx >= t.pos && idx < t.pos + take) t.w ^= (uint64_t)data[idx - t.pos] << (8 * j);
        }
        t.pos += take;
        data += take;
        n -= take;
        if (t.pos == STROBE_R) tc_run_f(t, sh, lane);
    }
}
PLONK_DEV void tc_squeeze(TcState& t, TcShared& sh, unsigned lane, uint8_t* out, unsigned n) {
    while (n) {
        const unsigned take = n < STROBE_R - t.pos ? n : STROBE_R - t.pos;
#pragma unroll
        for (unsigned j = 0; j < 8; j++) {
            const unsigned idx = 8 * lane + j;
            if (idx >= t.pos && idx < t.pos + take) {
                out[idx - t.pos] = (uint8_t)(t.w >> (8 * j));
                t.w &= ~((uint64_t)0xff << (8 * j));
            }
        }
        t.pos += take;
        out += take;
        n -= take;
        if (t.pos == STROBE_R) tc_run_f(t, sh, lane);
    }
}
PLONK_DEV void tc_begin_op(TcState& t, TcShared& sh, unsigned lane, uint32_t flags) {
    const uint8_t h0 = (uint8_t)t.pos_begin;
    t.pos_begin = t.pos + 1;
    tc_absorb_byte(t, sh, lane, h0);
    tc_absorb_byte(t, sh, lane, (uint8_t)flags);
    if ((flags & (STROBE_FLAG_C | STROBE_FLAG_K)) && t.pos != 0) tc_run_f(t, sh, lane);
}
// meta-AD of label || u32le(len): the framing merlin puts in front of every message and challenge
PLONK_DEV void tc_frame(TcState& t, TcShared& sh, unsigned lane, const char* label, unsigned llen, unsigned len) {
    tc_begin_op(t, sh, lane, STROBE_FLAG_M | STROBE_FLAG_A);
    tc_absorb(t, sh, lane, (const uint8_t*)label, llen);
    for (int k = 0; k < 4; k++) tc_absorb_byte(t, sh, lane, (uint8_t)(len >> (8 * k)));
}
PLONK_DEV void tc_append_message(TcState& t, TcShared& sh, unsigned lane, const char* label, unsigned llen,
                                 const uint8_t* msg, unsigned mlen) {
    tc_frame(t, sh, lane, label, llen, mlen);
    tc_begin_op(t, sh, lane, STROBE_FLAG_A);
    tc_absorb(t, sh, lane, msg, mlen);
}

// transcript.py:69-75: 255 PRF bytes -> big-endian integer mod r (retry on zero) -> re-appended.  Returns the
// challenge (Montgomery form) in every lane of the group.
PLONK_DEV Fr tc_draw(TcState& t, TcShared& sh, unsigned lane, const ChallengeConsts& cc, const char* label, unsigned llen) {
    for (;;) {
        tc_frame(t, sh, lane, label, llen, 255);
        tc_begin_op(t, sh, lane, STROBE_FLAG_I | STROBE_FLAG_A | STROBE_FLAG_C);
        __syncthreads();  // earlier readers of sh.msg are done
        tc_squeeze(t, sh, lane, sh.msg, 255);
        __syncthreads();
        if (lane < 8) {  // chunk 0 = the leading 31 bytes, chunk c >= 1 = the next 32; weight 2^(256 (7 - c))
            const unsigned take = lane ? 32 : 31, off = lane ? 31 + 32 * (lane - 1) : 0;
            Fr chunk;  // little-endian limbs of the big-endian chunk
#pragma unroll
            for (unsigned l = 0; l < 8; l++) {
                uint32_t wv = 0;
#pragma unroll
                for (unsigned k = 0; k < 4; k++) {
                    const unsigned sig = 4 * l + k;  // byte significance within the chunk
                    if (sig < take) wv |= (uint32_t)sh.msg[off + take - 1 - sig] << (8 * k);
                }
                chunk.v[l] = wv;
            }
            sh.part[lane] = fp_mul(chunk, cc.c[7 - lane]);
        }
        __syncthreads();
        Fr f = sh.part[0];
        for (int k = 1; k < 8; k++) f = fp_add(f, sh.part[k]);
        if (!fp_is_zero(f)) {
            tc_append_message(t, sh, lane, label, llen, sh.msg, 255);
            return f;
        }
    }
}

__global__ void __launch_bounds__(2 * TC_LANES) transcript_kernel(int round, ProofState* st, size_t B, const Fq* commit_xy,
                                                                  const uint8_t* flags, ChallengeConsts cc) {
    __shared__ TcShared shared[2];
    const unsigned grp = threadIdx.x / TC_LANES, lane = threadIdx.x % TC_LANES;
    TcShared& sh = shared[grp];
    size_t b = (size_t)blockIdx.x * 2 + grp;
    const bool live = b < B;
    if (!live) b = B - 1;  // shadow the last proof so the barriers stay uniform; nothing is stored
    ProofState& s = st[b];
    TcState t;
    t.w = lane < 25 ? s.transcript.st[lane] : 0;
    t.pos = s.transcript.pos;
    t.pos_begin = s.transcript.pos_begin;
    uint32_t error = 0;
    Fr c0 = fp_zero<FrParams>(), c1 = fp_zero<FrParams>();

    // 32-byte big-endian encodings of up to six values go to sh.msg[32 k]
    auto stage_points = [&](int first_slot, int count) {
        if (lane < 2u * count) {
            const size_t slot = (size_t)first_slot + lane / 2;
            const uint8_t fl = flags[slot * B + b];
            Fq v = fp_load(commit_xy + 2 * (slot * B + b) + (lane & 1));
            if (fl) {
                v = fp_zero<FqParams>();
                error = 1;
            }
            limbs_to_be32(v.v, sh.msg + 32 * lane);
        }
        __syncthreads();
    };
    auto absorb_point = [&](int k, const char* label, unsigned llen) {  // transcript.py:62-67: x then y
        tc_append_message(t, sh, lane, label, llen, sh.msg + 64 * k, 32);
        tc_append_message(t, sh, lane, label, llen, sh.msg + 64 * k + 32, 32);
    };

    if (round == 0) {  // Transcript(b"plonk"), prover.py:53
        const uint8_t init[18] = {1, STROBE_R + 2, 1, 0, 1, 96, 'S', 'T', 'R', 'O', 'B', 'E', 'v', '1', '.', '0', '.', '2'};
        t.w = 0;
        for (unsigned j = 0; j < 8; j++)
            if (8 * lane + j < 18) t.w |= (uint64_t)init[8 * lane + j] << (8 * j);
        tc_keccak(t, sh, lane);
        t.pos = 0;
        t.pos_begin = 0;
        tc_begin_op(t, sh, lane, STROBE_FLAG_M | STROBE_FLAG_A);
        tc_absorb(t, sh, lane, (const uint8_t*)"Merlin v1.0", 11);
        tc_append_message(t, sh, lane, "dom-sep", 7, (const uint8_t*)"plonk", 5);
    } else if (round == 1) {  // transcript.py:77-86
        stage_points(0, 3);
        absorb_point(0, "a_1", 3);
        absorb_point(1, "b_1", 3);
        absorb_point(2, "c_1", 3);
        c0 = tc_draw(t, sh, lane, cc, "beta", 4);
        c1 = tc_draw(t, sh, lane, cc, "gamma", 5);
    } else if (round == 2) {  // transcript.py:88-97
        stage_points(3, 1);
        absorb_point(0, "z_1", 3);
        c0 = tc_draw(t, sh, lane, cc, "alpha", 5);
        c1 = tc_draw(t, sh, lane, cc, "fft_cofactor", 12);
    } else if (round == 3) {  // transcript.py:99-105
        stage_points(4, 3);
        absorb_point(0, "t_lo_1", 6);
        absorb_point(1, "t_mid_1", 7);
        absorb_point(2, "t_hi_1", 6);
        c0 = tc_draw(t, sh, lane, cc, "zeta", 4);
    } else if (round == 4) {  // transcript.py:107-116
        if (lane < 6) {
            Fr e = fp_from_mont(s.evals[lane]);
            limbs_to_be32(e.v, sh.msg + 32 * lane);
        }
        __syncthreads();
        tc_append_message(t, sh, lane, "a_eval", 6, sh.msg, 32);
        tc_append_message(t, sh, lane, "b_eval", 6, sh.msg + 32, 32);
        tc_append_message(t, sh, lane, "c_eval", 6, sh.msg + 64, 32);
        tc_append_message(t, sh, lane, "s1_eval", 7, sh.msg + 96, 32);
        tc_append_message(t, sh, lane, "s2_eval", 7, sh.msg + 128, 32);
        tc_append_message(t, sh, lane, "z_shifted_eval", 14, sh.msg + 160, 32);
        c0 = tc_draw(t, sh, lane, cc, "v", 1);
    }
    if (!live) return;
    if (lane < 25) s.transcript.st[lane] = t.w;
    // `error` was raised by the lanes that staged a flagged coordinate
    if (error) s.error = 1;
    if (lane == 0) {
        s.transcript.pos = t.pos;
        s.transcript.pos_begin = t.pos_begin;
        if (round == 0) s.error = 0;
        if (round == 1) { s.beta = c0; s.gamma = c1; }
        if (round == 2) { s.alpha = c0; s.fft_cofactor = c1; }
        if (round == 3) s.zeta = c0;
        if (round == 4) s.v = c0;
    }
}

// ------------------------------------------------------------------------------------------------
// Round 2 (prover.py:121-152): the permutation grand product, one workgroup per proof.
//   num_i = (A_i + b w^i + g)(B_i + 2 b w^i + g)(C_i + 3 b w^i + g)
//   den_i = (A_i + b S1_i + g)(B_i + b S2_i + g)(C_i + b S3_i + g)
//   Z_0 = 1, Z_{i+1} = Z_i num_i / den_i
// With PN_i = prod_{j<i} num_j and SD_i = prod_{j>=i} den_j:  Z_i = PN_i * SD_i / prod_j den_j, so the
// whole column costs two block scans and ONE field inversion.  A zero denominator factor is skipped
// in the scans and zeroes the ratio it belongs to (py_ecc: x / 0 == 0).
#define GP_THREADS 256
// Inputs by pointer: proof b's columns at abc[k] + b n, the permutation polynomials sig[k] shared.  The challenges come
// from the proofs' transcript states (st, the lock-step prover) or, st == null, from `direct` (plonk_fr_grand_product).
struct RoundChallenges { Fr beta, gamma, alpha; };
struct GrandProductIn { const Fr* abc[3]; const Fr* sig[3]; };
__global__ void __launch_bounds__(GP_THREADS) grand_product_kernel(GrandProductIn in, const Fr* roots, const ProofState* st,
                                                                   RoundChallenges direct, size_t n, Fr* z_out,
                                                                   uint32_t* closes, Fr* num_buf, Fr* den_buf) {
    __shared__ Fr sc_n[GP_THREADS], sc_d[GP_THREADS];
    __shared__ Fr tot_inv;
    const size_t b = blockIdx.x;
    const unsigned tid = threadIdx.x;
    const Fr beta = st ? st[b].beta : direct.beta, gamma = st ? st[b].gamma : direct.gamma;
    const Fr *A = in.abc[0] + b * n, *Bv = in.abc[1] + b * n, *C = in.abc[2] + b * n;
    const Fr *S1 = in.sig[0], *S2 = in.sig[1], *S3 = in.sig[2];
    Fr *NUM = num_buf + b * n, *DEN = den_buf + b * n;  // this proof's factors; a lane only ever touches its own chunk
    const size_t per = (n + GP_THREADS - 1) / GP_THREADS;
    const size_t lo = tid * per, hi = (lo + per < n) ? lo + per : n;
    const Fr one = fp_one<FrParams>();

    // pass 1: the factors, once; per-lane products
    Fr pn = one, pd = one;
    for (size_t i = lo; i < hi; i++) {
        Fr a = fp_load(A + i), bb = fp_load(Bv + i), c = fp_load(C + i);
        Fr bw = fp_mul(beta, fp_load(roots + i));
        Fr ag = fp_add(a, gamma), bg = fp_add(bb, gamma), cg = fp_add(c, gamma);
        Fr num = fp_mul(fp_mul(fp_add(ag, bw), fp_add(bg, fp_dbl(bw))), fp_add(cg, fp_mul3(bw)));
        Fr den = fp_mul(fp_mul(fp_add(ag, fp_mul(beta, fp_load(S1 + i))), fp_add(bg, fp_mul(beta, fp_load(S2 + i)))),
                        fp_add(cg, fp_mul(beta, fp_load(S3 + i))));
        if (fp_is_zero(den)) {  // ratio num/0 == 0 (py_ecc): the factor leaves the denominator products
            num = fp_zero<FrParams>();
            den = one;
        }
        fp_store(NUM + i, num);
        fp_store(DEN + i, den);
        pd = fp_mul(pd, den);
        pn = fp_mul(pn, num);
    }
    sc_n[tid] = pn;
    sc_d[tid] = pd;
    __syncthreads();
    // inclusive prefix scan of sc_n (Hillis-Steele), inclusive suffix scan of sc_d
    for (unsigned off = 1; off < GP_THREADS; off <<= 1) {
        Fr vn = sc_n[tid], vd = sc_d[tid];
        if (tid >= off) vn = fp_mul(vn, sc_n[tid - off]);
        if (tid + off < GP_THREADS) vd = fp_mul(vd, sc_d[tid + off]);
        __syncthreads();
        sc_n[tid] = vn;
        sc_d[tid] = vd;
        __syncthreads();
    }
    if (tid == 0) tot_inv = fp_inv(sc_d[0]);  // product of all non-zero denominators
    __syncthreads();
    Fr run_n = tid ? sc_n[tid - 1] : one;                       // prod of num before this lane's chunk
    Fr after_d = (tid + 1 < GP_THREADS) ? sc_d[tid + 1] : one;  // prod of den after this lane's chunk
    const Fr tinv = tot_inv;
    // pass 2 (backwards): DEN[k] <- prod_{j >= k} den_j
    for (size_t k = hi; k-- > lo;) {
        after_d = fp_mul(after_d, fp_load(DEN + k));
        fp_store(DEN + k, after_d);
    }
    // pass 3: Z_i = PN_i * SD_i * tot_inv
    for (size_t i = lo; i < hi; i++) {
        fp_store(z_out + b * n + i, fp_mul(fp_mul(run_n, fp_load(DEN + i)), tinv));
        run_n = fp_mul(run_n, fp_load(NUM + i));
    }
    // prover.py:132 `assert Z_values.pop() == 1`: the full product of ratios must close to one
    if (tid == GP_THREADS - 1) closes[b] = fp_eq(fp_mul(run_n, tinv), fp_one<FrParams>()) ? 1u : 0u;
}

// ------------------------------------------------------------------------------------------------
// Round 3 (prover.py:188-203): quotient evaluations on the coset points, fully fused.
//   wit = A, B, C, PI, Z on the points, proof b at + b n4; fixed = QM, QL, QR, QO, QC, S1, S2, S3 (FX_* order), l0, xs: [n4].
//   Challenges from the transcript states (st) or, st == null, from `direct` (plonk_fr_quotient).
//   coset_log == 0: the reference's layout — n4 = 4n points g mu^k in natural order, Z(w x) four places ahead (prover.py:173),
//   Z_H by k & 3 (plonk_fr_quotient).  coset_log == log2 n: the lock-step prover's — n4 = 3n points [r][j] = g mu^r w^j, Z(w x)
//   one place ahead INSIDE the coset, Z_H by r.  The quotient goes to quot[b * out_stride + k].
struct ZhInv { Fr v[4]; };
struct QuotientIn { const Fr* wit[5]; const Fr* fixed[FX_COUNT]; const Fr* l0; const Fr* xs; };
// Round 4: the arithmetic runs on lazy limbs (fpl.h) — operands stay unpacked between the 19 products of a point, the gate's
// four products share two reductions (fpl_mul_add), sums of two are multiplied as they stand and only the seven sums of three
// or more are carry-swept: ~4 800 instructions per point against ~5 800 on packed residues.  Bounds, in units of m, beside
// each line ("n" = normalised: limbs 0..7 in [0, 2^29)); the emulator build asserts them on every operand.
__global__ void __launch_bounds__(256) quotient_kernel(QuotientIn in, ZhInv zh, const ProofState* st, RoundChallenges direct, unsigned n4, Fr* quot,
                                                       unsigned coset_log, unsigned out_stride) {
    typedef FpL<FrParams> L;
    // blockIdx.y = proof, blockIdx.x * blockDim.x + threadIdx.x = point (32-bit indices, no divisions)
    const unsigned b = blockIdx.y;
    // the challenges are the same for every lane of the block: limbs in scalar registers
    const L beta = fpl_from_fp_uniform(st ? st[b].beta : direct.beta), gamma = fpl_from_fp_uniform(st ? st[b].gamma : direct.gamma),
            alpha = fpl_from_fp_uniform(st ? st[b].alpha : direct.alpha);                                   // n, [0, 1)
    const L one = fpl_one<FrParams>();
    const size_t row = (size_t)b * n4;
    for (unsigned k = blockIdx.x * blockDim.x + threadIdx.x; k < n4; k += gridDim.x * blockDim.x) {
        const unsigned cr = coset_log ? k >> coset_log : 0u, cmask = coset_log ? (1u << coset_log) - 1u : 0u;
        const unsigned kw = coset_log ? ((k & ~cmask) | ((k + 1) & cmask))            // the next point of the same coset
                                      : ((k + 4 < n4) ? k + 4 : k + 4 - n4);          // Z(w x) = Z_big.shift(4), prover.py:173
        const auto ld = [&](const Fr* p) PLONK_LAMBDA_INLINE { return fpl_from_fp(fp_load(p)); };          // n, [0, 1)
        const L a = ld(in.wit[0] + row + k), bb = ld(in.wit[1] + row + k), c = ld(in.wit[2] + row + k);
        // gate: A QL + B QR + A B QM + C QO + PI + QC
        const L t1 = fpl_mul_add(a, ld(in.fixed[FX_QL] + k), bb, ld(in.fixed[FX_QR] + k));               // n, (-1, 2)
        const L ab = fpl_mul(a, bb);                                                                       // n, (-1, 2)
        const L t2 = fpl_mul_add(ab, ld(in.fixed[FX_QM] + k), c, ld(in.fixed[FX_QO] + k));               // n, (-1, 2)
        const L gate = fpl_norm(fpl_add(fpl_add(t1, t2), fpl_add(ld(in.wit[3] + row + k), ld(in.fixed[FX_QC] + k))));  // four terms: limbs < 2^31; n, (-2, 6)
        // permutation: (A + g + b x)(B + g + 2 b x)(C + g + 3 b x) Z - (A + g + b S1)(B + g + b S2)(C + g + b S3) Z(w x)
        const L bx = fpl_mul(beta, ld(in.xs + k));                                                         // n, (-1, 2)
        const L u1 = fpl_norm(fpl_add(gamma, bx));                                                         // n, (-1, 3)
        const L u2 = fpl_norm(fpl_add(u1, bx));                                                            // n, (-2, 5)
        const L u3 = fpl_norm(fpl_add(u2, bx));                                                            // n, (-3, 7)
        const L z = ld(in.wit[4] + row + k);
        L p1 = fpl_mul(fpl_add(a, u1), z);                                                                 // (two terms) x n: |.| < 4;  n, (-1, 2)
        p1 = fpl_mul(fpl_add(bb, u2), p1);                                                                 // < 6 x 2
        p1 = fpl_mul(fpl_add(c, u3), p1);                                                                  // < 8 x 2
        const L v1 = fpl_norm(fpl_add(gamma, fpl_mul(beta, ld(in.fixed[FX_S1] + k))));                     // n, (-1, 3)
        L p2 = fpl_mul(fpl_add(a, v1), ld(in.wit[4] + row + kw));                                          // < 4 x 1
        const L v2 = fpl_norm(fpl_add(gamma, fpl_mul(beta, ld(in.fixed[FX_S2] + k))));
        p2 = fpl_mul(fpl_add(bb, v2), p2);                                                                 // < 4 x 2
        const L v3 = fpl_norm(fpl_add(gamma, fpl_mul(beta, ld(in.fixed[FX_S3] + k))));
        p2 = fpl_mul(fpl_add(c, v3), p2);
        // (Z - 1) L0, and the sum under alpha
        const L first = fpl_mul(fpl_sub(z, one), ld(in.l0 + k));                     // difference x n;  n, (-1, 2)
        const L inner = fpl_add(fpl_sub(p1, p2), fpl_mul(alpha, first));                                   // limbs within (-2^29, 2^30); (-4, 5)
        const L acc = fpl_add(gate, fpl_mul(alpha, inner));                                                // two n terms; (-3, 8)
        L zhi;
        if (coset_log >= 6) zhi = fpl_from_fp_uniform(zh.v[cr]);  // a wave's 64 points lie in one coset (64 | n): scalar registers
        else zhi = fpl_from_fp(zh.v[coset_log ? cr : (k & 3)]);
        fp_store(quot + (size_t)b * out_stride + k, fpl_pack_canonical(fpl_mul(acc, zhi)));
    }
}

// The quotient's coefficients from its values on three cosets.  With t = T_0 + X^n T_1 + X^2n T_2 (deg T_q < n) and
// x^n = g^n i^r on coset r (i = mu^n, a primitive fourth root of unity), the polynomial U_r = T_0 + (g^n i^r) T_1 + (g^n i^r)^2 T_2
// agrees with t on coset r; the size-n inverse transforms (with their (g mu^r)^-i store-side scaling, which also halves)
// leave u_r[i] / 2 in slice r of quot.  Per i, with s_1 = g^n t_{n+i}, s_2 = g^2n t_{2n+i}:
//   u_0 = t_i + s_1 + s_2,   u_1 = t_i + i s_1 - s_2,   u_2 = t_i - s_1 + s_2
//   =>  s_1 = u_0/2 - u_2/2,   p = t_i + s_2 = u_0/2 + u_2/2,   m = t_i - s_2 = u_1 - i s_1,   t_i = (p + m)/2,  s_2 = (p - m)/2.
// in place: slice q of quot[b] <- T_q.  Three multiplications per i (by i, 1/g^n, 1/(2 g^2n)) and a halving.
__global__ void quotient_combine_kernel(Fr* quot, size_t n, size_t stride, size_t B, Fr ci, Fr g1_inv, Fr g2_inv_half, Fr half) {
    const size_t total = B * n;
    for (size_t gI = (size_t)blockIdx.x * blockDim.x + threadIdx.x; gI < total; gI += (size_t)gridDim.x * blockDim.x) {
        const size_t b = gI / n, i = gI - b * n;
        Fr* q = quot + b * stride + i;
        const Fr h0 = fp_load(q), h1 = fp_load(q + n), h2 = fp_load(q + 2 * n);  // u_r / 2
        const Fr s1 = fp_sub(h0, h2), p = fp_add(h0, h2);
        const Fr m = fp_sub(fp_dbl(h1), fp_mul(ci, s1));
        fp_store(q, fp_mul(fp_add(p, m), half));
        fp_store(q + n, fp_mul(s1, g1_inv));
        fp_store(q + 2 * n, fp_mul(fp_sub(p, m), g2_inv_half));
    }
}

// prover.py:108-116 — the gate identity on H, row by row: A QL + B QR + A B QM + C QO + PI + QC = 0.  flags[b] |= 1 otherwise
// (status bit 2).  Together with "Z closes to 1" (prover.py:132, status bit 1) this is exactly when the quotient's numerator is
// divisible by Z_H — the condition the reference re-checks on the quotient's top coefficients (prover.py:205-208), which a
// quotient interpolated from 3n points no longer has.
__global__ void gate_check_kernel(const Fr* abc, const Fr* pub, size_t l, const Fr* pi_or_null, const Fr* fixed_lag, size_t n, size_t B, uint32_t* bad) {
    const size_t total = B * n;
    for (size_t gI = (size_t)blockIdx.x * blockDim.x + threadIdx.x; gI < total; gI += (size_t)gridDim.x * blockDim.x) {
        const size_t b = gI / n, i = gI - b * n;
        const Fr a = fp_load(abc + gI), bb = fp_load(abc + total + gI), c = fp_load(abc + 2 * total + gI);
        Fr pi = fp_zero<FrParams>();
        if (pi_or_null) pi = fp_load(pi_or_null + gI);
        else if (i < l) pi = fp_neg(fp_load(pub + b * l + i));
        Fr gate = fp_add(fp_mul(a, fp_load(fixed_lag + FX_QL * n + i)), fp_mul(bb, fp_load(fixed_lag + FX_QR * n + i)));
        gate = fp_add(gate, fp_mul(fp_mul(a, bb), fp_load(fixed_lag + FX_QM * n + i)));
        gate = fp_add(gate, fp_mul(c, fp_load(fixed_lag + FX_QO * n + i)));
        gate = fp_add(gate, fp_add(pi, fp_load(fixed_lag + FX_QC * n + i)));
        if (!fp_is_zero(gate)) atomicOr(&bad[b], 1u);
    }
}

// ------------------------------------------------------------------------------------------------
// Round 4 (prover.py:228-239): evaluate coefficient forms at zeta (and Z at zeta*w), one workgroup
// per proof: lane t Horner-evaluates its chunk of each polynomial, scales by x^(chunk start) and
// the workgroup tree-reduces.  polys: Ac, Bc, Cc, S1c, S2c at zeta; Zc at zeta*w; PIc at zeta.
#define EV_THREADS 256
__global__ void __launch_bounds__(EV_THREADS) eval_kernel(const Fr* coef, const Fr* fixed_coef, Fr w, ProofState* st,
                                                          size_t n, size_t B) {
    __shared__ Fr red[NEVAL][EV_THREADS];
    const size_t b = blockIdx.x;
    const unsigned tid = threadIdx.x;
    const Fr zeta = st[b].zeta, zeta_w = fp_mul(zeta, w);
    const Fr* polys[NEVAL] = {coef + (0 * B + b) * n, coef + (1 * B + b) * n, coef + (2 * B + b) * n,
                              fixed_coef + FX_S1 * n,  fixed_coef + FX_S2 * n,  coef + (4 * B + b) * n,
                              coef + (3 * B + b) * n};
    const size_t per = (n + EV_THREADS - 1) / EV_THREADS;
    const size_t lo = tid * per, hi = (lo + per < n) ? lo + per : n;
    // x^(chunk start) once per evaluation point, not once per polynomial
    const Fr shift_z = fp_pow_u64(zeta, (uint64_t)lo), shift_zw = fp_pow_u64(zeta_w, (uint64_t)lo);
    // Horner on lazy limbs (round 4), the seven chains side by side: acc x is normalised in (-m, 2m), plus a coefficient it is a
    // sum of two — a valid multiplicand as it stands; the evaluation points sit in scalar registers
    typedef FpL<FrParams> L;
    const L zl = fpl_from_fp_uniform(zeta), zwl = fpl_from_fp_uniform(zeta_w);
    L acc[NEVAL];
    wave_for<NEVAL>([&](auto P_) { acc[decltype(P_)::value] = fpl_zero<FrParams>(); });
#pragma unroll 1
    for (size_t i = hi; i-- > lo;)
        wave_for<NEVAL>([&](auto P_) {
            constexpr unsigned p = decltype(P_)::value;
            acc[p] = fpl_add(fpl_mul(acc[p], p == 5 ? zwl : zl), fpl_from_fp(fp_load(polys[p] + i)));  // (-m, 3m), limbs < 2^30
        });
    const L sh_z = fpl_from_fp(shift_z), sh_zw = fpl_from_fp(shift_zw);
    wave_for<NEVAL>([&](auto P_) {
        constexpr unsigned p = decltype(P_)::value;
        red[p][tid] = lo < hi ? fpl_pack_canonical(fpl_mul(acc[p], p == 5 ? sh_zw : sh_z)) : fp_zero<FrParams>();
    });
    __syncthreads();
    for (unsigned s = EV_THREADS / 2; s > 0; s >>= 1) {  // the seven sums share the barriers
        if (tid < s)
            for (int p = 0; p < NEVAL; p++) red[p][tid] = fp_add(red[p][tid], red[p][tid + s]);
        __syncthreads();
    }
    if (tid < NEVAL) st[b].evals[tid] = red[tid][0];
}

// ------------------------------------------------------------------------------------------------
// Round 5 (prover.py:241-306) in coefficient form.  numerator of W_z:
//   R + v(A - a) + v^2(B - b) + v^3(C - c) + v^4(S1 - s1) + v^5(S2 - s2)
// is a linear combination of 15 coefficient vectors; the constants only change the remainder of the
// division by (X - zeta), which is zero by construction, so they are not needed for the quotient.
struct LinWeights { Fr w[15]; };
PLONK_DEV LinWeights linearisation_weights(const ProofState& s, unsigned log_n, Fr n_inv) {
    const Fr one = fp_one<FrParams>();
    const Fr a = s.evals[0], b = s.evals[1], c = s.evals[2], s1 = s.evals[3], s2 = s.evals[4], zw = s.evals[5];
    const Fr beta = s.beta, gamma = s.gamma, alpha = s.alpha, zeta = s.zeta, v = s.v;
    Fr zn = zeta;
    for (unsigned i = 0; i < log_n; i++) zn = fp_sqr(zn);
    const Fr zh = fp_sub(zn, one);                                       // Z_H(zeta)
    const Fr l0 = fp_mul(fp_mul(zh, n_inv), fp_inv(fp_sub(zeta, one)));  // L0(zeta) = Z_H / (n (zeta - 1))
    const Fr bz = fp_mul(beta, zeta);
    const Fr k1 = fp_mul(fp_mul(fp_add(fp_add(a, bz), gamma), fp_add(fp_add(b, fp_dbl(bz)), gamma)),
                         fp_add(fp_add(c, fp_mul3(bz)), gamma));
    const Fr k2 = fp_mul(fp_mul(fp_add(fp_add(a, fp_mul(beta, s1)), gamma), fp_add(fp_add(b, fp_mul(beta, s2)), gamma)), zw);
    const Fr a2 = fp_sqr(alpha);
    LinWeights L;
    L.w[0] = fp_mul(a, b);                                  // QM
    L.w[1] = a;                                             // QL
    L.w[2] = b;                                             // QR
    L.w[3] = c;                                             // QO
    L.w[4] = one;                                           // QC
    L.w[5] = fp_add(fp_mul(alpha, k1), fp_mul(a2, l0));     // Z
    L.w[6] = fp_neg(fp_mul(fp_mul(alpha, beta), k2));       // S3
    L.w[7] = fp_neg(zh);                                    // T1
    L.w[8] = fp_neg(fp_mul(zh, zn));                        // T2
    L.w[9] = fp_neg(fp_mul(zh, fp_sqr(zn)));                // T3
    Fr vp = v;
    L.w[10] = vp;                                           // A
    vp = fp_mul(vp, v); L.w[11] = vp;                       // B
    vp = fp_mul(vp, v); L.w[12] = vp;                       // C
    vp = fp_mul(vp, v); L.w[13] = vp;                       // S1
    vp = fp_mul(vp, v); L.w[14] = vp;                       // S2
    return L;
}

// one lane per proof: the 15 weights (one field inversion each) are computed once, not once per tile
__global__ void __launch_bounds__(64) linearisation_weights_kernel(const ProofState* st, unsigned log_n, Fr n_inv, size_t B, LinWeights* out) {
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) out[b] = linearisation_weights(st[b], log_n, n_inv);
}

// (lazy limbs, round 4: the 15 products of a coefficient pair up under 8 reductions — fpl_mul_add — with the weights unpacked
// once per block into LDS; the sum of the first seven results rides as "1 x sum" in the eighth, which leaves it reduced)
__global__ void __launch_bounds__(256, 4) linearisation_kernel(const Fr* coef, const Fr* fixed_coef, const Fr* tcoef,
                                                              const LinWeights* weights, unsigned log_n, size_t B,
                                                              Fr* out) {
    typedef FpL<FrParams> L;
    __shared__ int32_t WL[16][12];  // the weights as limbs (9 of 12 words used); [15] = R mod m, the Montgomery form of 1
    const size_t n = (size_t)1 << log_n;
    const size_t b = blockIdx.y;
    if (threadIdx.x < 16) {
        const L w = threadIdx.x < 15 ? fpl_from_fp(fp_load(&weights[b].w[threadIdx.x])) : fpl_one<FrParams>();
#pragma unroll
        for (int q = 0; q < 9; q++) WL[threadIdx.x][q] = w.l[q];
    }
    __syncthreads();
    const Fr* vec[15] = {fixed_coef + FX_QM * n, fixed_coef + FX_QL * n, fixed_coef + FX_QR * n, fixed_coef + FX_QO * n,
                         fixed_coef + FX_QC * n, coef + (4 * B + b) * n, fixed_coef + FX_S3 * n, tcoef + b * 4 * n,
                         tcoef + b * 4 * n + n,   tcoef + b * 4 * n + 2 * n, coef + (0 * B + b) * n,
                         coef + (1 * B + b) * n,  coef + (2 * B + b) * n,    fixed_coef + FX_S1 * n, fixed_coef + FX_S2 * n};
    const auto weight = [&](unsigned j) PLONK_LAMBDA_INLINE {
        L w;
#pragma unroll
        for (int q = 0; q < 9; q++) {
            w.l[q] = WL[j][q];
            FPL_ANY_SIGN(w.l[q]);
        }
        return w;  // normalised, [0, m)
    };
    const auto value = [&](unsigned j, size_t i) PLONK_LAMBDA_INLINE { return fpl_from_fp(fp_load(vec[j] + i)); };
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        L s = fpl_zero<FrParams>();
        wave_for<4>([&](auto Q) {  // four results, each normalised in (-m, 2m): limbs of the sum < 2^31
            constexpr unsigned j = 2 * decltype(Q)::value;
            s = fpl_add(s, fpl_mul_add(weight(j), value(j, i), weight(j + 1), value(j + 1, i)));
            PLONK_SCHED_FENCE();
        });
        s = fpl_norm(s);  // (-4 m, 8 m)
        L t = fpl_zero<FrParams>();
        wave_for<3>([&](auto Q) {
            constexpr unsigned j = 8 + 2 * decltype(Q)::value;
            t = fpl_add(t, fpl_mul_add(weight(j), value(j, i), weight(j + 1), value(j + 1, i)));
            PLONK_SCHED_FENCE();
        });
        const L u = fpl_norm(fpl_add(s, fpl_norm(t)));  // (-7 m, 14 m), normalised
        fp_store(out + b * n + i, fpl_pack_canonical(fpl_mul_add(weight(14), value(14, i), weight(15), u)));  // |.| < 1 + 14
    }
}

// q(X) = (p(X) - p(x0)) / (X - x0): q_{n-1} = 0, q_{i-1} = p_i + x0 q_i.  One workgroup per proof;
// lane t owns a chunk, the cross-chunk carries are a suffix scan under (a, m) o (b, m') = (a + m b, m m').
#define DV_THREADS 256
__global__ void __launch_bounds__(DV_THREADS) divide_linear_kernel(const Fr* p_in, size_t in_stride, int which,
                                                                  Fr w, const ProofState* st, size_t n, Fr* q_out) {
    __shared__ Fr sc[DV_THREADS];
    const size_t b = blockIdx.x;
    const unsigned tid = threadIdx.x;
    Fr x0 = st[b].zeta;
    if (which) x0 = fp_mul(x0, w);  // zeta * w for W_zw (prover.py:292-297)
    const Fr* p = p_in + b * in_stride;
    const size_t per = (n + DV_THREADS - 1) / DV_THREADS;
    const size_t lo = tid * per, hi = (lo + per < n) ? lo + per : n;
    // h_t = sum_{i in chunk} p_i x0^(i - lo)
    Fr h = fp_zero<FrParams>();
    for (size_t i = hi; i-- > lo;) h = fp_add(fp_mul(h, x0), fp_load(p + i));
    sc[tid] = h;
    __syncthreads();
    // suffix scan: after it, sc[t] = sum_{u >= t} h_u x0^((u - t) per)
    Fr m = fp_pow_u64(x0, (uint64_t)per);
    for (unsigned off = 1; off < DV_THREADS; off <<= 1) {
        Fr vv = sc[tid];
        if (tid + off < DV_THREADS) vv = fp_add(vv, fp_mul(m, sc[tid + off]));
        __syncthreads();
        sc[tid] = vv;
        m = fp_sqr(m);
        __syncthreads();
    }
    // carry into this chunk = value of q at index (hi - 1), i.e. contribution of all higher chunks:
    // q_{hi-1} = sum_{j >= hi} p_j x0^(j - hi) = sc[t + 1]
    Fr q = (tid + 1 < DV_THREADS) ? sc[tid + 1] : fp_zero<FrParams>();
    for (size_t i = hi; i-- > lo;) {
        fp_store(q_out + b * n + i, q);          // q_i
        q = fp_add(fp_load(p + i), fp_mul(x0, q));  // q_{i-1} = p_i + x0 q_i
    }
}

// ------------------------------------------------------------------------------------------------
// pack results: [B][768] = 9 x (x||y) canonical LE + 6 evaluations canonical LE
__global__ void pack_proofs_kernel(const Fq* commit_xy, const ProofState* st, size_t B, uint8_t* out, int compressed) {
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    if (compressed) {  // 480 bytes: nine compressed points, six big-endian scalars (g1_codec.h)
        uint8_t* o = out + b * 480;
        for (int slot = 0; slot < 9; slot++) {
            const Fq x = fp_load(commit_xy + 2 * ((size_t)slot * B + b)), y = fp_load(commit_xy + 2 * ((size_t)slot * B + b) + 1);
            g1c_compress<FqParams>(x.v, y.v, o + 32 * slot);
        }
        for (int e = 0; e < 6; e++) {
            const Fr v = fp_from_mont(st[b].evals[e]);
            g1c_be32(v.v, o + 288 + 32 * e);
        }
        return;
    }
    uint32_t* o = reinterpret_cast<uint32_t*>(out + b * 768);
    for (int slot = 0; slot < 9; slot++)
        for (int h = 0; h < 2; h++) {
            Fq v = fp_load(commit_xy + 2 * ((size_t)slot * B + b) + h);
            for (int i = 0; i < 8; i++) o[(slot * 2 + h) * 8 + i] = v.v[i];
        }
    for (int e = 0; e < 6; e++) {
        Fr v = fp_from_mont(st[b].evals[e]);
        for (int i = 0; i < 8; i++) o[144 + e * 8 + i] = v.v[i];
    }
}

// the status byte plonk_prover_download assembles on the host, on the device (plonk_gather_proofs_device)
__global__ void pack_status_kernel(const ProofState* st, const uint32_t* closes, const uint8_t* flags, const unsigned long long* bad_input,
                                   size_t n_vars, size_t B, uint8_t* out) {
    const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    uint8_t f = 0;
    for (int slot = 0; slot < 9; slot++) f |= flags[(size_t)slot * B + b] ? 1 : 0;
    if (st[b].error) f |= 1;
    if (!closes[b]) f |= 2;
    if (closes[B + b]) f |= 4;
    if (bad_input && *bad_input != ~0ull && n_vars && *bad_input / n_vars == b) f |= 8;
    out[b] = f;
}

// ================================================================================================
// host side
static int dev_alloc(void** p, size_t bytes) {
    if (!plonk_dev_malloc(p, bytes ? bytes : 32)) {
        plonk_set_error("hipMalloc(%zu) failed in the batched prover", bytes);
        return PLONK_ERR_NOMEM;
    }
    return PLONK_OK;
}

static Fr host_fr_u64(uint64_t x) {
    Fr a = fp_zero<FrParams>();
    a.v[0] = (uint32_t)x;
    a.v[1] = (uint32_t)(x >> 32);
    return fp_to_mont(a);
}

static void free_batch(plonk_prover* p) {
    void* bufs[] = {p->wit_lag, p->z_lag, p->coef, p->big, p->quot, p->num, p->den, p->wz, p->commit_xy, p->commit_flags, p->state, p->pub,
                    p->lin_w, p->vars};
    for (void* q : bufs)
        if (q) hipFree(q);
    p->pub = nullptr;
    p->lin_w = nullptr;
    p->vars = nullptr;
    p->vars_cap = 0;
    p->resident_b = 0;
    p->wit_lag = p->z_lag = p->coef = p->big = p->quot = p->num = p->den = p->wz = nullptr;
    p->commit_xy = nullptr;
    p->commit_flags = nullptr;
    p->state = nullptr;
    p->cap_b = 0;
}

static int ensure_batch(plonk_prover* p, size_t B) {
    if (B <= p->cap_b) return PLONK_OK;
    PLONK_CHECK_HIP(hipStreamSynchronize(p->ctx->stream));
    free_batch(p);
    const size_t n = p->n, e = sizeof(Fr);
    PLONK_TRY(dev_alloc((void**)&p->wit_lag, 4 * B * n * e));
    PLONK_TRY(dev_alloc((void**)&p->z_lag, B * n * e));
    PLONK_TRY(dev_alloc((void**)&p->coef, 5 * B * n * e));
    PLONK_TRY(dev_alloc((void**)&p->big, 5 * B * QCOSETS * n * e));
    PLONK_TRY(dev_alloc((void**)&p->quot, B * 4 * n * e));
    PLONK_TRY(dev_alloc((void**)&p->num, B * n * e));
    PLONK_TRY(dev_alloc((void**)&p->den, 2 * B * sizeof(uint32_t) + 64));
    PLONK_TRY(dev_alloc((void**)&p->wz, 2 * B * n * e));
    PLONK_TRY(dev_alloc((void**)&p->commit_xy, 9 * B * 2 * sizeof(Fq)));
    PLONK_TRY(dev_alloc((void**)&p->commit_flags, 9 * B));
    PLONK_TRY(dev_alloc((void**)&p->state, B * sizeof(ProofState)));
    PLONK_TRY(dev_alloc((void**)&p->pub, (B * p->n_public + 1) * e));
    PLONK_TRY(dev_alloc((void**)&p->lin_w, B * sizeof(LinWeights)));
    p->cap_b = B;
    return PLONK_OK;
}

static int prover_init(plonk_prover* p, plonk_ctx* ctx, plonk_srs* srs, unsigned log_n, const uint8_t* selectors_le32, size_t n_public);

extern "C" {

int plonk_prover_create(plonk_ctx* ctx, plonk_srs* srs, unsigned log_n, const uint8_t* selectors_le32,
                        size_t n_public, plonk_prover** out) {
    PLONK_REQUIRE(ctx && srs && selectors_le32 && out, PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    PLONK_REQUIRE(log_n >= 1 && log_n + 2 <= PLONK_FR_TWO_ADICITY, PLONK_ERR_ARG, "group_order 2^%u out of range", log_n);
    const size_t n = (size_t)1 << log_n;
    PLONK_REQUIRE(n <= 4096, PLONK_ERR_ARG, "the batched prover supports group_order <= 4096 (got %zu)", n);
    PLONK_REQUIRE(srs->n_points >= n, PLONK_ERR_ARG, "SRS has %zu powers, group_order is %zu", srs->n_points, n);
    PLONK_REQUIRE(n_public <= n, PLONK_ERR_ARG, "more public inputs than rows");
    plonk_prover* p = new plonk_prover();
    memset((void*)p, 0, sizeof *p);
    const int rc = prover_init(p, ctx, srs, log_n, selectors_le32, n_public);
    if (rc != PLONK_OK) {  // a single cleanup path: everything allocated so far goes with the half-built object
        plonk_prover_destroy(p);
        return rc;
    }
    *out = p;
    return PLONK_OK;
}

}  // extern "C"

static int prover_init(plonk_prover* p, plonk_ctx* ctx, plonk_srs* srs, unsigned log_n, const uint8_t* selectors_le32,
                       size_t n_public) {
    const size_t n = (size_t)1 << log_n;
    p->ctx = ctx;
    p->srs = srs;
    p->log_n = log_n;
    p->n = n;
    p->n_public = n_public;
    {   // c[j] = 2^(256 j) R^2: one Montgomery multiplication maps a 256-bit chunk to chunk * 2^(256 j) in Montgomery form
        Fr t = fp_zero<FrParams>();
        t.v[4] = 1;  // 2^128
        t = fp_to_mont(t);
        const Fr two256 = fp_mul(t, t);
        for (int i = 0; i < 8; i++) p->chal.c[0].v[i] = FrParams::r2(i);
        for (int j = 1; j < 8; j++) p->chal.c[j] = fp_mul(p->chal.c[j - 1], two256);
    }
    p->g = host_fr_u64(5);  // multiplicative generator (curve.py:5): g^(4n) != 1, so Z_H != 0 on the coset
    const size_t e = sizeof(Fr);
    PLONK_TRY(dev_alloc((void**)&p->fixed_lag, 8 * n * e));
    PLONK_TRY(dev_alloc((void**)&p->fixed_coef, 8 * n * e));
    const size_t n3 = QCOSETS * n;
    PLONK_TRY(dev_alloc((void**)&p->fixed_big, 8 * n3 * e));
    PLONK_TRY(dev_alloc((void**)&p->l0_big, n3 * e));
    PLONK_TRY(dev_alloc((void**)&p->x_big, n3 * e));
    PLONK_TRY(dev_alloc((void**)&p->g_pow, n3 * e));
    PLONK_TRY(dev_alloc((void**)&p->ginv_pow, n3 * e));
    PLONK_TRY(plonk_fr_upload(ctx, p->fixed_lag, selectors_le32, 8 * n));
    PLONK_TRY(ntt_get_roots(ctx, log_n, false, &p->roots));
    const Fr one = fp_one<FrParams>();
    const Fr half = fp_inv(host_fr_u64(2));
    const Fr mu = host_root_of_unity(log_n + 2, false), w = host_root_of_unity(log_n, false);
    Fr base = p->g;  // g mu^r
    for (unsigned r = 0; r < QCOSETS; r++) {
        PLONK_TRY(k_fr_powers(ctx, base, one, p->g_pow + r * n, n));
        PLONK_TRY(k_fr_powers(ctx, fp_inv(base), fp_mul(half, fp_inv(host_fr_u64((uint64_t)n))), p->ginv_pow + r * n, n));
        PLONK_TRY(k_fr_powers(ctx, w, base, p->x_big + r * n, n));
        base = fp_mul(base, mu);
    }
    // coefficient forms of the 8 circuit polynomials, and their values on the three cosets: P(g mu^r w^j) is the size-n
    // transform of c_i (g mu^r)^i
    PLONK_TRY(ntt_run(ctx, p->fixed_lag, p->fixed_coef, log_n, true, 8, n, n, n, nullptr, nullptr, true));
    const NttFan fan{QCOSETS, 0u, (unsigned)n, (unsigned)n};
    PLONK_TRY(ntt_run(ctx, p->fixed_coef, p->fixed_big, log_n, false, 8, n, n, n3, p->g_pow, nullptr, false, &fan));
    // L0: Lagrange vector e_0 has coefficient form (1/n, 1/n, ...)          prover.py:184-186
    Fr ninv = fp_inv(host_fr_u64((uint64_t)n));
    void* tmpv;
    PLONK_TRY(ctx_scratch(ctx, 2, n * e, &tmpv));  // context-owned scratch: nothing to leak on an error path
    Fr* tmp = (Fr*)tmpv;
    PLONK_TRY(k_fr_powers(ctx, one, ninv, tmp, n));
    PLONK_TRY(ntt_run(ctx, tmp, p->l0_big, log_n, false, 1, n, n, n3, p->g_pow, nullptr, false, &fan));
    PLONK_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    // Z_H on coset r is the constant (g mu^r)^n - 1 = g^n i^r - 1, i = mu^n            prover.py:178
    Fr gn = p->g;
    for (unsigned i = 0; i < log_n; i++) gn = fp_sqr(gn);
    Fr i4 = host_root_of_unity(2, false), cur = gn;
    Zh4 zh4;
    for (int k = 0; k < 4; k++) {
        zh4.v[k] = fp_sub(cur, one);
        if (k < QCOSETS) p->zh_inv[k] = fp_inv(zh4.v[k]);
        cur = fp_mul(cur, i4);
    }
    p->comb_i = i4;
    p->comb_g1 = fp_inv(gn);
    p->comb_g2 = fp_mul(fp_sqr(p->comb_g1), half);
    p->sparse_pi = n_public <= PI_SPARSE_MAX;
    if (p->sparse_pi && n_public) {
        PLONK_TRY(ntt_get_roots(ctx, log_n, true, &p->roots_inv));
        PLONK_TRY(dev_alloc((void**)&p->li_big, n_public * n3 * e));
        PLONK_LAUNCH(li_coset_kernel, grid1(n_public * n3), dim3(256), 0, ctx->stream, (const Fr*)p->x_big, p->roots, n3, n, n_public,
                     zh4, ninv, p->li_big);
        PLONK_CHECK_HIP(hipGetLastError());
        PLONK_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    }
    return PLONK_OK;  // (the MSM tables are built by the first commitment: lookup table or bucket-method window table)
}

extern "C" {

int plonk_prover_set_options(plonk_prover* p, unsigned flags) {
    PLONK_REQUIRE(p && !(flags & ~PLONK_PROVER_LAGRANGE_COMMITS), PLONK_ERR_ARG, "unknown prover option bits %#x", flags);
    PLONK_ENTER(p->ctx);
    p->lag_srs = nullptr;
    if (flags & PLONK_PROVER_LAGRANGE_COMMITS) PLONK_TRY(msm_lagrange_srs(p->ctx, p->srs, p->log_n, &p->lag_srs));
    return PLONK_OK;
}

int plonk_prover_destroy(plonk_prover* p) {
    if (!p) return PLONK_OK;
    if (p->ctx) {
        plonk_use_device(p->ctx->device);
        hipStreamSynchronize(p->ctx->stream);
    }
    free_batch(p);
    void* bufs[] = {p->fixed_lag, p->fixed_coef, p->fixed_big, p->l0_big, p->x_big, p->g_pow, p->ginv_pow, p->li_big, p->cell_index, p->pub_index};
    for (void* q : bufs)
        if (q) hipFree(q);
    if (p->bad_input) {
        hipFree(p->bad_input);
        hipEventDestroy(p->ev_copied);
        hipEventDestroy(p->ev_vars_read);
    }
    delete p;
    return PLONK_OK;
}

// witness columns [3][B][n] (A, B, C) and public inputs [B][n_public], canonical LE
int plonk_prover_upload_witness(plonk_prover* p, const uint8_t* abc_le32, const uint8_t* public_le32, size_t B) {
    PLONK_REQUIRE(p && abc_le32 && B && (public_le32 || !p->n_public), PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(p->ctx);
    PLONK_TRY(ensure_batch(p, B));
    plonk_ctx* ctx = p->ctx;
    const size_t n = p->n;
    // a verdict left by an earlier asynchronous upload does not belong to this batch (plonk_fr_upload reports its own
    // non-canonical values synchronously, as PLONK_ERR_ARG)
    if (p->bad_input) PLONK_CHECK_HIP(hipMemsetAsync(p->bad_input, 0xff, sizeof(unsigned long long), ctx->stream));
    PLONK_TRY(plonk_fr_upload(ctx, p->wit_lag, abc_le32, 3 * B * n));
    if (p->n_public) {
        PLONK_TRY(plonk_fr_upload(ctx, p->pub, public_le32, B * p->n_public));
        if (!p->sparse_pi)
            PLONK_LAUNCH(pi_fill_kernel, grid1(B * n), dim3(256), 0, ctx->stream, (const Fr*)p->pub, p->n_public, n, B,
                         p->wit_lag + 3 * B * n);
    } else {
        PLONK_CHECK_HIP(hipMemsetAsync(p->wit_lag + 3 * B * n, 0, B * n * sizeof(Fr), ctx->stream));
    }
    PLONK_CHECK_HIP(hipGetLastError());
    PLONK_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    p->resident_b = B;
    return PLONK_OK;
}

// cell_index[3][n]: variable index carried by each wire cell (column L/R/O, row), n_vars for an empty cell or a
// padding row; public_index[n_public]: the public variables, in the order of the public rows (prover.py:57-62).
int plonk_prover_set_wiring(plonk_prover* p, const uint32_t* cell_index, const uint32_t* public_index, size_t n_vars) {
    PLONK_REQUIRE(p && cell_index && n_vars && (public_index || !p->n_public), PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(p->ctx);
    for (size_t k = 0; k < 3 * p->n; k++)
        PLONK_REQUIRE(cell_index[k] <= n_vars, PLONK_ERR_ARG, "wire cell %zu names variable %u of %zu", k, cell_index[k], n_vars);
    for (size_t k = 0; k < p->n_public; k++)
        PLONK_REQUIRE(public_index[k] < n_vars, PLONK_ERR_ARG, "public input %zu names variable %u of %zu", k, public_index[k], n_vars);
    if (!p->cell_index) PLONK_TRY(dev_alloc((void**)&p->cell_index, 3 * p->n * sizeof(uint32_t)));
    if (!p->pub_index) PLONK_TRY(dev_alloc((void**)&p->pub_index, (p->n_public + 1) * sizeof(uint32_t)));
    PLONK_CHECK_HIP(hipMemcpyAsync(p->cell_index, cell_index, 3 * p->n * sizeof(uint32_t), hipMemcpyHostToDevice, p->ctx->stream));
    if (p->n_public)
        PLONK_CHECK_HIP(hipMemcpyAsync(p->pub_index, public_index, p->n_public * sizeof(uint32_t), hipMemcpyHostToDevice, p->ctx->stream));
    PLONK_CHECK_HIP(hipStreamSynchronize(p->ctx->stream));
    p->n_vars = n_vars;
    return PLONK_OK;
}

// values of the n_vars variables of each witness, [B][n_vars] canonical LE (n_vars * 32 bytes per proof instead of
// 3 * n * 32): the wire columns and the public inputs are gathered from them on the device (prover.py:94-103, 57-62).
// async: the host-to-device copy goes to the context's copy stream (it overlaps whatever the compute stream is running —
// another prover's rounds, or this prover's previous batch, which no longer reads `vars`), the conversion and the gather
// follow on the compute stream behind an event, nothing waits on the host; the canonical-range verdict stays on the
// device and comes back as status bit 3 of plonk_prover_download.  The caller keeps vars_le32 alive (and, for a copy
// that really is asynchronous, in pinned memory: plonk_host_alloc) until the batch has been downloaded.
static int prover_upload_vars(plonk_prover* p, const uint8_t* vars_le32, size_t B, bool async) {
    PLONK_REQUIRE(p && vars_le32 && B, PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(p->ctx);
    PLONK_REQUIRE(p->n_vars, PLONK_ERR_STATE, "plonk_prover_set_wiring has not been called");
    PLONK_TRY(ensure_batch(p, B));
    plonk_ctx* ctx = p->ctx;
    const size_t n = p->n, V = p->n_vars;
    if (p->vars_cap < B * V) {
        PLONK_CHECK_HIP(hipStreamSynchronize(ctx->stream));
        if (p->vars) hipFree(p->vars);
        p->vars = nullptr;
        p->vars_cap = 0;
        PLONK_TRY(dev_alloc((void**)&p->vars, B * V * sizeof(Fr)));
        p->vars_cap = B * V;
    }
    if (!p->bad_input) {
        PLONK_TRY(dev_alloc((void**)&p->bad_input, sizeof(unsigned long long)));
        PLONK_CHECK_HIP(hipEventCreate(&p->ev_copied));
        PLONK_CHECK_HIP(hipEventCreate(&p->ev_vars_read));
    }
    p->resident_b = 0;  // until the new batch is in place (a failed upload leaves no batch to run)
    if (async) {
        PLONK_TRY(ctx_copy_stream(ctx));
        if (p->vars_read_pending) PLONK_CHECK_HIP(hipStreamWaitEvent(ctx->copy_stream, p->ev_vars_read, 0));  // the previous gather has read vars
        PLONK_CHECK_HIP(hipMemcpyAsync(p->vars, vars_le32, B * V * sizeof(Fr), hipMemcpyHostToDevice, ctx->copy_stream));
        PLONK_CHECK_HIP(hipEventRecord(p->ev_copied, ctx->copy_stream));
        PLONK_CHECK_HIP(hipStreamWaitEvent(ctx->stream, p->ev_copied, 0));
        PLONK_TRY(k_fr_to_mont_checked(ctx, p->vars, B * V, p->bad_input));
    } else {
        PLONK_CHECK_HIP(hipMemsetAsync(p->bad_input, 0xff, sizeof(unsigned long long), ctx->stream));
        PLONK_TRY(plonk_fr_upload(ctx, p->vars, vars_le32, B * V));  // waits, and reports a non-canonical value as PLONK_ERR_ARG
    }
    PLONK_LAUNCH(witness_scatter_kernel, grid1(3 * B * n), dim3(256), 0, ctx->stream, (const Fr*)p->vars, (const uint32_t*)p->cell_index, V,
                 n, B, p->wit_lag);
    if (p->n_public) {
        PLONK_LAUNCH(public_gather_kernel, grid1(B * p->n_public), dim3(256), 0, ctx->stream, (const Fr*)p->vars,
                     (const uint32_t*)p->pub_index, V, p->n_public, B, p->pub);
        if (!p->sparse_pi)
            PLONK_LAUNCH(pi_fill_kernel, grid1(B * n), dim3(256), 0, ctx->stream, (const Fr*)p->pub, p->n_public, n, B,
                         p->wit_lag + 3 * B * n);
    } else {
        PLONK_CHECK_HIP(hipMemsetAsync(p->wit_lag + 3 * B * n, 0, B * n * sizeof(Fr), ctx->stream));
    }
    PLONK_CHECK_HIP(hipGetLastError());
    PLONK_CHECK_HIP(hipEventRecord(p->ev_vars_read, ctx->stream));
    p->vars_read_pending = true;
    if (!async) PLONK_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    p->resident_b = B;
    return PLONK_OK;
}

int plonk_prover_upload_variables(plonk_prover* p, const uint8_t* vars_le32, size_t B) { return prover_upload_vars(p, vars_le32, B, false); }
int plonk_prover_upload_variables_async(plonk_prover* p, const uint8_t* vars_le32, size_t B) { return prover_upload_vars(p, vars_le32, B, true); }

// Enqueue all five rounds for the B resident witnesses.  Asynchronous.
int plonk_prover_run(plonk_prover* p, size_t B) {
    PLONK_REQUIRE(p && B, PLONK_ERR_ARG, "bad argument");
    // every buffer is laid out [k][B][n] with the B of the upload: another B would read the wrong strides
    PLONK_REQUIRE(B == p->resident_b, PLONK_ERR_STATE, "run: batch %zu, but %zu witnesses are resident", B, p->resident_b);
    PLONK_ENTER(p->ctx);
    plonk_ctx* ctx = p->ctx;
    const size_t n = p->n, n4 = 4 * n;
    const unsigned log_n = p->log_n;
    const unsigned tb = (unsigned)((B + 63) / 64), tg = (unsigned)((B + 1) / 2);
    hipStream_t s = ctx->stream;
    Fq* cxy = p->commit_xy;
    uint8_t* cfl = p->commit_flags;
    uint32_t* closes = reinterpret_cast<uint32_t*>(p->den);

    PLONK_LAUNCH(transcript_kernel, dim3(tg), dim3(2 * TC_LANES), 0, s, 0, p->state, B, (const Fq*)cxy, (const uint8_t*)cfl, p->chal);
    // ---- round 1: coefficient forms of A, B, C, PI; commit A, B, C            prover.py:86-119
    const Fr n_inv = fp_inv(host_fr_u64((uint64_t)n));
    PLONK_CHECK_HIP(hipMemsetAsync(closes + B, 0, B * sizeof(uint32_t), s));
    PLONK_LAUNCH(gate_check_kernel, grid1(B * n), dim3(256), 0, s, (const Fr*)p->wit_lag, (const Fr*)p->pub, p->n_public,
                 p->sparse_pi ? (const Fr*)nullptr : (const Fr*)(p->wit_lag + 3 * B * n), (const Fr*)p->fixed_lag, n, B, closes + B);  // prover.py:108-116
    if (p->sparse_pi) {
        PLONK_TRY(ntt_run(ctx, p->wit_lag, p->coef, log_n, true, 3 * B, n, n, n, nullptr, nullptr, true));
        if (p->n_public)
            PLONK_LAUNCH(pi_coeffs_kernel, grid1(B * n), dim3(256), 0, s, (const Fr*)p->pub, p->n_public, p->roots_inv, n, B, n_inv,
                         p->coef + 3 * B * n);
        else
            PLONK_CHECK_HIP(hipMemsetAsync(p->coef + 3 * B * n, 0, B * n * sizeof(Fr), s));
    } else {
        PLONK_TRY(ntt_run(ctx, p->wit_lag, p->coef, log_n, true, 4 * B, n, n, n, nullptr, nullptr, true));
    }
    if (p->lag_srs) PLONK_TRY(msm_run_device(ctx, p->lag_srs, p->wit_lag, n, 3 * B, n, cxy, cfl));  // same points from Lagrange values
    else PLONK_TRY(msm_run_device(ctx, p->srs, p->coef, n, 3 * B, n, cxy, cfl));
    PLONK_LAUNCH(transcript_kernel, dim3(tg), dim3(2 * TC_LANES), 0, s, 1, p->state, B, (const Fq*)cxy, (const uint8_t*)cfl, p->chal);
    // ---- round 2: grand product Z, commit                                      prover.py:121-152
    GrandProductIn gp;
    for (int k = 0; k < 3; k++) {
        gp.abc[k] = p->wit_lag + (size_t)k * B * n;
        gp.sig[k] = p->fixed_lag + (FX_S1 + k) * n;
    }
    PLONK_LAUNCH(grand_product_kernel, dim3((unsigned)B), dim3(GP_THREADS), 0, s, gp, p->roots, (const ProofState*)p->state,
                 RoundChallenges{}, n, p->z_lag, closes, p->num, p->wz);  // num / wz: scratch until round 5
    PLONK_TRY(ntt_run(ctx, p->z_lag, p->coef + 4 * B * n, log_n, true, B, n, n, n, nullptr, nullptr, true));
    if (p->lag_srs) PLONK_TRY(msm_run_device(ctx, p->lag_srs, p->z_lag, n, B, n, cxy + 2 * 3 * B, cfl + 3 * B));
    else PLONK_TRY(msm_run_device(ctx, p->srs, p->coef + 4 * B * n, n, B, n, cxy + 2 * 3 * B, cfl + 3 * B));
    PLONK_LAUNCH(transcript_kernel, dim3(tg), dim3(2 * TC_LANES), 0, s, 2, p->state, B, (const Fq*)cxy, (const uint8_t*)cfl, p->chal);
    // ---- round 3: coset extensions, fused quotient, back to coefficients, commit T1..T3   prover.py:154-226
    // deg t < 3n: three cosets g mu^r H of the n-th roots of unity determine the quotient, so A, B, C, Z are evaluated on 3n
    // points — per coset a size-n transform of c_i (g mu^r)^i, on the 2^log_n kernel — instead of the reference's 4n, the
    // fused pass runs over 3n points, and three size-n inverse transforms + quotient_combine_kernel give T1, T2, T3
    const size_t n3 = QCOSETS * n;
    const NttFan fan{QCOSETS, 0u, (unsigned)n, (unsigned)n};  // one input, three scalings (g mu^r)^i, outputs [r][n] side by side
    if (p->sparse_pi) {  // A, B, C and Z through the transform, PI from the Lagrange basis on the coset
        PLONK_TRY(ntt_run(ctx, p->coef, p->big, log_n, false, 3 * B, n, n, n3, p->g_pow, nullptr, false, &fan));
        PLONK_TRY(ntt_run(ctx, p->coef + 4 * B * n, p->big + 4 * B * n3, log_n, false, B, n, n, n3, p->g_pow, nullptr, false, &fan));
    } else {
        PLONK_TRY(ntt_run(ctx, p->coef, p->big, log_n, false, 5 * B, n, n, n3, p->g_pow, nullptr, false, &fan));
    }
    if (p->sparse_pi) {
        if (p->n_public)
            PLONK_LAUNCH(pi_coset_kernel, grid1(B * n3), dim3(256), 0, s, (const Fr*)p->pub, p->n_public, (const Fr*)p->li_big, n3, B,
                         p->big + 3 * B * n3);
        else
            PLONK_CHECK_HIP(hipMemsetAsync(p->big + 3 * B * n3, 0, B * n3 * sizeof(Fr), s));
    }
    ZhInv zh;
    for (int k = 0; k < 4; k++) zh.v[k] = p->zh_inv[k < QCOSETS ? k : 0];
    QuotientIn qi;
    for (int k = 0; k < 5; k++) qi.wit[k] = p->big + (size_t)k * B * n3;
    for (int k = 0; k < FX_COUNT; k++) qi.fixed[k] = p->fixed_big + (size_t)k * n3;
    qi.l0 = p->l0_big;
    qi.xs = p->x_big;
    PLONK_LAUNCH(quotient_kernel, dim3((unsigned)((n3 + 255) / 256), (unsigned)B), dim3(256), 0, s, qi, zh, (const ProofState*)p->state, RoundChallenges{},
                 (unsigned)n3, p->quot, log_n, (unsigned)n4);
    const NttFan slices{QCOSETS, (unsigned)n, (unsigned)n, (unsigned)n};  // the three coset slices of every quotient row, in place
    PLONK_TRY(ntt_run(ctx, p->quot, p->quot, log_n, true, B, n, n4, n4, nullptr, p->ginv_pow, false, &slices));
    PLONK_LAUNCH(quotient_combine_kernel, grid1(B * n), dim3(256), 0, s, p->quot, n, n4, B, p->comb_i, p->comb_g1, p->comb_g2, fp_inv(host_fr_u64(2)));
    // T1..T3 = the three n-coefficient slices of each quotient row, one batched call (MSM k*B + b = slice k of proof b)
    PLONK_TRY(msm_run_device(ctx, p->srs, p->quot, n, 3 * B, n4, cxy + 2 * 4 * B, cfl + 4 * B, B, n));
    PLONK_LAUNCH(transcript_kernel, dim3(tg), dim3(2 * TC_LANES), 0, s, 3, p->state, B, (const Fq*)cxy, (const uint8_t*)cfl, p->chal);
    // ---- round 4: evaluations                                                  prover.py:228-239
    Fr w = host_root_of_unity(log_n, false);
    PLONK_LAUNCH(eval_kernel, dim3((unsigned)B), dim3(EV_THREADS), 0, s, (const Fr*)p->coef, (const Fr*)p->fixed_coef, w,
                 p->state, n, B);
    PLONK_LAUNCH(transcript_kernel, dim3(tg), dim3(2 * TC_LANES), 0, s, 4, p->state, B, (const Fq*)cxy, (const uint8_t*)cfl, p->chal);
    // ---- round 5: opening polynomials in coefficient form, commit              prover.py:241-306
    Fr ninv = fp_inv(host_fr_u64((uint64_t)n));
    unsigned gx = (unsigned)((n + 255) / 256);
    LinWeights* lw = p->lin_w;
    PLONK_LAUNCH(linearisation_weights_kernel, dim3(tb), dim3(64), 0, s, (const ProofState*)p->state, log_n, ninv, B, lw);
    PLONK_LAUNCH(linearisation_kernel, dim3(gx, (unsigned)B), dim3(256), 0, s, (const Fr*)p->coef,
                 (const Fr*)p->fixed_coef, (const Fr*)p->quot, (const LinWeights*)lw, log_n, B, p->num);
    PLONK_LAUNCH(divide_linear_kernel, dim3((unsigned)B), dim3(DV_THREADS), 0, s, (const Fr*)p->num, n, 0, w,
                 (const ProofState*)p->state, n, p->wz);
    PLONK_LAUNCH(divide_linear_kernel, dim3((unsigned)B), dim3(DV_THREADS), 0, s, (const Fr*)(p->coef + 4 * B * n), n, 1,
                 w, (const ProofState*)p->state, n, p->wz + B * n);
    PLONK_TRY(msm_run_device(ctx, p->srs, p->wz, n, 2 * B, n, cxy + 2 * 7 * B, cfl + 7 * B));
    PLONK_CHECK_HIP(hipGetLastError());
    return PLONK_OK;
}

// Synchronise and fetch: proofs [B][768], status[B] (0 ok; bit0: identity commitment; bit1: Z does
// not close to 1, i.e. the witness breaks the copy constraints — prover.py:132; bit2: a gate constraint fails on some row —
// prover.py:108-116; with bit1 clear that is exactly the condition of the reference's quotient-degree assert, prover.py:205-208).
static int prover_download(plonk_prover* p, size_t B, uint8_t* out_proofs, uint8_t* out_status, bool compressed);
int plonk_prover_download(plonk_prover* p, size_t B, uint8_t* out_proofs, uint8_t* out_status) {
    return prover_download(p, B, out_proofs, out_status, false);
}
// the same proofs as 480-byte records: nine compressed G1 points + six big-endian scalars (g1_codec.h)
int plonk_prover_download_compressed(plonk_prover* p, size_t B, uint8_t* out_proofs, uint8_t* out_status) {
    return prover_download(p, B, out_proofs, out_status, true);
}
static int prover_download(plonk_prover* p, size_t B, uint8_t* out_proofs, uint8_t* out_status, bool compressed) {
    PLONK_REQUIRE(p && B && out_proofs && out_status, PLONK_ERR_ARG, "bad argument");
    PLONK_REQUIRE(B == p->resident_b, PLONK_ERR_STATE, "download: batch %zu, but %zu witnesses are resident", B, p->resident_b);
    PLONK_ENTER(p->ctx);
    plonk_ctx* ctx = p->ctx;
    void* packed;
    PLONK_TRY(ctx_scratch(ctx, 2, B * 768, &packed));
    unsigned tb = (unsigned)((B + 63) / 64);
    PLONK_LAUNCH(pack_proofs_kernel, dim3(tb), dim3(64), 0, ctx->stream, (const Fq*)p->commit_xy,
                 (const ProofState*)p->state, B, (uint8_t*)packed, compressed ? 1 : 0);
    PLONK_CHECK_HIP(hipMemcpyAsync(out_proofs, packed, B * (compressed ? 480 : 768), hipMemcpyDeviceToHost, ctx->stream));
    std::vector<ProofState> st(B);
    std::vector<uint32_t> closes(2 * B);
    std::vector<uint8_t> flags(9 * B);
    PLONK_CHECK_HIP(hipMemcpyAsync(st.data(), p->state, B * sizeof(ProofState), hipMemcpyDeviceToHost, ctx->stream));
    PLONK_CHECK_HIP(hipMemcpyAsync(closes.data(), p->den, 2 * B * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    PLONK_CHECK_HIP(hipMemcpyAsync(flags.data(), p->commit_flags, 9 * B, hipMemcpyDeviceToHost, ctx->stream));
    unsigned long long bad_input = ~0ull;
    if (p->bad_input) PLONK_CHECK_HIP(hipMemcpyAsync(&bad_input, p->bad_input, sizeof bad_input, hipMemcpyDeviceToHost, ctx->stream));
    PLONK_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    for (size_t b = 0; b < B; b++) {
        uint8_t f = 0;
        if (bad_input != ~0ull && p->n_vars && bad_input / p->n_vars == b) f |= 8;  // an asynchronously uploaded value was >= r
        for (int slot = 0; slot < 9; slot++) f |= flags[(size_t)slot * B + b] ? 1 : 0;
        if (st[b].error) f |= 1;
        if (!closes[b]) f |= 2;
        if (closes[B + b]) f |= 4;
        out_status[b] = f;
    }
    return PLONK_OK;
}

// Debug / test access: the six challenges of proof b, canonical LE (beta, gamma, alpha, fft_cofactor, zeta, v)
int plonk_prover_challenges(plonk_prover* p, size_t b, uint8_t out_le32[6 * 32]) {
    PLONK_REQUIRE(p && b < p->resident_b && out_le32, PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(p->ctx);
    ProofState st;
    PLONK_CHECK_HIP(hipStreamSynchronize(p->ctx->stream));
    PLONK_CHECK_HIP(hipMemcpy(&st, p->state + b, sizeof st, hipMemcpyDeviceToHost));
    const Fr* ch[6] = {&st.beta, &st.gamma, &st.alpha, &st.fft_cofactor, &st.zeta, &st.v};
    for (int i = 0; i < 6; i++) {
        Fr c = fp_from_mont(*ch[i]);
        memcpy(out_le32 + 32 * i, c.v, 32);
    }
    return PLONK_OK;
}

}  // extern "C"

// ---- the fused round kernels on their own: what Prover.round_2 / round_3 of the reference-shaped API call
extern "C" {

// prover.py:121-146: Z_0 = 1, Z_{i+1} = Z_i * num_i / den_i from the wire values and the permutation polynomials
int plonk_fr_grand_product(plonk_ctx* ctx, const void* d_a, const void* d_b, const void* d_c, const void* d_s1, const void* d_s2,
                           const void* d_s3, unsigned log_n, const uint8_t beta_le32[32], const uint8_t gamma_le32[32], void* d_z_out,
                           int* out_closes) {
    PLONK_REQUIRE(ctx && d_a && d_b && d_c && d_s1 && d_s2 && d_s3 && beta_le32 && gamma_le32 && d_z_out && out_closes, PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    PLONK_REQUIRE(log_n <= PLONK_FR_TWO_ADICITY, PLONK_ERR_ARG, "size 2^%u exceeds the 2-adicity of Fr", log_n);
    PLONK_REQUIRE(le32_below_modulus(beta_le32, false) && le32_below_modulus(gamma_le32, false), PLONK_ERR_ARG, "challenge is not a canonical Fr value");
    const size_t n = (size_t)1 << log_n;
    const Fr* roots;
    PLONK_TRY(ntt_get_roots(ctx, log_n, false, &roots));
    void* scratch;
    PLONK_TRY(ctx_scratch(ctx, 2, (2 * n + 2) * sizeof(Fr), &scratch));  // num, den, and the closes flag
    Fr* num = (Fr*)scratch;
    uint32_t* closes = reinterpret_cast<uint32_t*>(num + 2 * n);
    GrandProductIn gp = {{(const Fr*)d_a, (const Fr*)d_b, (const Fr*)d_c}, {(const Fr*)d_s1, (const Fr*)d_s2, (const Fr*)d_s3}};
    RoundChallenges ch;
    ch.beta = fr_from_le32(beta_le32);
    ch.gamma = fr_from_le32(gamma_le32);
    ch.alpha = fp_zero<FrParams>();
    PLONK_LAUNCH(grand_product_kernel, dim3(1), dim3(GP_THREADS), 0, ctx->stream, gp, roots, (const ProofState*)nullptr, ch, n,
                 (Fr*)d_z_out, closes, num, num + n);
    PLONK_CHECK_HIP(hipGetLastError());
    uint32_t c = 0;
    PLONK_CHECK_HIP(hipMemcpyAsync(&c, closes, sizeof c, hipMemcpyDeviceToHost, ctx->stream));
    PLONK_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    *out_closes = (int)c;
    return PLONK_OK;
}

// prover.py:188-203: QUOT_big = (gate + alpha * permutation + alpha^2 * (Z - 1) L0) / Z_H on the 4n-point coset
// offset * mu^k.  d_evals = the coset extensions (fft_expand, 4n values each) of A, B, C, PI, Z, QL, QR, QM, QO, QC, S1,
// S2, S3, L0 in that order; X_big and 1 / Z_H (four distinct values) are derived from the offset here.
int plonk_fr_quotient(plonk_ctx* ctx, unsigned log_n, const void* const d_evals[14], const uint8_t offset_le32[32],
                      const uint8_t alpha_le32[32], const uint8_t beta_le32[32], const uint8_t gamma_le32[32], void* d_out) {
    PLONK_REQUIRE(ctx && d_evals && offset_le32 && alpha_le32 && beta_le32 && gamma_le32 && d_out, PLONK_ERR_ARG, "bad argument");
    for (int k = 0; k < 14; k++) PLONK_REQUIRE(d_evals[k], PLONK_ERR_ARG, "d_evals[%d] is NULL", k);
    PLONK_ENTER(ctx);
    PLONK_REQUIRE(log_n + 2 <= PLONK_FR_TWO_ADICITY, PLONK_ERR_ARG, "size 2^%u exceeds the 2-adicity of Fr", log_n + 2);
    PLONK_REQUIRE(le32_below_modulus(offset_le32, false) && le32_below_modulus(alpha_le32, false) && le32_below_modulus(beta_le32, false) &&
                      le32_below_modulus(gamma_le32, false), PLONK_ERR_ARG, "offset / challenge is not a canonical Fr value");
    const size_t n4 = (size_t)4 << log_n;
    const Fr off = fr_from_le32(offset_le32), one = fp_one<FrParams>();
    QuotientIn qi;
    for (int k = 0; k < 5; k++) qi.wit[k] = (const Fr*)d_evals[k];
    qi.fixed[FX_QL] = (const Fr*)d_evals[5];
    qi.fixed[FX_QR] = (const Fr*)d_evals[6];
    qi.fixed[FX_QM] = (const Fr*)d_evals[7];
    qi.fixed[FX_QO] = (const Fr*)d_evals[8];
    qi.fixed[FX_QC] = (const Fr*)d_evals[9];
    qi.fixed[FX_S1] = (const Fr*)d_evals[10];
    qi.fixed[FX_S2] = (const Fr*)d_evals[11];
    qi.fixed[FX_S3] = (const Fr*)d_evals[12];
    qi.l0 = (const Fr*)d_evals[13];
    PLONK_TRY(get_power_table(ctx, host_root_of_unity(log_n + 2, false), off, n4, &qi.xs));  // X_big[k] = offset * mu^k
    // Z_H(x_k) = (offset mu^k)^n - 1 = offset^n i^k - 1, i = mu^n: four values (prover.py:178); 1 / 0 == 0 as py_ecc
    Fr on = off;
    for (unsigned i = 0; i < log_n; i++) on = fp_sqr(on);
    const Fr i4 = host_root_of_unity(2, false);
    ZhInv zh;
    Fr cur = on;
    for (int k = 0; k < 4; k++) {
        zh.v[k] = fp_inv(fp_sub(cur, one));
        cur = fp_mul(cur, i4);
    }
    RoundChallenges ch;
    ch.alpha = fr_from_le32(alpha_le32);
    ch.beta = fr_from_le32(beta_le32);
    ch.gamma = fr_from_le32(gamma_le32);
    PLONK_LAUNCH(quotient_kernel, dim3((unsigned)((n4 + 255) / 256), 1), dim3(256), 0, ctx->stream, qi, zh, (const ProofState*)nullptr, ch, (unsigned)n4,
                 (Fr*)d_out, 0u, (unsigned)n4);
    PLONK_CHECK_HIP(hipGetLastError());
    return PLONK_OK;
}

}  // extern "C"

// Packs the resident batch's proofs (768-byte records, or 480-byte compressed ones) and status bytes into caller-owned
// DEVICE memory on the prover's own stream and records `done` there: the send side of plonk_gather_proofs_device.
int prover_pack_device(plonk_prover* p, size_t B, int compressed, uint8_t* d_proofs, uint8_t* d_status, hipEvent_t done) {
    PLONK_REQUIRE(p && B && d_proofs && d_status, PLONK_ERR_ARG, "bad argument");
    PLONK_REQUIRE(B == p->resident_b, PLONK_ERR_STATE, "gather: batch %zu, but %zu witnesses are resident", B, p->resident_b);
    plonk_ctx* ctx = p->ctx;
    const unsigned tb = (unsigned)((B + 63) / 64);
    PLONK_LAUNCH(pack_proofs_kernel, dim3(tb), dim3(64), 0, ctx->stream, (const Fq*)p->commit_xy, (const ProofState*)p->state, B, d_proofs,
                 compressed ? 1 : 0);
    PLONK_LAUNCH(pack_status_kernel, dim3(tb), dim3(64), 0, ctx->stream, (const ProofState*)p->state, (const uint32_t*)p->den,
                 (const uint8_t*)p->commit_flags, (const unsigned long long*)p->bad_input, p->n_vars, B, d_status);
    PLONK_CHECK_HIP(hipGetLastError());
    PLONK_CHECK_HIP(hipEventRecord(done, ctx->stream));
    return PLONK_OK;
}

plonk_ctx* prover_ctx(plonk_prover* p) { return p->ctx; }
