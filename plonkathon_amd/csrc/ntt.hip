// ntt.hip — BN254-Fr number-theoretic transform engine (forward / inverse / coset), natural order
// in and out, batched, any size 2^0 .. 2^28.
//
// Reference behaviour replaced: Polynomial.fft / ifft (/root/reference/poly.py:113-148, the
// recursive `_fft` at 117-127), to_coset_extended_lagrange (poly.py:156-163) and
// coset_extended_lagrange_to_coeffs (poly.py:169-177).  The transform is the plain DFT
// X[k] = sum_j x[j] w^(jk), w = 5^((r-1)/N) (curve.py:14-16); the inverse uses w^-1 and 1/N.
//
// Algorithm: N = R1*R2*..*RP (P <= 3 passes, Ri <= 2^10 when P > 1, <= 2^11 for
// a single pass).  Pass p runs Rp-point sub-transforms for a tile of C adjacent columns in LDS (tile
// of up to 4096 elements, stored as two 16-byte planes so unit-stride lanes are bank-conflict-free)
// as Stockham auto-sort levels of radix-8 register butterflies (3 LDS round trips for 2^11 points),
// with the Rp/2 twiddles w_Rp^k staged in LDS when they fit beside the tile, then multiplies by the
// inter-pass twiddle w_N^(H*j*k) (two-level table: one extra multiplication) on the way out.  Global accesses are chunks of C*32 B >= 128 B; the last pass
// reads whole rows and performs the digit-reversing write that restores natural order, so no
// separate transpose or bit-reversal kernel exists.  Coset scaling, zero padding, the 1/N factor
// and the inverse-coset scaling are fused into the first-pass load / last-pass store.
#include <string.h>

#include "plonk_internal.h"
#include "wave.h"
#include "fpl.h"

typedef FpL<FrParams> FrL;
typedef FpLS<FrParams> FrLS;

// defaults live in plonk_ctx (ntt_tile_log = 12: 4096 elements = 128 KiB of LDS; ntt_single_log = 11;
// ntt_radix_log = 10) and can be changed with plonk_ntt_configure for tuning / small-size tests.
#define NTT_TW_LO_LOG 10

// f(0) .. f(N-1) with compile-time arguments: the element array must never be indexed by a run-time value, or it moves
// from VGPRs to scratch memory (clang gives up unrolling loops whose bodies hold two field multiplications)
template <unsigned J> struct WaveIdx { static constexpr unsigned value = J; };
template <unsigned N, class F> PLONK_DEV void wave_for(F f) {
    if constexpr (N > 0) {
        wave_for<N - 1>(f);
        f(WaveIdx<N - 1>{});
    }
}

struct NttPass {
    const Fr* in;
    Fr* out;
    size_t in_bstride, out_bstride;
    unsigned log_n, log_r, log_c;
    unsigned log_h;  // product of the earlier passes' radices
    unsigned log_s;  // element stride of this pass's digit
    unsigned first, last;
    unsigned in_len;
    const Fr* small_tw;  // w_R^k, k < R/2 (global; staged into LDS when tw_in_lds)
    unsigned tw_in_lds;
    const Fr* tw_lo;
    const Fr* tw_hi;
    const Fr* in_scale;
    const Fr* out_scale;
    Fr out_scalar;
    unsigned has_out_scalar;
    unsigned nprev;
    unsigned prev_log_r[3];
    unsigned n_levels;
    unsigned level_radices;  // log2 radix of each in-LDS level, 2 bits per level: 3, 3, .., then 2 or 1
    Fr w8_1, w8_2, w8_3;     // w_8, w_8^2 (= w_4), w_8^3 for the transform direction
};

PLONK_DEV Fr lds_ld(const u32x4* lo, const u32x4* hi, unsigned i) {
    u32x4 a = lo[i], b = hi[i];
    Fr r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    return r;
}
PLONK_DEV void lds_st(u32x4* lo, u32x4* hi, unsigned i, const Fr& a) {
    lo[i] = u32x4{a.v[0], a.v[1], a.v[2], a.v[3]};
    hi[i] = u32x4{a.v[4], a.v[5], a.v[6], a.v[7]};
}

// 8/4/2-point DFTs in registers (decimation in frequency), outputs in natural frequency order.
PLONK_DEV void dft8(Fr x[8], const Fr& w1, const Fr& w2, const Fr& w3) {
    Fr a0 = fp_add(x[0], x[4]), a1 = fp_add(x[1], x[5]), a2 = fp_add(x[2], x[6]), a3 = fp_add(x[3], x[7]);
    Fr b0 = fp_sub(x[0], x[4]), b1 = fp_mul(fp_sub(x[1], x[5]), w1), b2 = fp_mul(fp_sub(x[2], x[6]), w2),
       b3 = fp_mul(fp_sub(x[3], x[7]), w3);
    Fr c0 = fp_add(a0, a2), c1 = fp_add(a1, a3), d0 = fp_sub(a0, a2), d1 = fp_mul(fp_sub(a1, a3), w2);
    Fr e0 = fp_add(b0, b2), e1 = fp_add(b1, b3), f0 = fp_sub(b0, b2), f1 = fp_mul(fp_sub(b1, b3), w2);
    x[0] = fp_add(c0, c1); x[4] = fp_sub(c0, c1); x[2] = fp_add(d0, d1); x[6] = fp_sub(d0, d1);
    x[1] = fp_add(e0, e1); x[5] = fp_sub(e0, e1); x[3] = fp_add(f0, f1); x[7] = fp_sub(f0, f1);
}
PLONK_DEV void dft4(Fr x[4], const Fr& w2) {
    Fr a0 = fp_add(x[0], x[2]), a1 = fp_add(x[1], x[3]), d0 = fp_sub(x[0], x[2]), d1 = fp_mul(fp_sub(x[1], x[3]), w2);
    x[0] = fp_add(a0, a1); x[2] = fp_sub(a0, a1); x[1] = fp_add(d0, d1); x[3] = fp_sub(d0, d1);
}
PLONK_DEV void dft2(Fr x[2]) {
    Fr s = fp_add(x[0], x[1]), d = fp_sub(x[0], x[1]);
    x[0] = s; x[1] = d;
}

// One pass: each workgroup transforms a tile of C columns x R points held in LDS with the Stockham
// auto-sort recurrence in radix-8 (then 4 / 2) register butterflies:
//   level with sub-length n, stride s, radix rho, m = n/rho; group (p, q), p < m, q < s:
//     inputs   x[q + s (p + j m)]         (= g + j R/rho for the flat group index g = p s + q)
//     outputs  y[q + s (rho p + j')] = w_n^(p j') * DFT_rho(inputs)[j']
// The first level reads its inputs straight from HBM, the last level writes straight to HBM, so a
// 2^11-point transform makes 3 LDS round trips (11 in a radix-2 formulation) and ends in natural order.
__global__ void __launch_bounds__(512) ntt_pass_stockham_kernel(NttPass p) {
    PLONK_DYN_SMEM(smem);
    const unsigned R = 1u << p.log_r, C = 1u << p.log_c;
    const unsigned T = R * C;
    // LDS index of tile element (i, c): column-interleaved for strided passes, padded rows for the
    // row (last) pass so both the i-fastest and the c-fastest phases are conflict-free.
    const unsigned row_pitch = p.last ? (R + (C > 1 ? 1 : 0)) : 0;
    const unsigned t_pad = p.last ? row_pitch * C : T;
    u32x4* d_lo = reinterpret_cast<u32x4*>(smem);
    u32x4* d_hi = d_lo + t_pad;
    u32x4* w_lo = d_hi + t_pad;
    u32x4* w_hi = w_lo + (R / 2 ? R / 2 : 1);
#define LIDX(i, c) (p.last ? ((c) * row_pitch + (i)) : (((i) << p.log_c) + (c)))

    const unsigned tid = threadIdx.x, nthr = blockDim.x;
    const Fr* in = p.in + (size_t)blockIdx.y * p.in_bstride;
    Fr* out = p.out + (size_t)blockIdx.y * p.out_bstride;
    const unsigned tile = blockIdx.x;

    // tile coordinates
    unsigned cb = 0, kb = 0, rest = 0, log_r1 = 0;
    size_t base = 0;
    if (!p.last) {
        const unsigned log_tiles_per_hi = p.log_s - p.log_c;
        const unsigned hi_idx = tile >> log_tiles_per_hi;
        cb = tile & ((1u << log_tiles_per_hi) - 1);
        base = ((size_t)hi_idx << (p.log_n - p.log_h)) + ((size_t)cb << p.log_c);
    } else if (p.nprev) {
        log_r1 = p.prev_log_r[0];
        const unsigned log_kb = log_r1 - p.log_c;
        kb = tile & ((1u << log_kb) - 1);
        rest = tile >> log_kb;
    }
    unsigned rev_rest = 0;
    if (p.last && p.nprev > 1) {
        // rest = (k2, .., k_{P-1}) with k_{P-1} least significant; output weight of k_q is R1*..*R_{q-1}
        unsigned rr = rest, weight = p.prev_log_r[0];
        unsigned w_of[3] = {0, 0, 0};
        for (unsigned q = 1; q < p.nprev; q++) { w_of[q] = weight; weight += p.prev_log_r[q]; }
        for (int q = (int)p.nprev - 1; q >= 1; q--) {
            unsigned kq = rr & ((1u << p.prev_log_r[q]) - 1);
            rr >>= p.prev_log_r[q];
            rev_rest += kq << w_of[q];
        }
    }

    if (p.tw_in_lds) {
        for (unsigned i = tid; i < R / 2; i += nthr) lds_st(w_lo, w_hi, i, fp_load(p.small_tw + i));
    }

    // global element index of tile position (i, c) on the input side
    auto in_index = [&](unsigned i, unsigned c) PLONK_LAMBDA_INLINE -> size_t {
        if (!p.last) return base + ((size_t)i << p.log_s) + c;
        size_t row = p.nprev ? ((((size_t)(kb << p.log_c) + c) << (p.log_h - log_r1)) + rest) : 0;
        return (row << p.log_r) + i;
    };
    auto load_in = [&](unsigned i, unsigned c) PLONK_LAMBDA_INLINE -> Fr {
        const size_t g = in_index(i, c);
        if (p.first && g >= p.in_len) return fp_zero<FrParams>();
        Fr v = fp_load(in + g);
        if (p.first && p.in_scale) v = fp_mul(v, fp_load(p.in_scale + g));
        return v;
    };
    auto store_out = [&](unsigned k, unsigned c, Fr v) PLONK_LAMBDA_INLINE {
        if (!p.last) {
            const size_t jrest = ((size_t)cb << p.log_c) + c;
            const size_t ex = (jrest * k) << p.log_h;  // < N
            if (ex) {
                Fr tw = fp_load(p.tw_lo + (ex & ((1u << NTT_TW_LO_LOG) - 1)));
                if (p.log_n > NTT_TW_LO_LOG) tw = fp_mul(tw, fp_load(p.tw_hi + (ex >> NTT_TW_LO_LOG)));
                v = fp_mul(v, tw);
            }
            fp_store(out + base + ((size_t)k << p.log_s) + c, v);
        } else {
            size_t o = p.nprev ? (((size_t)(kb << p.log_c) + c) + rev_rest + ((size_t)k << p.log_h)) : k;
            if (p.out_scale) v = fp_mul(v, fp_load(p.out_scale + o));
            if (p.has_out_scalar) v = fp_mul(v, p.out_scalar);
            fp_store(out + o, v);
        }
    };
    // w_R^e for any e (table holds e < R/2; w^(e + R/2) = -w^e)
    auto twiddle = [&](unsigned e) PLONK_LAMBDA_INLINE -> Fr {
        e &= R - 1;
        const unsigned idx = e & (R / 2 - 1);
        Fr t = p.tw_in_lds ? lds_ld(w_lo, w_hi, idx) : fp_load(p.small_tw + idx);
        return (e & (R / 2)) ? fp_neg(t) : t;
    };

    if (p.n_levels == 0) {  // R == 1: a pure (scaled) copy
        for (unsigned e = tid; e < T; e += nthr) store_out(0, e, load_in(0, e));
        return;
    }
    if (p.tw_in_lds) __syncthreads();

    const Fr w8_1 = p.w8_1, w8_2 = p.w8_2, w8_3 = p.w8_3;
    unsigned log_nn = p.log_r;  // log2 of the current sub-transform length n
    unsigned log_ss = 0;        // log2 of the current stride s
    for (unsigned lev = 0; lev < p.n_levels; lev++) {
        const unsigned lr = (p.level_radices >> (2 * lev)) & 3u, rho = 1u << lr;
        const bool from_global = lev == 0, to_global = lev + 1 == p.n_levels;
        const unsigned groups_per_col = R >> lr, n_groups = T >> lr;
        const unsigned log_m = log_nn - lr;
        // column-fastest thread order keeps HBM chunks and LDS rows contiguous; the row (last) pass
        // walks i-fastest while it reads and c-fastest when it finally writes to HBM
        const bool c_fastest = !p.last || to_global;
        Fr x[8];
        // A thread may own several groups per level (radix < 8, or tiny tiles); when the level reads
        // LDS and writes LDS, all reads of the level must finish before any write: two sweeps.
        const unsigned sweeps = (n_groups + nthr - 1) / nthr;
        // (only the LAST level can have radix < 8, hence several sweeps, and it writes to HBM, so no
        //  LDS->LDS level ever overwrites inputs another sweep still has to read)
        for (unsigned sw = 0; sw < sweeps; sw++) {
            const unsigned gid = sw * nthr + tid;
            const bool active = gid < n_groups;
            unsigned c = 0, g = 0;
            if (active) {
                if (c_fastest) { c = gid & (C - 1); g = gid >> p.log_c; }
                else { g = gid & (groups_per_col - 1); c = gid >> (p.log_r - lr); }
                wave_for<8>([&](auto J) {  // compile-time expansion: `#pragma unroll` gives up on bodies this large and x[] would move to scratch
                    constexpr unsigned j = decltype(J)::value;
                    if (j < rho) {
                        const unsigned i = g + j * groups_per_col;
                        x[j] = from_global ? load_in(i, c) : lds_ld(d_lo, d_hi, LIDX(i, c));
                    }
                });
            }
            if (!from_global) __syncthreads();  // every lane has its inputs in registers
            if (active) {
                if (lr == 3) dft8(x, w8_1, w8_2, w8_3);
                else if (lr == 2) dft4(x, w8_2);
                else dft2(x);
                const unsigned pp = g >> log_ss, q = g & ((1u << log_ss) - 1);
                const unsigned tw_scale = p.log_r - log_nn;  // w_n^(p j') = w_R^((p j') << tw_scale)
                wave_for<8>([&](auto J) {
                    constexpr unsigned j = decltype(J)::value;
                    if (j < rho) {
                        Fr v = x[j];
                        if (j && pp) v = fp_mul(v, twiddle((pp * j) << tw_scale));
                        const unsigned o = q + (((pp << lr) + j) << log_ss);
                        if (to_global) store_out(o, c, v);
                        else lds_st(d_lo, d_hi, LIDX(o, c), v);
                    }
                });
            }
            if (!to_global) __syncthreads();  // outputs visible before the next sweep / level reads
        }
        (void)log_m;
        log_nn -= lr;
        log_ss += lr;
    }
#undef LIDX
}

// ------------------------------------------------------------------------------------------------
// Variant A: radix-2 Gentleman-Sande stages, one LDS round trip + barrier per stage, bit-reversed read
// at the end.  Tiny register footprint (40 VGPRs, 8 waves/SIMD).  plonk_ntt_configure(kind = 1).
PLONK_DEV unsigned bitrev(unsigned x, unsigned bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

__global__ void __launch_bounds__(1024) ntt_pass_radix2_kernel(NttPass p) {
    PLONK_DYN_SMEM(smem);
    const unsigned R = 1u << p.log_r, C = 1u << p.log_c;
    const unsigned T = R * C;
    // LDS index of tile element (r, c): column-interleaved for strided passes, padded rows for the
    // row (last) pass so both the r-fastest and the c-fastest phases are conflict-free.
    const unsigned row_pitch = p.last ? (R + (C > 1 ? 1 : 0)) : 0;
    const unsigned t_pad = p.last ? row_pitch * C : T;
    u32x4* d_lo = reinterpret_cast<u32x4*>(smem);
    u32x4* d_hi = d_lo + t_pad;
    u32x4* w_lo = d_hi + t_pad;
    u32x4* w_hi = w_lo + (R / 2 ? R / 2 : 1);
#define LIDX(r, c) (p.last ? ((c) * row_pitch + (r)) : (((r) << p.log_c) + (c)))

    const unsigned tid = threadIdx.x, nthr = blockDim.x;
    const Fr* in = p.in + (size_t)blockIdx.y * p.in_bstride;
    Fr* out = p.out + (size_t)blockIdx.y * p.out_bstride;
    const unsigned tile = blockIdx.x;

    // tile coordinates
    unsigned hi_idx = 0, cb = 0, kb = 0, rest = 0, log_r1 = 0;
    size_t base = 0;
    if (!p.last) {
        const unsigned log_tiles_per_hi = p.log_s - p.log_c;
        hi_idx = tile >> log_tiles_per_hi;
        cb = tile & ((1u << log_tiles_per_hi) - 1);
        base = ((size_t)hi_idx << (p.log_n - p.log_h)) + ((size_t)cb << p.log_c);
    } else if (p.nprev) {
        log_r1 = p.prev_log_r[0];
        const unsigned log_kb = log_r1 - p.log_c;
        kb = tile & ((1u << log_kb) - 1);
        rest = tile >> log_kb;
    }
    unsigned rev_rest = 0;
    if (p.last && p.nprev > 1) {
        unsigned rr = rest, weight = p.prev_log_r[0];
        // rest = (k2, .., k_{P-1}) with k_{P-1} least significant; output weight of k_q is R1*..*R_{q-1}
        unsigned w_of[3] = {0, 0, 0};
        for (unsigned q = 1; q < p.nprev; q++) { w_of[q] = weight; weight += p.prev_log_r[q]; }
        for (int q = (int)p.nprev - 1; q >= 1; q--) {
            unsigned kq = rr & ((1u << p.prev_log_r[q]) - 1);
            rr >>= p.prev_log_r[q];
            rev_rest += kq << w_of[q];
        }
    }

    // tile element (r, c) <- global memory (zero padding and the input scaling of the first pass fused in)
    auto gload = [&](unsigned r, unsigned c) -> Fr {
        size_t g;
        if (!p.last) {
            g = base + ((size_t)r << p.log_s) + c;
        } else {
            size_t row = p.nprev ? ((((size_t)(kb << p.log_c) + c) << (p.log_h - log_r1)) + rest) : 0;
            g = (row << p.log_r) + r;
        }
        if (p.first && g >= p.in_len) return fp_zero<FrParams>();
        Fr v = fp_load(in + g);
        if (p.first && p.in_scale) v = fp_mul(v, fp_load(p.in_scale + g));
        return v;
    };
    // frequency k of column c -> global memory (inter-pass twiddle, or the output scalings of the last pass)
    auto gstore = [&](unsigned k, unsigned c, Fr v) {
        if (!p.last) {
            const size_t jrest = ((size_t)cb << p.log_c) + c;
            const size_t ex = (jrest * k) << p.log_h;  // < N
            if (ex) {
                Fr tw = fp_load(p.tw_lo + (ex & ((1u << NTT_TW_LO_LOG) - 1)));
                if (p.log_n > NTT_TW_LO_LOG) tw = fp_mul(tw, fp_load(p.tw_hi + (ex >> NTT_TW_LO_LOG)));
                v = fp_mul(v, tw);
            }
            fp_store(out + base + ((size_t)k << p.log_s) + c, v);
        } else {
            size_t o = p.nprev ? (((size_t)(kb << p.log_c) + c) + rev_rest + ((size_t)k << p.log_h)) : k;
            if (p.out_scale) v = fp_mul(v, fp_load(p.out_scale + o));
            if (p.has_out_scalar) v = fp_mul(v, p.out_scalar);
            fp_store(out + o, v);
        }
    };

    // stage the pass's small twiddles
    for (unsigned i = tid; i < R / 2; i += nthr) lds_st(w_lo, w_hi, i, fp_load(p.small_tw + i));

    // With at least four levels the first stage reads its operands straight from HBM and the last one writes
    // its results straight to HBM (two LDS round trips and two barriers fewer per pass); tiny tiles keep the
    // plain load / stages / store sequence.
    const bool fuse = p.log_r >= 4;
    if (!fuse) {
        for (unsigned e = tid; e < T; e += nthr) {
            unsigned r, c;
            if (!p.last) {
                c = e & (C - 1);
                r = e >> p.log_c;
            } else {
                r = e & (R - 1);
                c = e >> p.log_r;
            }
            lds_st(d_lo, d_hi, LIDX(r, c), gload(r, c));
        }
    }
    __syncthreads();

    // decimation-in-frequency stages, (a, b) -> (a + b, (a - b) * w).  Two levels (half sizes h and h/2) are
    // done per LDS round trip: a thread owns rows {r0, r0 + h/2, r0 + h, r0 + 3h/2} of one column, which
    // halves the LDS traffic and the barrier count of a level-at-a-time schedule; an odd level count starts
    // with one plain radix-2 level.
    int lh = (int)p.log_r - 1;
    bool from_global = fuse;
    if (p.log_r & 1) {
        const unsigned h = 1u << lh;
        for (unsigned b = tid; b < T / 2; b += nthr) {
            unsigned c, bf;
            if (!p.last) {
                c = b & (C - 1);
                bf = b >> p.log_c;
            } else {
                bf = b & (R / 2 - 1);
                c = b >> (p.log_r - 1);
            }
            const unsigned off = bf & (h - 1);
            const unsigned r0 = ((bf >> lh) << (lh + 1)) + off;
            const unsigned i0 = LIDX(r0, c), i1 = LIDX(r0 + h, c);
            const Fr x = from_global ? gload(r0, c) : lds_ld(d_lo, d_hi, i0);
            const Fr y = from_global ? gload(r0 + h, c) : lds_ld(d_lo, d_hi, i1);
            Fr s = fp_add(x, y), d = fp_sub(x, y);
            if (lh != 0) d = fp_mul(d, lds_ld(w_lo, w_hi, off << (p.log_r - 1 - lh)));
            lds_st(d_lo, d_hi, i0, s);
            lds_st(d_lo, d_hi, i1, d);
        }
        __syncthreads();
        lh--;
        from_global = false;
    }
    for (; lh >= 1; lh -= 2) {
        const unsigned h = 1u << lh, q = h >> 1;
        const unsigned sh_a = p.log_r - 1 - lh;  // twiddle index shift of the level with half size h
        const bool to_global = fuse && lh == 1;  // rows r0 .. r0 + 3 hold frequencies bitrev(r0 + j)
        // the row pass walks rows fastest while it reads HBM and columns fastest when it finally writes
        const bool c_fastest = !p.last || to_global;
        for (unsigned b = tid; b < T / 4; b += nthr) {
            unsigned c, bf;
            if (c_fastest) {
                c = b & (C - 1);
                bf = b >> p.log_c;
            } else {
                bf = b & (R / 4 - 1);
                c = b >> (p.log_r - 2);
            }
            const unsigned off = bf & (q - 1);
            const unsigned r0 = ((bf >> (lh - 1)) << (lh + 1)) + off;
            const unsigned i0 = LIDX(r0, c), i1 = LIDX(r0 + q, c), i2 = LIDX(r0 + h, c), i3 = LIDX(r0 + h + q, c);
            Fr x0, x1, x2, x3;
            if (from_global) {
                x0 = gload(r0, c);
                x1 = gload(r0 + q, c);
                x2 = gload(r0 + h, c);
                x3 = gload(r0 + h + q, c);
            } else {
                x0 = lds_ld(d_lo, d_hi, i0);
                x1 = lds_ld(d_lo, d_hi, i1);
                x2 = lds_ld(d_lo, d_hi, i2);
                x3 = lds_ld(d_lo, d_hi, i3);
            }
            const Fr s0 = fp_add(x0, x2), s1 = fp_add(x1, x3);
            const Fr d0 = fp_mul(fp_sub(x0, x2), lds_ld(w_lo, w_hi, off << sh_a));
            const Fr d1 = fp_mul(fp_sub(x1, x3), lds_ld(w_lo, w_hi, (off + q) << sh_a));
            Fr y1 = fp_sub(s0, s1), y3 = fp_sub(d0, d1);
            if (lh != 1) {  // the level with half size 1 has unit twiddles
                const Fr wb = lds_ld(w_lo, w_hi, off << (sh_a + 1));
                y1 = fp_mul(y1, wb);
                y3 = fp_mul(y3, wb);
            }
            const Fr y0 = fp_add(s0, s1), y2 = fp_add(d0, d1);
            if (to_global) {
                gstore(bitrev(r0, p.log_r), c, y0);
                gstore(bitrev(r0 + 1, p.log_r), c, y1);
                gstore(bitrev(r0 + 2, p.log_r), c, y2);
                gstore(bitrev(r0 + 3, p.log_r), c, y3);
            } else {
                lds_st(d_lo, d_hi, i0, y0);
                lds_st(d_lo, d_hi, i1, y1);
                lds_st(d_lo, d_hi, i2, y2);
                lds_st(d_lo, d_hi, i3, y3);
            }
        }
        if (!to_global) __syncthreads();
        from_global = false;
    }

    if (!fuse) {  // store: frequency k of column c sits at row bitrev(k)
        for (unsigned e = tid; e < T; e += nthr) {
            const unsigned c = e & (C - 1);
            const unsigned k = e >> p.log_c;
            gstore(k, c, lds_ld(d_lo, d_hi, LIDX(bitrev(k, p.log_r), c)));
        }
    }
#undef LIDX
}

// ------------------------------------------------------------------------------------------------
// Variant C ("wave" kernels): the whole transform in registers, exchanges INSIDE a wave by cross-lane moves.
// The default wherever it applies: every size 2^8 .. 2^13 in one launch, 2^16 .. 2^26 as two passes of those.
//
// N = 2^(LOG_E + 6 + 2 L) points, L = 0, 1, 2; N / E threads (64, 256, 1024), each holding E = 2^LOG_E elements in
// registers as 9 signed 29-bit limbs (fpl.h) from the first load to the last store:
//   E = 8: N = 2^9, 2^11, 2^13   radix 8, L x radix 4, radix 8, radix 8     (3 waves per SIMD; 4 at 1024 threads)
//   E = 4: N = 2^8, 2^10, 2^12   radix 4, L x radix 4, 3 x radix 4          (half the registers: 4+ waves per SIMD, and
//                                 twice the workgroups for a lone transform — round 3)
// Decimation in frequency by digits.  The bits of the element index live in three places — the register index (LOG_E
// bits), the lane (6 bits) and, for L > 0, the wave (2 L bits).  A stage works on the digit currently held in the
// register index; between stages that digit is swapped with
//   * two WAVE bits: the only exchange that needs LDS, in rounds of 4 elements per thread (36 B x 4 x threads);
//   * LANE bits: single-bit swaps of a register-index bit with a lane bit.  lane ^ 32 and lane ^ 16 are ONE instruction
//     per pair of words (v_permlane32_swap / v_permlane16_swap exchange exactly the halves / rows a bit swap trades);
//     lane ^ 1, 2, 8 are two selects whose moved operand comes through DPP (quad_perm / row_ror); lane ^ 4 goes
//     through ds_swizzle — no LDS memory, no barrier for the last six levels of every transform.
// After a stage on a digit of a sub-transform of size S (remaining points indexed by `low`), output f is multiplied by
// w_S^(low f) = roots[(N / S) low f]  (the Cooley-Tukey twiddle between the digit DFT and the remaining sub-transforms);
// the root table is stored AS LIMBS (12 words per entry: no unpacking in the loop).  Outputs appear at frequency
// k = d_A + r_A d_B + ... (first digit least significant), which the final store turns into a natural-order write.
// Coset scaling, zero padding n -> 4n, 1/N and the inverse-coset scaling are fused into the first load / last store.
//
// Range discipline (m = the modulus; "N-form" = limbs 0..7 in [0, 2^29), limb 8 signed and small).  Every stage receives
// N-form elements with |value| < 2.8 m: loads are canonical, twiddle multiplications (fpl_mul_shoup: the factor is a known
// constant, so the product needs 143 multiply-adds instead of fpl_mul's 171) return N-form in (-1.8 m, 2.8 m), and the one
// output of each butterfly group that carries no twiddle factor (index 0) goes through fpl_reduce_small (N-form,
// |value| < 0.51 m).  Inside a radix-8 butterfly five carry sweeps keep every limb inside int32 and every multiplicand
// inside the multiplications' operand bound (|limb| < 1.27 * 2^30): the bounds are written on each line of dft8l / dft4l.
// |value| never exceeds 22.4 m (a sum of eight inputs): fpl_reduce_small's table reaches 23 m, the multiplications 128 m.
#define NTT_LIMB_STRIDE 12   // int32 words per entry of a limb-form table (9 used): Montgomery residues (inter-pass twiddles)
#define NTT_SHOUP_STRIDE 20  // int32 words per entry of a root table: w (9), floor(w 2^261 / m) (9), 2 unused
struct NttWave {
    const Fr* in;
    Fr* out;
    size_t in_bstride, out_bstride;
    unsigned in_len;
    // mode 0: the whole transform, blockIdx.x = batch index.  Two-pass transforms N = R1 R2 (index i1 R2 + c -> frequency
    // k1 + R1 k2): mode 1 = R1-point transforms down the R2 columns (element i1 of column c at in[i1 R2 + c], output k1
    // times w_N^(c k1) to out[k1 R2 + c]); mode 2 = R2-point transforms along the R1 rows (row k1 at in[k1 R2 ..],
    // output k2 to out[k1 + R1 k2]).  blockIdx.x = column / row (XCD-aware order), blockIdx.y = batch index.
    unsigned mode, log_n, log_other;  // log2 of the whole transform and of the stride between successive positions
    // Distributed (multi-GPU) transforms run the same two passes on a slice: rank g of W owns R2 / W columns for the
    // column pass (twiddle column = sub + sub_base) and R1 / W rows for the row pass, whose input arrives from the
    // all-to-all as W chunks [source rank][local row][source's columns]: position c of a row sits at
    // (c >> chunk_log) * chunk_stride + row * 2^chunk_log + (c & (2^chunk_log - 1)).  chunk_log = 0 means contiguous rows.
    unsigned sub_base, chunk_log, chunk_stride;
    const int32_t* tw_lo;  // inter-pass twiddles w_N^e = tw_lo[e & 1023] * tw_hi[e >> 10], Shoup pairs: applied one after the other (mode 1)
    const int32_t* tw_hi;  //   (for an inverse transform tw_hi carries the factor 1/N as well: tw_always)
    unsigned tw_always;    // multiply even when e == 0 (tw_hi[0] = 1/N)
    const int32_t* roots;  // the twiddles of this kernel's transform size and direction, Shoup pairs in program order (wavel_tw_*)
    const Fr* in_scale;    // per-element factor at load (coset offset powers) or null
    const Fr* out_scale;   // per-element factor at store or null
    Fr out_scalar;
    unsigned has_out_scalar;
    FrLS w8[3];            // w_8, w_8^2 (= w_4), w_8^3 for the transform direction, Shoup pairs: kernel arguments live in SGPRs
    const int32_t* jm;     // fpl_reduce_small's table of j * m
};

// entry idx of a limb-form table (Montgomery residue)
PLONK_DEV FrL wavel_ld_tw(const int32_t* tab, unsigned idx) {
    // (a 32-bit byte offset from a uniform base: SGPR-base addressing, one VGPR per address instead of two)
    const int32_t* t = reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(tab) + idx * (unsigned)(NTT_LIMB_STRIDE * sizeof(int32_t)));
    const u32x4 a = *reinterpret_cast<const u32x4*>(t), b = *reinterpret_cast<const u32x4*>(t + 4);
    FrL r;
    r.l[0] = (int32_t)a.x; r.l[1] = (int32_t)a.y; r.l[2] = (int32_t)a.z; r.l[3] = (int32_t)a.w;
    r.l[4] = (int32_t)b.x; r.l[5] = (int32_t)b.y; r.l[6] = (int32_t)b.z; r.l[7] = (int32_t)b.w;
    r.l[8] = t[8];
#pragma unroll
    for (int i = 0; i < 9; i++) FPL_ANY_SIGN(r.l[i]);
    return r;
}
// entry idx of a root table: the Shoup pair of w^idx, 80 bytes as five 16-byte loads
PLONK_DEV FrLS wavel_ld_root(const int32_t* tab, unsigned idx) {
    const u32x4* t = reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(tab) + idx * (unsigned)(NTT_SHOUP_STRIDE * sizeof(int32_t)));
    const u32x4 a = t[0], b = t[1], c = t[2], d = t[3], e = t[4];
    FrLS r;
    r.w[0] = (int32_t)a.x; r.w[1] = (int32_t)a.y; r.w[2] = (int32_t)a.z; r.w[3] = (int32_t)a.w;
    r.w[4] = (int32_t)b.x; r.w[5] = (int32_t)b.y; r.w[6] = (int32_t)b.z; r.w[7] = (int32_t)b.w;
    r.w[8] = (int32_t)c.x; r.wp[0] = (int32_t)c.y; r.wp[1] = (int32_t)c.z; r.wp[2] = (int32_t)c.w;
    r.wp[3] = (int32_t)d.x; r.wp[4] = (int32_t)d.y; r.wp[5] = (int32_t)d.z; r.wp[6] = (int32_t)d.w;
    r.wp[7] = (int32_t)e.x; r.wp[8] = (int32_t)e.y;
    return r;
}

// swap register-index bit RB with the lane bit of MASK: lanes with the bit clear keep x[r] and trade x[r | 1 << RB],
// lanes with the bit set keep x[r | 1 << RB] and trade x[r]
template <unsigned E, unsigned RB, unsigned MASK> PLONK_DEV void wavel_swap_bit(FrL (&x)[E], unsigned lane) {
    const bool hi = (lane & MASK) != 0;
    wave_for<E / 2>([&](auto I) {
        constexpr unsigned ih = decltype(I)::value;
        constexpr unsigned r = ((ih >> RB) << (RB + 1)) | (ih & ((1u << RB) - 1)), r1 = r | (1u << RB);  // the indices with bit RB clear
        wave_for<9>([&](auto W) {
            constexpr unsigned i = decltype(W)::value;
            wave_swap_words<MASK>(x[r].l[i], x[r1].l[i], hi, lane);
        });
    });
}

// ---- twiddle tables in the order the kernel consumes them ("program order") ----------------------------------------------
// Stage s of a wave kernel multiplies register f (f = 1 .. count) by w_R^(low f mult), low < nb: the table holds, stage after
// stage and f after f, one BLOCK of nb Shoup pairs indexed by low — stored as five planes of nb x 16 bytes, so that a load
// instruction of 64 lanes with consecutive `low` reads 1 KB of consecutive bytes (8 cache lines).  The natural-order table
// (80-byte entries at index low f mult) made every one of the five loads of a twiddle touch 40 .. 120 different lines and
// use a fifth to a fifteenth of each: at 2^10 .. 2^13, whose tables do not fit the 32 KB L1, that was ~0.5 MB of L2 -> L1
// traffic per 64 KB transform.  Entries: ~N per kernel size (1020 at 2^10, 2040 at 2^11).
//   twiddled stages: A (digit in the registers at load), the L wave-bit stages, the lane stages except the last
PLONK_HD constexpr unsigned wavel_tw_stages(unsigned log_e, unsigned nlds) { return 1 + nlds + (log_e == 3 ? 1 : 2); }
PLONK_HD constexpr unsigned wavel_tw_nb(unsigned log_e, unsigned nlds, unsigned s) {  // distinct values of `low` in stage s
    if (s == 0) return 64u << (2 * nlds);
    if (s <= nlds) return 1u << (6 + 2 * (nlds - s));
    return log_e == 3 ? 8u : (s == nlds + 1 ? 16u : 4u);
}
PLONK_HD constexpr unsigned wavel_tw_count(unsigned log_e, unsigned nlds, unsigned s) {  // factors f = 1 .. count
    return (s >= 1 && s <= nlds) ? 3u : (1u << log_e) - 1;
}
PLONK_HD constexpr unsigned wavel_tw_mult(unsigned log_e, unsigned nlds, unsigned s) {  // N / S of stage s
    const unsigned log_n = log_e + 6 + 2 * nlds;
    if (s == 0) return 1;
    if (s <= nlds) return 1u << (log_n - (6 + 2 * (nlds - s) + 2));
    return log_e == 3 ? 1u << (log_n - 6) : (s == nlds + 1 ? 1u << (log_n - 6) : 1u << (log_n - 4));
}
PLONK_HD constexpr unsigned wavel_tw_offset(unsigned log_e, unsigned nlds, unsigned s) {  // first entry of stage s's blocks
    unsigned o = 0;
    for (unsigned t = 0; t < s; t++) o += wavel_tw_nb(log_e, nlds, t) * wavel_tw_count(log_e, nlds, t);
    return o;
}
#define NTT_PLANE_WORDS 4  // a plane holds 16 bytes of every entry of its block; five planes per block

// entry `low` of a block of nb entries
PLONK_DEV FrLS wavel_ld_root_planar(const int32_t* block, unsigned nb, unsigned low) {
    // (each plane as "uniform base + 32-bit lane offset": SGPR-base addressing, no 64-bit address arithmetic per lane)
    const unsigned off = low * 16u;
    const auto plane = [&](unsigned pl) PLONK_LAMBDA_INLINE {
        return *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(block + pl * nb * NTT_PLANE_WORDS) + off);
    };
    const u32x4 a = plane(0), b = plane(1), c = plane(2), d = plane(3), e = plane(4);
    FrLS r;
    r.w[0] = (int32_t)a.x; r.w[1] = (int32_t)a.y; r.w[2] = (int32_t)a.z; r.w[3] = (int32_t)a.w;
    r.w[4] = (int32_t)b.x; r.w[5] = (int32_t)b.y; r.w[6] = (int32_t)b.z; r.w[7] = (int32_t)b.w;
    r.w[8] = (int32_t)c.x; r.wp[0] = (int32_t)c.y; r.wp[1] = (int32_t)c.z; r.wp[2] = (int32_t)c.w;
    r.wp[3] = (int32_t)d.x; r.wp[4] = (int32_t)d.y; r.wp[5] = (int32_t)d.z; r.wp[6] = (int32_t)d.w;
    r.wp[7] = (int32_t)e.x; r.wp[8] = (int32_t)e.y;
    return r;
}

// the kernels compiled for 128 VGPRs with 8 elements per thread (WavelCfg::TIGHT): 2^13 and 2^11
#ifdef PLONK_NTT_W11_3
#define WAVEL_TIGHT_LOG_N(log_n) ((log_n) == 13)
#else
#define WAVEL_TIGHT_LOG_N(log_n) ((log_n) == 13 || (log_n) == 11 || (log_n) == 9)
#endif
// x[BASE + f] *= w^(low f mult), f = 1 .. COUNT-1, from stage STAGE's blocks of the program-order table;  x[BASE] (no
// factor) is range-reduced instead
template <unsigned LOG_E, unsigned NLDS, unsigned STAGE, unsigned BASE, unsigned COUNT, unsigned E>
PLONK_DEV void wavel_twiddle(FrL (&x)[E], unsigned low, const int32_t* roots, const int32_t* jm) {
    constexpr unsigned LOG_N = LOG_E + 6 + 2 * NLDS, NB = wavel_tw_nb(LOG_E, NLDS, STAGE);
    static_assert(COUNT - 1 == wavel_tw_count(LOG_E, NLDS, STAGE), "twiddle layout");
    const int32_t* blocks = roots + (size_t)wavel_tw_offset(LOG_E, NLDS, STAGE) * NTT_SHOUP_STRIDE;
    x[BASE] = fpl_reduce_small(x[BASE], jm);
    wave_for<COUNT - 1>([&](auto F) {
        constexpr unsigned f = decltype(F)::value + 1;
        x[BASE + f] = fpl_mul_shoup<FrParams, WAVEL_TIGHT_LOG_N(LOG_N)>(x[BASE + f], wavel_ld_root_planar(blocks + (f - 1) * NB * NTT_SHOUP_STRIDE, NB, low));
        if constexpr (WAVEL_TIGHT_LOG_N(LOG_N)) PLONK_SCHED_FENCE();  // 128 VGPRs: keeps the scheduler from holding several twiddles in flight
    });
}
// inputs N-form, |value| < 2.8.  Outputs: x0 in [0, 2^31) (for fpl_reduce_small), x1..x3 multiplicands; |value| < 11.2
PLONK_DEV void dft4l(FrL& x0, FrL& x1, FrL& x2, FrL& x3, const FrLS& w2) {
    const FrL a0 = fpl_add(x0, x2), a1 = fpl_add(x1, x3);                 // [0, 2^30)
    const FrL d0 = fpl_sub(x0, x2);                                       // (-2^29, 2^29)
    const FrL d1 = fpl_mul_shoup(fpl_sub(x1, x3), w2);                    // N-form, (-1.8 m, 2.8 m)
    x0 = fpl_add(a0, a1);                                                 // [0, 2^31)
    x2 = fpl_sub(a0, a1);                                                 // (-2^30, 2^30)
    x1 = fpl_add(d0, d1);                                                 // (-2^29, 2^30)
    x3 = fpl_sub(d0, d1);                                                 // (-2^30, 2^29)
}
// inputs N-form, |value| < 2.8.  Outputs: every limb within (-2^30, 2^30] (multiplicands, and fit for fpl_reduce_small);
// |value| <= 22.4
PLONK_DEV void dft8l(FrL (&x)[8], const FrLS& w1, const FrLS& w2, const FrLS& w3) {
    const FrL a0 = fpl_add(x[0], x[4]), a1 = fpl_add(x[1], x[5]), a2 = fpl_add(x[2], x[6]), a3 = fpl_add(x[3], x[7]);  // [0, 2^30)
    const FrL b0 = fpl_norm(fpl_sub(x[0], x[4]));                         // N-form (sweep 1)
    const FrL b1 = fpl_mul_shoup(fpl_sub(x[1], x[5]), w1), b2 = fpl_mul_shoup(fpl_sub(x[2], x[6]), w2), b3 = fpl_mul_shoup(fpl_sub(x[3], x[7]), w3);  // operands (-2^29, 2^29)
    const FrL c0 = fpl_norm(fpl_add(a0, a2)), c1 = fpl_norm(fpl_add(a1, a3));  // sums [0, 2^31) -> N-form (sweeps 2, 3)
    const FrL d0 = fpl_norm(fpl_sub(a0, a2));                             // (-2^30, 2^30) -> N-form (sweep 4)
    const FrL d1 = fpl_mul_shoup(fpl_sub(a1, a3), w2);                    // operand (-2^30, 2^30)
    const FrL e0 = fpl_add(b0, b2), e1 = fpl_add(b1, b3);                 // [0, 2^30)
    const FrL f0 = fpl_sub(b0, b2);                                       // (-2^29, 2^29)
    const FrL f1 = fpl_mul_shoup(fpl_sub(b1, b3), w2);                    // operand (-2^29, 2^29)
    x[0] = fpl_add(c0, c1);                                               // [0, 2^30)
    x[4] = fpl_sub(c0, c1);                                               // (-2^29, 2^29)
    x[2] = fpl_add(d0, d1);                                               // [0, 2^30)
    x[6] = fpl_sub(d0, d1);                                               // (-2^29, 2^29)
    x[1] = fpl_norm(fpl_add(e0, e1));                                     // [0, 2^31) -> N-form (sweep 5)
    x[5] = fpl_sub(e0, e1);                                               // (-2^30, 2^30)
    x[3] = fpl_add(f0, f1);                                               // (-2^29, 2^30)
    x[7] = fpl_sub(f0, f1);                                               // (-2^30, 2^29)
}
// the digit DFT on the register index: radix 8 (E = 8) or radix 4 (E = 4)
template <unsigned E> PLONK_DEV void wavel_dft(FrL (&x)[E], const FrLS& w1, const FrLS& w2, const FrLS& w3) {
    if constexpr (E == 8) dft8l(x, w1, w2, w3);
    else dft4l(x[0], x[1], x[2], x[3], w2);
}
PLONK_DEV void wavel_lds_st(u32x4* lo, u32x4* hi, uint32_t* top, unsigned i, const FrL& a) {
    lo[i] = u32x4{(uint32_t)a.l[0], (uint32_t)a.l[1], (uint32_t)a.l[2], (uint32_t)a.l[3]};
    hi[i] = u32x4{(uint32_t)a.l[4], (uint32_t)a.l[5], (uint32_t)a.l[6], (uint32_t)a.l[7]};
    top[i] = (uint32_t)a.l[8];
}
PLONK_DEV FrL wavel_lds_ld(const u32x4* lo, const u32x4* hi, const uint32_t* top, unsigned i) {
    const u32x4 a = lo[i], b = hi[i];
    FrL r;
    r.l[0] = (int32_t)a.x; r.l[1] = (int32_t)a.y; r.l[2] = (int32_t)a.z; r.l[3] = (int32_t)a.w;
    r.l[4] = (int32_t)b.x; r.l[5] = (int32_t)b.y; r.l[6] = (int32_t)b.z; r.l[7] = (int32_t)b.w;
    r.l[8] = (int32_t)top[i];
    return r;
}

// threadIdx.x again, as a value the compiler cannot connect to earlier reads: in the 1024-thread kernel (128 VGPRs) the
// per-thread LDS addresses, lane masks and twiddle indices of later stages were otherwise computed at the top of the
// kernel and carried — spilled — through the first stages
template <bool OPAQUE> PLONK_DEV unsigned wavel_tid() {
    unsigned t = threadIdx.x;
#if defined(__HIP_DEVICE_COMPILE__)
    if (OPAQUE) asm volatile("" : "+v"(t));
#endif
    return t;
}
// the same, pinned behind a value the previous stage produces last (the compiler moved the plain form up to the last barrier)
template <bool OPAQUE> PLONK_DEV unsigned wavel_tid_after(int32_t dep) {
    unsigned t = threadIdx.x;
#if defined(__HIP_DEVICE_COMPILE__)
    if (OPAQUE) asm volatile("" : "+v"(t) : "v"(dep));
#else
    (void)dep;
#endif
    return t;
}

// element g of a uniform base as a 32-bit byte offset (g < 2^27: the wave kernels' transforms have at most 2^26 points):
// SGPR-base addressing, one VGPR per address instead of two and no 64-bit address arithmetic
PLONK_DEV const Fr* wavel_at(const Fr* base, unsigned g) { return reinterpret_cast<const Fr*>(reinterpret_cast<const char*>(base) + (g << 5)); }
PLONK_DEV Fr* wavel_at(Fr* base, unsigned g) { return reinterpret_cast<Fr*>(reinterpret_cast<char*>(base) + (g << 5)); }

// waves per SIMD the register allocation aims at: 1024-thread workgroups must fit 128 VGPRs (4); the E = 8 forms run
// faster without spills at 3 (measured in round 2: 18.1 vs 16.8 G elements/s at 2^11 x 2048); E = 4 fits 4 without spills
template <unsigned LOG_E, unsigned NLDS> struct WavelCfg {
    static constexpr unsigned E = 1u << LOG_E, LOG_N = LOG_E + 6 + 2 * NLDS, NT = 64u << (2 * NLDS);
    // TIGHT: the kernel is compiled for 128 VGPRs with the register-saving measures of the 1024-thread kernel (opaque
    // threadIdx re-reads per stage, scheduling fences around the twiddle multiplications)
#ifdef PLONK_NTT_W11_3  // A/B: the 256-thread E = 8 kernel at 3 waves per SIMD (158 VGPRs), as in round 2
    static constexpr bool TIGHT = NLDS == 2;
#else
    static constexpr bool TIGHT = NLDS == 2 || LOG_E == 3;
#endif
    static constexpr unsigned WAVES = (NLDS == 2 || LOG_E == 2) ? 4 : (TIGHT ? 4 : 3);
};

// One transform (or one column / row of a two-pass transform) by one workgroup: the body of both kernels below.
template <unsigned LOG_E, unsigned NLDS>
PLONK_DEV void wavel_transform(const NttWave& p, unsigned char* smem) {
    constexpr unsigned E = 1u << LOG_E, LOG_N = LOG_E + 6 + 2 * NLDS, NT = 64u << (2 * NLDS), LOG_T = 6 + 2 * NLDS;
    u32x4* l_lo = reinterpret_cast<u32x4*>(smem);  // 4 * NT elements as two 16-byte planes and one 4-byte plane
    u32x4* l_hi = l_lo + 4 * NT;
    uint32_t* l_top = reinterpret_cast<uint32_t*>(l_hi + 4 * NT);
    const unsigned tid0 = threadIdx.x;
    const unsigned bidx = p.mode ? blockIdx.y : blockIdx.x;
    const Fr* in = p.in + (size_t)bidx * p.in_bstride;
    Fr* out = p.out + (size_t)bidx * p.out_bstride;
    // column / row of a two-pass transform.  Workgroup b runs on XCD b % 8 (each XCD has its own L2): the remap gives
    // every XCD four ADJACENT columns (rows) per group of 32, so the 32-byte elements it touches share 128-byte lines.
    const unsigned b = blockIdx.x;
    const unsigned sub = !p.mode ? 0 : ((gridDim.x & 31u) ? b : ((b & ~31u) | ((b & 7u) << 2) | ((b >> 3) & 3u)));
    // global index of sub-transform position pos on the input side, of frequency o on the output side
    const unsigned in_shift = p.mode == 1 ? p.log_other : 0, out_shift = p.mode ? p.log_other : 0;
    const unsigned in_off = p.mode == 1 ? sub : (p.mode == 2 ? (p.chunk_log ? sub << p.chunk_log : sub << LOG_N) : 0);
    const unsigned out_off = p.mode ? sub : 0;
    const unsigned chunk_mask = (1u << p.chunk_log) - 1;
    const int32_t* jm = p.jm;
    const FrLS &w8_1 = p.w8[0], &w8_2 = p.w8[1], &w8_3 = p.w8[2];  // kernel arguments: scalar registers

    FrL x[E];
    wave_for<E>([&](auto J) {  // position j * NT + tid: consecutive lanes read consecutive positions
        constexpr unsigned j = decltype(J)::value;
        const unsigned pos = j * NT + tid0;
        const unsigned g = p.chunk_log ? (pos >> p.chunk_log) * p.chunk_stride + (pos & chunk_mask) + in_off : (pos << in_shift) + in_off;
        x[j] = g < p.in_len ? fpl_from_fp(fp_load(wavel_at(in, g))) : fpl_zero<FrParams>();  // [0, 2m): canonical input, or the column pass's redundant residues
    });
    if (p.in_scale) {
        wave_for<E>([&](auto J) {
            constexpr unsigned j = decltype(J)::value;
            const unsigned g = ((j * NT + tid0) << in_shift) + in_off;
            if (g < p.in_len) x[j] = fpl_mul(x[j], fpl_from_fp(fp_load(wavel_at(p.in_scale, g))));
        });
    }
    // stage A: digit = the top LOG_E index bits, low = tid0
    wavel_dft<E>(x, w8_1, w8_2, w8_3);
    wavel_twiddle<LOG_E, NLDS, 0, 0, E>(x, tid0, p.roots, jm);
    // L radix-4 stages on the wave bits: swap register bits (1, 0) with thread bits (tb + 1, tb)
    wave_for<NLDS>([&](auto S) {
        constexpr unsigned s = decltype(S)::value;
        constexpr unsigned tb = 6 + 2 * (NLDS - 1 - s);
        const unsigned tid = wavel_tid<WavelCfg<LOG_E, NLDS>::TIGHT>();
        const unsigned mine = (tid >> tb) & 3u, rest = tid & ~(3u << tb);
        wave_for<E / 4>([&](auto R2) {
            constexpr unsigned r2 = decltype(R2)::value;
            wave_for<4>([&](auto Q) { wavel_lds_st(l_lo, l_hi, l_top, decltype(Q)::value * NT + tid, x[4 * r2 + decltype(Q)::value]); });
            __syncthreads();
            wave_for<4>([&](auto Q) { x[4 * r2 + decltype(Q)::value] = wavel_lds_ld(l_lo, l_hi, l_top, mine * NT + (rest | (decltype(Q)::value << tb))); });
            __syncthreads();
        });
        const unsigned low = tid & ((1u << tb) - 1);
        wave_for<E / 4>([&](auto R2) {
            constexpr unsigned r2 = decltype(R2)::value;
            dft4l(x[4 * r2], x[4 * r2 + 1], x[4 * r2 + 2], x[4 * r2 + 3], w8_2);
            wavel_twiddle<LOG_E, NLDS, 1 + s, 4 * r2, 4>(x, low, p.roots, jm);
        });
    });
    const unsigned lane = wavel_tid_after<WavelCfg<LOG_E, NLDS>::TIGHT>(x[E - 1].l[8]) & 63u;
    if constexpr (E == 8) {
        // stage on lane bits 5..3
        wavel_swap_bit<E, 2, 32>(x, lane);
        wavel_swap_bit<E, 1, 16>(x, lane);
        wavel_swap_bit<E, 0, 8>(x, lane);
        dft8l(x, w8_1, w8_2, w8_3);
        wavel_twiddle<LOG_E, NLDS, NLDS + 1, 0, 8>(x, lane & 7u, p.roots, jm);
        // stage on lane bits 2..0
        wavel_swap_bit<E, 2, 4>(x, lane);
        wavel_swap_bit<E, 1, 2>(x, lane);
        wavel_swap_bit<E, 0, 1>(x, lane);
        dft8l(x, w8_1, w8_2, w8_3);
    } else {
        // stages on lane bits (5, 4), (3, 2), (1, 0)
        wavel_swap_bit<E, 1, 32>(x, lane);
        wavel_swap_bit<E, 0, 16>(x, lane);
        dft4l(x[0], x[1], x[2], x[3], w8_2);
        wavel_twiddle<LOG_E, NLDS, NLDS + 1, 0, 4>(x, lane & 15u, p.roots, jm);
        wavel_swap_bit<E, 1, 8>(x, lane);
        wavel_swap_bit<E, 0, 4>(x, lane);
        dft4l(x[0], x[1], x[2], x[3], w8_2);
        wavel_twiddle<LOG_E, NLDS, NLDS + 2, 0, 4>(x, lane & 3u, p.roots, jm);
        wavel_swap_bit<E, 1, 2>(x, lane);
        wavel_swap_bit<E, 0, 1>(x, lane);
        dft4l(x[0], x[1], x[2], x[3], w8_2);
        x[0] = fpl_norm(x[0]);  // [0, 2^31) -> N-form: the optional multiplications below take limbs within (-2^30, 2^30]
    }
    // frequency of register j: digits in processing order, first digit least significant
    unsigned k, shift;
    const unsigned tid = wavel_tid<WavelCfg<LOG_E, NLDS>::TIGHT>();
    if constexpr (E == 8) {
        //   d_A = (lane bit 5) * 4 + thread bits (top pair);  then the remaining wave pairs;  (lane bits 4, 3);  (lane bits 2..0);  j
        //   (without wave stages the first digit is simply lane bits 5..3)
        shift = 3;
        if (NLDS) {
            k = (((lane >> 5) & 1u) << 2) | ((tid >> (6 + 2 * (NLDS > 0 ? NLDS - 1 : 0))) & 3u);
            for (unsigned s = 1; s < NLDS; s++) {
                k |= ((tid >> (6 + 2 * (NLDS - 1 - s))) & 3u) << shift;
                shift += 2;
            }
            k |= ((lane >> 3) & 3u) << shift;
            shift += 2;
        } else {
            k = (lane >> 3) & 7u;
        }
        k |= (lane & 7u) << shift;
        shift += 3;
    } else {
        //   every digit has two bits: the thread-index pairs from the top down hold d_A, d_B, ..; j is the last digit
        k = 0;
        wave_for<LOG_T / 2>([&](auto I) {
            constexpr unsigned i = decltype(I)::value;
            k |= ((tid >> (LOG_T - 2 - 2 * i)) & 3u) << (2 * i);
        });
        shift = LOG_T;
    }
    if (p.mode == 1) {  // inter-pass twiddle w_N^(column * frequency)
        wave_for<E>([&](auto J) {
            constexpr unsigned j = decltype(J)::value;
            const unsigned e = (sub + p.sub_base) * (k | (j << shift));  // < N
            if (e || p.tw_always) {  // two multiplications by table constants (380 instructions) instead of forming their product first (434)
                x[j] = fpl_mul_shoup(x[j], wavel_ld_root(p.tw_lo, e & ((1u << NTT_TW_LO_LOG) - 1)));
                if (p.log_n > NTT_TW_LO_LOG) x[j] = fpl_mul_shoup(x[j], wavel_ld_root(p.tw_hi, e >> NTT_TW_LO_LOG));
            }
        });
    }
    if (p.out_scale) {
        wave_for<E>([&](auto J) {
            constexpr unsigned j = decltype(J)::value;
            x[j] = fpl_mul(x[j], fpl_from_fp(fp_load(wavel_at(p.out_scale, ((k | (j << shift)) << out_shift) + out_off))));
        });
    }
    if (p.has_out_scalar) {
        const FrL sc = fpl_from_fp_uniform(p.out_scalar);
        wave_for<E>([&](auto J) { x[decltype(J)::value] = fpl_mul(x[decltype(J)::value], sc); });
    }
    wave_for<E>([&](auto J) {  // |value| <= 22.4 m whatever happened above -> (0.49 m, 1.51 m) -> canonical (the column pass skips that last step)
        constexpr unsigned j = decltype(J)::value;
        fp_store(wavel_at(out, ((k | (j << shift)) << out_shift) + out_off), fpl_pack_positive(fpl_reduce_small<FrParams, 1>(x[j], jm), p.mode != 1));
    });
}

template <unsigned LOG_E, unsigned NLDS>
__global__ void __launch_bounds__(64u << (2 * NLDS), (WavelCfg<LOG_E, NLDS>::WAVES)) ntt_wavel_kernel(NttWave p) {
    PLONK_DYN_SMEM(smem);
    wavel_transform<LOG_E, NLDS>(p, smem);
}


// ------------------------------------------------------------------------------------------------
// host side: roots of unity, cached tables, pass planning

static Fr host_fr_from_u64(uint64_t x) {
    Fr a = fp_zero<FrParams>();
    a.v[0] = (uint32_t)x;
    a.v[1] = (uint32_t)(x >> 32);
    return fp_to_mont(a);
}

Fr host_root_of_unity(unsigned log_n, bool inverse) {
    Fr w;
    for (int i = 0; i < 8; i++) w.v[i] = inverse ? FrRoots::w28_inv(i) : FrRoots::w28(i);
    for (unsigned i = log_n; i < PLONK_FR_TWO_ADICITY; i++) w = fp_sqr(w);
    return w;
}

static int alloc_table(plonk_ctx* ctx, size_t n, Fr** out) {
    void* p = nullptr;
    if (hipMalloc(&p, (n ? n : 1) * sizeof(Fr)) != hipSuccess) {
        plonk_set_error("hipMalloc of %zu-entry twiddle table failed", n);
        return PLONK_ERR_NOMEM;
    }
    ctx->owned.push_back(p);
    *out = (Fr*)p;
    return PLONK_OK;
}

static int get_small_tw(plonk_ctx* ctx, unsigned log_r, bool inverse, const Fr** out) {
    unsigned key = log_r | (inverse ? 256u : 0u);
    auto it = ctx->tw.small.find(key);
    if (it == ctx->tw.small.end()) {
        Fr* t;
        size_t n = log_r ? ((size_t)1 << (log_r - 1)) : 1;
        PLONK_TRY(alloc_table(ctx, n, &t));
        PLONK_TRY(k_fr_powers(ctx, host_root_of_unity(log_r, inverse), fp_one<FrParams>(), t, n));
        it = ctx->tw.small.emplace(key, t).first;
    }
    *out = it->second;
    return PLONK_OK;
}

static int get_lo_hi(plonk_ctx* ctx, unsigned log_n, bool inverse, const Fr** lo, const Fr** hi) {
    unsigned key = log_n | (inverse ? 256u : 0u);
    auto it = ctx->tw.lo.find(key);
    if (it == ctx->tw.lo.end()) {
        Fr w = host_root_of_unity(log_n, inverse);
        unsigned log_lo = log_n < NTT_TW_LO_LOG ? log_n : NTT_TW_LO_LOG;
        Fr *tl, *th;
        PLONK_TRY(alloc_table(ctx, (size_t)1 << log_lo, &tl));
        PLONK_TRY(k_fr_powers(ctx, w, fp_one<FrParams>(), tl, (size_t)1 << log_lo));
        size_t nhi = log_n > NTT_TW_LO_LOG ? ((size_t)1 << (log_n - NTT_TW_LO_LOG)) : 1;
        Fr whi = w;
        for (unsigned i = 0; i < NTT_TW_LO_LOG; i++) whi = fp_sqr(whi);
        PLONK_TRY(alloc_table(ctx, nhi, &th));
        PLONK_TRY(k_fr_powers(ctx, whi, fp_one<FrParams>(), th, nhi));
        ctx->tw.lo.emplace(key, tl);
        ctx->tw.hi.emplace(key, th);
        it = ctx->tw.lo.find(key);
    }
    *lo = it->second;
    *hi = ctx->tw.hi[key];
    return PLONK_OK;
}

// full table w^0 .. w^(N-1) (barycentric evaluation, permutation argument)
int ntt_get_roots(plonk_ctx* ctx, unsigned log_n, bool inverse, const Fr** out) {
    unsigned key = log_n | (inverse ? 256u : 0u);
    auto it = ctx->tw.full.find(key);
    if (it == ctx->tw.full.end()) {
        Fr* t;
        PLONK_TRY(alloc_table(ctx, (size_t)1 << log_n, &t));
        PLONK_TRY(k_fr_powers(ctx, host_root_of_unity(log_n, inverse), fp_one<FrParams>(), t, (size_t)1 << log_n));
        it = ctx->tw.full.emplace(key, t).first;
    }
    *out = it->second;
    return PLONK_OK;
}

// pass radices, most significant digit first
static unsigned plan_passes(const plonk_ctx* ctx, unsigned log_n, unsigned radices[4]) {
    if (log_n <= ctx->ntt_single_log) {
        radices[0] = log_n;
        return 1;
    }
    unsigned P = (log_n + ctx->ntt_radix_log - 1) / ctx->ntt_radix_log;
    if (P < 2) P = 2;
    unsigned base = log_n / P, extra = log_n % P;
    for (unsigned i = 0; i < P; i++) radices[i] = base + (i < extra ? 1 : 0);
    return P;
}

// the wave kernels (variant C): N = 2^8 .. 2^13 in one pass (one workgroup of N / 4 or N / 8 threads per transform), and
// N = R1 R2 with R1, R2 from that set in two passes (columns, then rows).  Default splits: the fastest measured on MI355X
// for a lone transform (profiles/r03_b_ntt_splits.jsonl: the 4-element-per-thread kernels where a size allows them —
// twice the waves —, and short column transforms for the largest sizes); plonk_ntt_set_split overrides one size.
bool ntt_wave_plan(const plonk_ctx* ctx, unsigned log_n, unsigned* log_r1, unsigned* log_r2) {
    if (log_n >= 8 && log_n <= 13) {
        *log_r1 = log_n;
        *log_r2 = 0;
        return true;
    }
    if (log_n < 16 || log_n > 26) return false;
    //                                   2^16 17  18  19  20  21  22  23  24  25  26
    static const unsigned char best[] = {8,   9, 10, 10, 10, 11, 13, 13, 11, 12, 13};
    unsigned r1 = best[log_n - 16];
    if (ctx && log_n < sizeof ctx->ntt_split / sizeof ctx->ntt_split[0] && ctx->ntt_split[log_n]) r1 = ctx->ntt_split[log_n];
    if (r1 < 8 || r1 > 13 || log_n - r1 < 8 || log_n - r1 > 13) return false;
    *log_r1 = r1;
    *log_r2 = log_n - r1;
    return true;
}

// limb form of a packed table: NTT_LIMB_STRIDE words per entry (what wavel_ld_tw reads); with shoup != 0 the Shoup pair of
// every entry, NTT_SHOUP_STRIDE words (what wavel_ld_root reads)
struct Ninv261 { uint32_t l[9]; };
__global__ void ntt_limb_table_kernel(const Fr* in, int32_t* out, size_t n, int shoup, Ninv261 ninv) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fr v = fp_load(in + i);
    if (shoup) {
        const FrLS a = fpl_shoup_from_mont(v, ninv.l);
        int32_t* o = out + i * NTT_SHOUP_STRIDE;
        for (int w = 0; w < 9; w++) {
            o[w] = a.w[w];
            o[9 + w] = a.wp[w];
        }
        o[18] = o[19] = 0;
    } else {
        const FrL a = fpl_from_fp(v);
        int32_t* o = out + i * NTT_LIMB_STRIDE;
        for (int w = 0; w < 9; w++) o[w] = a.l[w];
        o[9] = o[10] = o[11] = 0;
    }
}

static int ntt_limb_table(plonk_ctx* ctx, std::map<unsigned, int32_t*>& cache, unsigned key, const Fr* packed, size_t n, bool shoup, const int32_t** out) {
    auto it = cache.find(key);
    if (it == cache.end()) {
        void* d = nullptr;
        Ninv261 ninv;
        fpl_ninv261<FrParams>(ninv.l);
        if (hipMalloc(&d, n * (shoup ? NTT_SHOUP_STRIDE : NTT_LIMB_STRIDE) * sizeof(int32_t)) != hipSuccess) {
            plonk_set_error("hipMalloc of a %zu-entry limb-form twiddle table failed", n);
            return PLONK_ERR_NOMEM;
        }
        ctx->owned.push_back(d);
        PLONK_LAUNCH(ntt_limb_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, packed, (int32_t*)d, n, shoup ? 1 : 0, ninv);
        PLONK_CHECK_HIP(hipGetLastError());
        it = cache.emplace(key, (int32_t*)d).first;
    }
    *out = it->second;
    return PLONK_OK;
}

// one block of a program-order twiddle table (wavel_tw_*): entry low = the Shoup pair of roots[(low f mult) mod N], five planes
__global__ void ntt_program_block_kernel(const Fr* roots, unsigned log_n, unsigned nb, unsigned f, unsigned mult, int32_t* block, Ninv261 ninv) {
    const unsigned low = blockIdx.x * blockDim.x + threadIdx.x;
    if (low >= nb) return;
    const FrLS a = fpl_shoup_from_mont(fp_load(roots + ((low * f * mult) & ((1u << log_n) - 1))), ninv.l);
    int32_t e[NTT_SHOUP_STRIDE];
    for (int w = 0; w < 9; w++) {
        e[w] = a.w[w];
        e[9 + w] = a.wp[w];
    }
    e[18] = e[19] = 0;
    for (unsigned pl = 0; pl < 5; pl++)
        for (unsigned w = 0; w < NTT_PLANE_WORDS; w++) block[((size_t)pl * nb + low) * NTT_PLANE_WORDS + w] = e[pl * NTT_PLANE_WORDS + w];
}

// the twiddles of the wave kernel serving 2^log_n, in program order (built once per size and direction)
static int ntt_get_roots_limbs(plonk_ctx* ctx, unsigned log_n, bool inverse, const int32_t** out) {
    const unsigned key = log_n | (inverse ? 256u : 0u);
    auto it = ctx->tw.full_l.find(key);
    if (it == ctx->tw.full_l.end()) {
        const Fr* packed;
        PLONK_TRY(ntt_get_roots(ctx, log_n, inverse, &packed));
        const unsigned log_e = (log_n & 1) ? 3 : 2, nlds = (log_n - 6 - log_e) / 2, stages = wavel_tw_stages(log_e, nlds);
        const size_t entries = wavel_tw_offset(log_e, nlds, stages);
        void* d = nullptr;
        if (hipMalloc(&d, entries * NTT_SHOUP_STRIDE * sizeof(int32_t)) != hipSuccess) {
            plonk_set_error("hipMalloc of a %zu-entry twiddle table failed", entries);
            return PLONK_ERR_NOMEM;
        }
        ctx->owned.push_back(d);
        Ninv261 ninv;
        fpl_ninv261<FrParams>(ninv.l);
        for (unsigned st = 0; st < stages; st++) {
            const unsigned nb = wavel_tw_nb(log_e, nlds, st), count = wavel_tw_count(log_e, nlds, st), mult = wavel_tw_mult(log_e, nlds, st);
            for (unsigned f = 1; f <= count; f++) {
                int32_t* block = (int32_t*)d + ((size_t)wavel_tw_offset(log_e, nlds, st) + (size_t)(f - 1) * nb) * NTT_SHOUP_STRIDE;
                PLONK_LAUNCH(ntt_program_block_kernel, dim3((nb + 63) / 64), dim3(64), 0, ctx->stream, packed, log_n, nb, f, mult, block, ninv);
            }
        }
        PLONK_CHECK_HIP(hipGetLastError());
        it = ctx->tw.full_l.emplace(key, (int32_t*)d).first;
    }
    *out = it->second;
    return PLONK_OK;
}

// inter-pass twiddle tables as Shoup pairs; scaled: the hi table times 1/N (the inverse transform's factor, folded in)
static int get_lo_hi_limbs(plonk_ctx* ctx, unsigned log_n, bool inverse, bool scaled, const int32_t** lo, const int32_t** hi) {
    const Fr *plo, *phi;
    PLONK_TRY(get_lo_hi(ctx, log_n, inverse, &plo, &phi));
    const unsigned key = log_n | (inverse ? 256u : 0u);
    const unsigned log_lo = log_n < NTT_TW_LO_LOG ? log_n : NTT_TW_LO_LOG;
    const size_t nhi = log_n > NTT_TW_LO_LOG ? ((size_t)1 << (log_n - NTT_TW_LO_LOG)) : 1;
    PLONK_TRY(ntt_limb_table(ctx, ctx->tw.lo_l, key, plo, (size_t)1 << log_lo, true, lo));
    if (!scaled) return ntt_limb_table(ctx, ctx->tw.hi_l, key, phi, nhi, true, hi);
    if (ctx->tw.hi_l.find(key | 512u) == ctx->tw.hi_l.end()) {  // (1/N) * w_hi^k, built once
        Fr whi = host_root_of_unity(log_n, inverse);
        for (unsigned i = 0; i < NTT_TW_LO_LOG; i++) whi = fp_sqr(whi);
        void* tmp = nullptr;  // (not a scratch slot: callers hold those across this call)
        if (hipMalloc(&tmp, nhi * sizeof(Fr)) != hipSuccess) {
            plonk_set_error("hipMalloc of a %zu-entry twiddle table failed", nhi);
            return PLONK_ERR_NOMEM;
        }
        int rc = k_fr_powers(ctx, whi, fp_inv(host_fr_from_u64((uint64_t)1 << log_n)), (Fr*)tmp, nhi);
        if (rc == PLONK_OK) rc = ntt_limb_table(ctx, ctx->tw.hi_l, key | 512u, (const Fr*)tmp, nhi, true, hi);
        hipStreamSynchronize(ctx->stream);  // the packed copy must outlive the conversion kernel only
        hipFree(tmp);
        return rc;
    }
    return ntt_limb_table(ctx, ctx->tw.hi_l, key | 512u, nullptr, nhi, true, hi);
}

// fpl_reduce_small's table of j * m for the limb-form kernel: 49 entries of 12 words, built on the host once per context
static int ntt_get_jm(plonk_ctx* ctx, const int32_t** out) {
    if (!ctx->ntt_jm) {
        int32_t host[(2 * FPL_RS_J + 1) * 12];
        for (int j = -FPL_RS_J; j <= FPL_RS_J; j++) fpl_jm_entry<FrParams>(j, host + (j + FPL_RS_J) * 12);
        void* d = nullptr;
        if (hipMalloc(&d, sizeof host) != hipSuccess) {
            plonk_set_error("hipMalloc of the NTT range-reduction table failed");
            return PLONK_ERR_NOMEM;
        }
        ctx->owned.push_back(d);
        PLONK_CHECK_HIP(hipMemcpy(d, host, sizeof host, hipMemcpyHostToDevice));
        ctx->ntt_jm = (const int32_t*)d;
    }
    *out = ctx->ntt_jm;
    return PLONK_OK;
}

// the radix-8 / radix-4 roots w_8^k of a transform direction as Shoup pairs (kernel arguments)
static void ntt_wave_w8(NttWave* p, bool inverse) {
    uint32_t ninv[9];
    fpl_ninv261<FrParams>(ninv);
    const Fr w8 = host_root_of_unity(3, inverse), w4 = fp_sqr(w8);
    p->w8[0] = fpl_shoup_from_mont(w8, ninv);
    p->w8[1] = fpl_shoup_from_mont(w4, ninv);
    p->w8[2] = fpl_shoup_from_mont(fp_mul(w4, w8), ninv);
}

template <unsigned LOG_E, unsigned NLDS> static int ntt_wavel_launch_as(plonk_ctx* ctx, const NttWave& q, unsigned grid_x, unsigned grid_y) {
    constexpr unsigned nt = 64u << (2 * NLDS);
    const size_t shmem = NLDS ? (size_t)4 * nt * 36 : 0;  // one round of the wave-bit exchange: 4 elements of 9 words per thread
    if (NLDS == 2 && !ctx->ntt_wavel_attr_set[LOG_E - 2]) {  // 144 KiB: above the default limit; a per-device attribute, tracked per context
        PLONK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ntt_wavel_kernel<LOG_E, NLDS>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(144 * 1024)));
        ctx->ntt_wavel_attr_set[LOG_E - 2] = true;
    }
    void (*const kern)(NttWave) = ntt_wavel_kernel<LOG_E, NLDS>;  // (a template-id's comma would split the macro's arguments)
    PLONK_LAUNCH(kern, dim3(grid_x, grid_y), dim3(nt), shmem, ctx->stream, q);
    return PLONK_OK;
}

// log_r = 8, 10, 12: 4 elements per thread; 9, 11, 13: 8 elements per thread
static int ntt_wave_launch(plonk_ctx* ctx, const NttWave& p, unsigned log_r, unsigned grid_x, unsigned grid_y) {
    NttWave q = p;
    PLONK_TRY(ntt_get_jm(ctx, &q.jm));
    switch (log_r) {
        case 8: return ntt_wavel_launch_as<2, 0>(ctx, q, grid_x, grid_y);
        case 10: return ntt_wavel_launch_as<2, 1>(ctx, q, grid_x, grid_y);
        case 12: return ntt_wavel_launch_as<2, 2>(ctx, q, grid_x, grid_y);
        case 9: return ntt_wavel_launch_as<3, 0>(ctx, q, grid_x, grid_y);
        case 11: return ntt_wavel_launch_as<3, 1>(ctx, q, grid_x, grid_y);
        case 13: return ntt_wavel_launch_as<3, 2>(ctx, q, grid_x, grid_y);
    }
    plonk_set_error("no wave kernel for a 2^%u-point transform", log_r);
    return PLONK_ERR_ARG;
}

static int ntt_run_wave(plonk_ctx* ctx, const Fr* in, Fr* out, unsigned log_n, bool inverse, size_t batch, size_t in_len,
                        size_t in_bstride, size_t out_bstride, const Fr* in_scale, const Fr* out_scale, bool scale_by_n_inv) {
    const size_t N = (size_t)1 << log_n;
    unsigned log_r1 = 0, log_r2 = 0;
    ntt_wave_plan(ctx, log_n, &log_r1, &log_r2);
    NttWave p;
    memset(&p, 0, sizeof p);
    p.log_n = log_n;
    ntt_wave_w8(&p, inverse);
    Fr n_inv = fp_zero<FrParams>();
    if (scale_by_n_inv) n_inv = fp_inv(host_fr_from_u64((uint64_t)N));
    const unsigned in_len32 = (unsigned)(in_len < N ? in_len : N);
    if (!log_r2) {
        p.in = in;
        p.out = out;
        p.in_bstride = in_bstride;
        p.out_bstride = out_bstride;
        p.in_len = in_len32;
        PLONK_TRY(ntt_get_roots_limbs(ctx, log_n, inverse, &p.roots));
        p.in_scale = in_scale;
        p.out_scale = out_scale;
        p.out_scalar = n_inv;
        p.has_out_scalar = scale_by_n_inv;
        // an in-place transform is safe: every thread has read all of its inputs before any thread stores (the stages
        // in between are separated by barriers for L > 0; for L = 0 the single wave runs in lock step)
        PLONK_TRY(prof_begin(ctx, "ntt_pass", 64.0 * (double)N * (double)batch));
        for (size_t b0 = 0; b0 < batch; b0 += (size_t)1 << 30) {  // grid.x carries the batch
            const size_t nb = batch - b0 < ((size_t)1 << 30) ? batch - b0 : (size_t)1 << 30;
            NttWave q = p;
            q.in = in + b0 * in_bstride;
            q.out = out + b0 * out_bstride;
            PLONK_TRY(ntt_wave_launch(ctx, q, log_n, (unsigned)nb, 1));
        }
        PLONK_TRY(prof_end(ctx));
        PLONK_CHECK_HIP(hipGetLastError());
        return PLONK_OK;
    }
    // two passes through a scratch copy: columns (R1 points each, stride R2), then rows (R2 points each)
    PLONK_REQUIRE(batch <= 65535, PLONK_ERR_ARG, "NTT batch %zu exceeds 65535", batch);
    void* sc;
    PLONK_TRY(ctx_scratch(ctx, 0, batch * N * sizeof(Fr), &sc));
    Fr* tmp = (Fr*)sc;
    PLONK_TRY(get_lo_hi_limbs(ctx, log_n, inverse, scale_by_n_inv, &p.tw_lo, &p.tw_hi));
    p.tw_always = scale_by_n_inv ? 1u : 0u;
    NttWave a = p;
    a.mode = 1;
    a.log_other = log_r2;
    a.in = in;
    a.out = tmp;
    a.in_bstride = in_bstride;
    a.out_bstride = N;
    a.in_len = in_len32;
    a.in_scale = in_scale;
    PLONK_TRY(ntt_get_roots_limbs(ctx, log_r1, inverse, &a.roots));
    NttWave c = p;
    c.mode = 2;
    c.log_other = log_r1;
    c.in = tmp;
    c.out = out;
    c.in_bstride = N;
    c.out_bstride = out_bstride;
    c.in_len = (unsigned)N;
    c.out_scale = out_scale;
    c.has_out_scalar = 0;  // 1/N went into the column pass's inter-pass twiddles (tw_hi)
    c.tw_always = 0;
    PLONK_TRY(ntt_get_roots_limbs(ctx, log_r2, inverse, &c.roots));
    PLONK_TRY(prof_begin(ctx, "ntt_pass", 32.0 * (double)N * (double)batch));
    PLONK_TRY(ntt_wave_launch(ctx, a, log_r1, 1u << log_r2, (unsigned)batch));
    PLONK_TRY(prof_end(ctx));
    PLONK_TRY(prof_begin(ctx, "ntt_pass", 32.0 * (double)N * (double)batch));
    PLONK_TRY(ntt_wave_launch(ctx, c, log_r2, 1u << log_r1, (unsigned)batch));
    PLONK_TRY(prof_end(ctx));
    PLONK_CHECK_HIP(hipGetLastError());
    return PLONK_OK;
}

// ---- distributed four-step transform: the two local steps (the all-to-all between them is comm.hip's) ------------------
// N = R1 R2 over W = 2^log_w ranks.  Rank g holds the columns c = g R2/W .. of the R1 x R2 matrix x[i1 R2 + c] as
// [R1][R2/W]; ntt_dist_columns leaves (k1, c) * w_N^(c k1) in the same layout, whose block of rows k1 = h R1/W .. is what
// rank h needs; ntt_dist_rows takes the W received blocks [source rank][R1/W][R2/W] and leaves the frequencies
// k1 + R1 k2 of its rows as [R2][R1/W] (times 1/N for the inverse).
int ntt_dist_plan(unsigned log_n, unsigned log_w, unsigned* log_r1, unsigned* log_r2) {
    unsigned r1 = 0, r2 = 0;
    // the default split (no per-context override: every rank must pick the same one)
    PLONK_REQUIRE(ntt_wave_plan(nullptr, log_n, &r1, &r2) && r2, PLONK_ERR_ARG,
                  "distributed NTT supports sizes 2^16 .. 2^26 (got 2^%u)", log_n);
    PLONK_REQUIRE(log_w + 5 <= r2 && log_w + 5 <= r1, PLONK_ERR_ARG, "2^%u ranks are too many for a 2^%u-point transform", log_w, log_n);
    *log_r1 = r1;
    *log_r2 = r2;
    return PLONK_OK;
}

static void ntt_wave_consts(NttWave* p, unsigned log_n, bool inverse) {
    memset(p, 0, sizeof *p);
    p->log_n = log_n;
    ntt_wave_w8(p, inverse);
}

int ntt_dist_columns(plonk_ctx* ctx, const Fr* in, Fr* out, unsigned log_n, unsigned log_w, unsigned rank, bool inverse) {
    unsigned log_r1, log_r2;
    PLONK_TRY(ntt_dist_plan(log_n, log_w, &log_r1, &log_r2));
    const unsigned log_cl = log_r2 - log_w;
    NttWave a;
    ntt_wave_consts(&a, log_n, inverse);
    PLONK_TRY(get_lo_hi_limbs(ctx, log_n, inverse, false, &a.tw_lo, &a.tw_hi));
    PLONK_TRY(ntt_get_roots_limbs(ctx, log_r1, inverse, &a.roots));
    a.mode = 1;
    a.log_other = log_cl;
    a.sub_base = rank << log_cl;
    a.in = in;
    a.out = out;
    a.in_len = 1u << (log_n - log_w);
    PLONK_TRY(prof_begin(ctx, "ntt_pass", 32.0 * (double)((size_t)1 << (log_n - log_w))));
    PLONK_TRY(ntt_wave_launch(ctx, a, log_r1, 1u << log_cl, 1));
    PLONK_TRY(prof_end(ctx));
    PLONK_CHECK_HIP(hipGetLastError());
    return PLONK_OK;
}

int ntt_dist_rows(plonk_ctx* ctx, const Fr* in, Fr* out, unsigned log_n, unsigned log_w, unsigned rank, bool inverse) {
    (void)rank;
    unsigned log_r1, log_r2;
    PLONK_TRY(ntt_dist_plan(log_n, log_w, &log_r1, &log_r2));
    const unsigned log_cl = log_r2 - log_w, log_kl = log_r1 - log_w;
    NttWave c;
    ntt_wave_consts(&c, log_n, inverse);
    PLONK_TRY(ntt_get_roots_limbs(ctx, log_r2, inverse, &c.roots));
    c.mode = 2;
    c.log_other = log_kl;  // output stride: frequency k2 of local row kl at out[k2 * R1/W + kl]
    if (log_w) {           // W chunks [source rank][R1/W][R2/W]; one rank: plain contiguous rows
        c.chunk_log = log_cl;
        c.chunk_stride = 1u << (log_kl + log_cl);
    }
    c.in = in;
    c.out = out;
    c.in_len = 1u << (log_n - log_w);
    if (inverse) {
        c.out_scalar = fp_inv(host_fr_from_u64((uint64_t)1 << log_n));
        c.has_out_scalar = 1;
    }
    PLONK_TRY(prof_begin(ctx, "ntt_pass", 32.0 * (double)((size_t)1 << (log_n - log_w))));
    PLONK_TRY(ntt_wave_launch(ctx, c, log_r2, 1u << log_kl, 1));
    PLONK_TRY(prof_end(ctx));
    PLONK_CHECK_HIP(hipGetLastError());
    return PLONK_OK;
}

int ntt_run(plonk_ctx* ctx, const Fr* in, Fr* out, unsigned log_n, bool inverse, size_t batch, size_t in_len,
            size_t in_bstride, size_t out_bstride, const Fr* in_scale, const Fr* out_scale, bool scale_by_n_inv) {
    PLONK_REQUIRE(log_n <= PLONK_FR_TWO_ADICITY, PLONK_ERR_ARG, "NTT size 2^%u exceeds the 2-adicity (28) of BN254 Fr", log_n);
    if (!batch) return PLONK_OK;
    const size_t N = (size_t)1 << log_n;
    unsigned wr1, wr2;
    // The wave kernels serve every size they cover (2^8 .. 2^13 in one launch, 2^16 .. 2^26 in two): measured on MI355X
    // they beat the LDS kernels at every such size and batch (profiles/r02_e_ntt_kinds.json, r02_v_ntt_kinds.json,
    // r03_*).  kind 4 = automatic choice among the LDS kernels only (A/B runs); a plonk_ntt_configure with small tiles
    // (the multi-pass tests) also keeps a transform on the LDS kernels.
    if ((ctx->ntt_kind == 0 || ctx->ntt_kind == 5) && ntt_wave_plan(ctx, log_n, &wr1, &wr2) && ((ctx->ntt_single_log >= 11 && ctx->ntt_radix_log >= 10) || ctx->ntt_kind == 5))
        return ntt_run_wave(ctx, in, out, log_n, inverse, batch, in_len, in_bstride, out_bstride, in_scale, out_scale, scale_by_n_inv);
    PLONK_REQUIRE(batch <= 65535, PLONK_ERR_ARG, "NTT batch %zu exceeds 65535", batch);
    unsigned radices[4];
    const unsigned P = plan_passes(ctx, log_n, radices);
    PLONK_REQUIRE(P <= 4, PLONK_ERR_ARG, "NTT of size 2^%u needs %u passes at radix 2^%u (max 4)", log_n, P, ctx->ntt_radix_log);

    Fr* tmp = nullptr;
    if (P > 1) {
        void* s;
        PLONK_TRY(ctx_scratch(ctx, 0, batch * N * sizeof(Fr), &s));
        tmp = (Fr*)s;
    }
    const Fr *lo = nullptr, *hi = nullptr;
    if (P > 1) PLONK_TRY(get_lo_hi(ctx, log_n, inverse, &lo, &hi));

    unsigned log_h = 0;
    for (unsigned pi = 0; pi < P; pi++) {
        NttPass p;
        memset(&p, 0, sizeof p);
        const bool first = pi == 0, last = pi == P - 1;
        p.in = first ? in : tmp;
        p.out = last ? out : tmp;
        p.in_bstride = first ? in_bstride : N;
        p.out_bstride = last ? out_bstride : N;
        p.log_n = log_n;
        p.log_r = radices[pi];
        p.log_h = log_h;
        p.log_s = log_n - log_h - p.log_r;
        p.first = first;
        p.last = last;
        p.in_len = (unsigned)(in_len < N ? in_len : N);
        // three-pass transforms (N >= 2^21) run faster with 2048-element tiles (two workgroups per CU):
        // measured 2.40 ms vs 2.90 ms at 2^24 (profiles/r01_h_sweep.jsonl); 2^12..2^20 prefer 4096
        unsigned tile_log = (ctx->ntt_tile_log == 12 && P >= 3) ? 11 : ctx->ntt_tile_log;
        // small jobs: shrink the tile until the pass has >= 512 workgroups (two per CU) or one column per tile — a lone
        // 2^16 transform used to run on 16 workgroups of 4096 elements.  Narrow tiles give up 128-byte chunks, but such
        // a job's data (<= 32 MiB) sits in L2 / MALL anyway.
        // Measured (profiles/r02_h_ntt_adaptive_tiles.json): 2^16 0.096 -> 0.042 ms, 2^14 0.088 -> 0.040, 2^18 0.113 -> 0.075,
        // 2^19 0.122 -> 0.087; from 2^20 elements up the 4096-element tiles win (2^20: 0.139 vs 0.151 ms).
        if (ctx->ntt_tile_log == 12 && ctx->ntt_adaptive_tiles && ((size_t)batch << log_n) < ((size_t)1 << 20))
            while (tile_log > p.log_r && (((size_t)batch << log_n) >> tile_log) < 512) tile_log--;
        unsigned log_c = tile_log > p.log_r ? tile_log - p.log_r : 0;
        unsigned avail = last ? (P > 1 ? radices[0] : 0) : p.log_s;
        if (log_c > avail) log_c = avail;
        p.log_c = log_c;
        PLONK_TRY(get_small_tw(ctx, p.log_r, inverse, &p.small_tw));
        p.tw_lo = lo;
        p.tw_hi = hi;
        p.in_scale = first ? in_scale : nullptr;
        p.out_scale = last ? out_scale : nullptr;
        if (last && scale_by_n_inv) {
            Fr nn = host_fr_from_u64((uint64_t)N);
            p.out_scalar = fp_inv(nn);
            p.has_out_scalar = 1;
        }
        p.nprev = last ? pi : 0;
        for (unsigned q = 0; q < pi && q < 3; q++) p.prev_log_r[q] = radices[q];

        const unsigned R = 1u << p.log_r, C = 1u << p.log_c, T = R * C;
        // in-LDS levels: radix 8 while three bits remain, then 4 or 2
        p.n_levels = 0;
        p.level_radices = 0;
        for (unsigned bits = p.log_r; bits;) {
            unsigned lr = bits >= 3 ? 3 : bits;
            p.level_radices |= lr << (2 * p.n_levels++);
            bits -= lr;
        }
        Fr w8 = host_root_of_unity(3, inverse);
        p.w8_1 = w8;
        p.w8_2 = fp_sqr(w8);
        p.w8_3 = fp_mul(p.w8_2, w8);
        const bool stockham = ctx->ntt_kind == 2 || ((ctx->ntt_kind == 0 || ctx->ntt_kind >= 3) && P == 1);
        unsigned nthr = stockham ? T / 8 : T / 4;
        if (nthr < 64) nthr = 64;
        if (nthr > (stockham ? 512u : 1024u)) nthr = stockham ? 512 : 1024;
        const unsigned row_pitch = last ? (R + (C > 1 ? 1 : 0)) : 0;
        const size_t t_pad = last ? (size_t)row_pitch * C : T;
        const size_t tw_bytes = 32 * (size_t)(R / 2 ? R / 2 : 1);
        // keep two workgroups per CU when the tile allows it: twiddles go to LDS only if they fit in 80 KiB
        p.tw_in_lds = !stockham || ((32 * t_pad + tw_bytes <= 80 * 1024 || 32 * t_pad > 80 * 1024) && (32 * t_pad + tw_bytes <= 160 * 1024));
        const size_t shmem = 32 * t_pad + (p.tw_in_lds ? tw_bytes : 32);
        const size_t tiles = N / T;
        if (!ctx->ntt_attr_set) {  // a per-device attribute: tracked per context, not per process
            PLONK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ntt_pass_stockham_kernel),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
            PLONK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ntt_pass_radix2_kernel),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
            ctx->ntt_attr_set = true;
        }
        // algorithmic bytes of a size-N transform: 64 * N (read once, write once), split over its passes
        PLONK_TRY(prof_begin(ctx, "ntt_pass", 64.0 * (double)N * (double)batch / (double)P));
        if (stockham) PLONK_LAUNCH(ntt_pass_stockham_kernel, dim3((unsigned)tiles, (unsigned)batch), dim3(nthr), shmem, ctx->stream, p);
        else PLONK_LAUNCH(ntt_pass_radix2_kernel, dim3((unsigned)tiles, (unsigned)batch), dim3(nthr), shmem, ctx->stream, p);
        PLONK_TRY(prof_end(ctx));
        PLONK_CHECK_HIP(hipGetLastError());
        log_h += p.log_r;
    }
    return PLONK_OK;
}
