// ntt.hip — BN254-Fr number-theoretic transform engine (forward / inverse / coset), natural order
// in and out, batched, any size 2^0 .. 2^28.
//
// Reference behaviour replaced: Polynomial.fft / ifft (/root/reference/poly.py:113-148, the
// recursive `_fft` at 117-127), to_coset_extended_lagrange (poly.py:156-163) and
// coset_extended_lagrange_to_coeffs (poly.py:169-177).  The transform is the plain DFT
// X[k] = sum_j x[j] w^(jk), w = 5^((r-1)/N) (curve.py:14-16); the inverse uses w^-1 and 1/N.
//
// Kernels: the in-register wave kernels (ntt_wave.h, driven by ntt_wave_host.h) serve 2^7 .. 2^13 in one launch and
// 2^14 .. 2^26 in two.  What they do not cover (sizes below 2^7, above 2^26, and the multi-pass plans the tests force with
// plonk_ntt_configure) runs on ntt_pass_radix2_kernel below: N = R1*R2*..*RP (P <= 4 passes), pass p runs Rp-point
// sub-transforms for a tile of C adjacent columns in LDS (two 16-byte planes: unit-stride lanes are conflict-free) as
// radix-2 stages, multiplies by the inter-pass twiddle w_N^(H*j*k) (two-level table) on the way out; the last pass reads
// whole rows and performs the digit-reversing write that restores natural order.  Coset scaling, zero padding, the 1/N
// factor and the inverse-coset scaling are fused into the first-pass load / last-pass store of either family.
// (Round 1's Stockham radix-8 LDS kernel — 198 VGPRs, two waves per SIMD — served only sizes below 2^7 after round 3 and
// was removed in round 4.)
#include <string.h>

#include "ntt_wave_host.h"

typedef FpL<FrParams> FrL;
typedef FpLS<FrParams> FrLS;
typedef NttWaveT<FrParams> NttWave;

// defaults live in plonk_ctx (ntt_tile_log = 12: 4096 elements = 128 KiB of LDS; ntt_single_log = 11;
// ntt_radix_log = 10) and can be changed with plonk_ntt_configure for tuning / small-size tests.

struct NttPass {
    const Fr* in;
    Fr* out;
    size_t in_bstride, out_bstride;
    unsigned log_n, log_r, log_c;
    unsigned log_h;  // product of the earlier passes' radices
    unsigned log_s;  // element stride of this pass's digit
    unsigned first, last;
    unsigned in_len;
    const Fr* small_tw;  // w_R^k, k < R/2 (global; staged into LDS)
    const Fr* tw_lo;
    const Fr* tw_hi;
    const Fr* in_scale;
    const Fr* out_scale;
    Fr out_scalar;
    unsigned has_out_scalar;
    unsigned nprev;
    unsigned prev_log_r[3];
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

// ------------------------------------------------------------------------------------------------
// The LDS kernel: radix-2 Gentleman-Sande stages, two levels per LDS round trip, bit-reversed read at the end.
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
    if (!plonk_dev_malloc(&p, (n ? n : 1) * sizeof(Fr))) {
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

// the wave kernels (variant C): N = 2^7 .. 2^13 in one pass (one workgroup of N / 2, N / 4 or N / 8 threads per transform), and
// N = R1 R2 with R1, R2 from that set in two passes (columns, then rows).  Default splits: the fastest measured on MI355X
// for a lone transform (profiles/r03_b_ntt_splits.jsonl: the 4-element-per-thread kernels where a size allows them —
// twice the waves —, and short column transforms for the largest sizes); plonk_ntt_set_split overrides one size.
// latency: the call is a small job (<= 2^18 elements in all) — bound by the instruction chain of one wave, not by
// throughput: 2^16 .. 2^18 then split so that the two-element kernels (2^7, and 2^9 in its two-element form) serve them.
bool ntt_wave_plan(const plonk_ctx* ctx, unsigned log_n, bool latency, unsigned* log_r1, unsigned* log_r2) {
    if (log_n >= 7 && log_n <= 13) {
        *log_r1 = log_n;
        *log_r2 = 0;
        return true;
    }
    if (log_n < 14 || log_n > 26) return false;
    //                                   2^14 15  16 17  18  19  20  21  22  23  24  25  26
    static const unsigned char best[] = {7,   7,  8,  8, 10, 10, 10, 11, 11, 10, 11, 12, 13};  // (profiles/r03_r_ntt_splits_final.jsonl: every admissible split, final kernels)
    static const unsigned char lat[] = {7,    7,  7,  8,  9};                                   // (profiles/r04_c_ntt_latency_splits.jsonl)
    unsigned r1 = (latency && log_n <= 18) ? lat[log_n - 14] : best[log_n - 14];
    if (ctx && log_n < sizeof ctx->ntt_split / sizeof ctx->ntt_split[0] && ctx->ntt_split[log_n]) r1 = ctx->ntt_split[log_n];
    if (r1 < 7 || r1 > 13 || log_n - r1 < 7 || log_n - r1 > 13) return false;
    *log_r1 = r1;
    *log_r2 = log_n - r1;
    return true;
}

// ---- the wave kernels' host side for BN254 Fr: ntt_wave_host.h over this context's tables ---------------------------------
struct Bn254FrField {
    typedef FrParams P;
    static Fr root_of_unity(unsigned log_n, bool inverse) { return host_root_of_unity(log_n, inverse); }
    static Fr from_u64(uint64_t x) { return host_fr_from_u64(x); }
    static int packed_roots(plonk_ctx* ctx, unsigned log_n, bool inverse, const Fr** out) { return ntt_get_roots(ctx, log_n, inverse, out); }
    static int packed_lo_hi(plonk_ctx* ctx, unsigned log_n, bool inverse, const Fr** lo, const Fr** hi) { return get_lo_hi(ctx, log_n, inverse, lo, hi); }
    static int powers(plonk_ctx* ctx, const Fr& base, const Fr& first, Fr* out, size_t n) { return k_fr_powers(ctx, base, first, out, n); }
    static WaveTables& tables(plonk_ctx* ctx) { return ctx->tw.wave; }
};

static int ntt_run_wave(plonk_ctx* ctx, const Fr* in, Fr* out, unsigned log_n, bool inverse, size_t batch, size_t in_len,
                        size_t in_bstride, size_t out_bstride, const Fr* in_scale, const Fr* out_scale, bool scale_by_n_inv, const NttFan* fan) {
    return wave_run<Bn254FrField>(ctx, in, out, log_n, inverse, batch, in_len, in_bstride, out_bstride, in_scale, out_scale, scale_by_n_inv, fan);
}

// ---- distributed four-step transform: the two local steps (the all-to-all between them is comm.hip's) ------------------
// N = R1 R2 over W = 2^log_w ranks.  Rank g holds the columns c = g R2/W .. of the R1 x R2 matrix x[i1 R2 + c] as
// [R1][R2/W]; ntt_dist_columns leaves (k1, c) * w_N^(c k1) in the same layout, whose block of rows k1 = h R1/W .. is what
// rank h needs; ntt_dist_rows takes the W received blocks [source rank][R1/W][R2/W] and leaves the frequencies
// k1 + R1 k2 of its rows as [R2][R1/W] (times 1/N for the inverse).
int ntt_dist_plan(unsigned log_n, unsigned log_w, unsigned* log_r1, unsigned* log_r2) {
    unsigned r1 = 0, r2 = 0;
    // the default split (no per-context override: every rank must pick the same one)
    PLONK_REQUIRE(ntt_wave_plan(nullptr, log_n, false, &r1, &r2) && r2 && log_n >= 16, PLONK_ERR_ARG,
                  "distributed NTT supports sizes 2^16 .. 2^26 (got 2^%u)", log_n);
    PLONK_REQUIRE(log_w + 5 <= r2 && log_w + 5 <= r1, PLONK_ERR_ARG, "2^%u ranks are too many for a 2^%u-point transform", log_w, log_n);
    *log_r1 = r1;
    *log_r2 = r2;
    return PLONK_OK;
}

static void ntt_wave_consts(NttWave* p, unsigned log_n, bool inverse) { wave_params_init<Bn254FrField>(p, log_n, inverse); }

int ntt_dist_columns(plonk_ctx* ctx, const Fr* in, Fr* out, unsigned log_n, unsigned log_w, unsigned rank, bool inverse) {
    unsigned log_r1, log_r2;
    PLONK_TRY(ntt_dist_plan(log_n, log_w, &log_r1, &log_r2));
    const unsigned log_cl = log_r2 - log_w;
    NttWave a;
    ntt_wave_consts(&a, log_n, inverse);
    PLONK_TRY(wave_lo_hi<Bn254FrField>(ctx, log_n, inverse, false, &a.tw_lo, &a.tw_hi));
    PLONK_TRY(wave_program_table<Bn254FrField>(ctx, log_r1, wavel_log_e(log_r1), inverse, &a.roots));
    a.mode = 1;
    a.log_other = log_cl;
    a.sub_base = rank << log_cl;
    a.in = in;
    a.out = out;
    a.in_len = 1u << (log_n - log_w);
    PLONK_TRY(prof_begin(ctx, "ntt_pass", 32.0 * (double)((size_t)1 << (log_n - log_w))));
    PLONK_TRY(wave_launch<Bn254FrField>(ctx, a, log_r1, wavel_log_e(log_r1), 1u << log_cl, 1));
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
    PLONK_TRY(wave_program_table<Bn254FrField>(ctx, log_r2, wavel_log_e(log_r2), inverse, &c.roots));
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
    PLONK_TRY(wave_launch<Bn254FrField>(ctx, c, log_r2, wavel_log_e(log_r2), 1u << log_kl, 1));
    PLONK_TRY(prof_end(ctx));
    PLONK_CHECK_HIP(hipGetLastError());
    return PLONK_OK;
}

int ntt_run(plonk_ctx* ctx, const Fr* in, Fr* out, unsigned log_n, bool inverse, size_t batch, size_t in_len,
            size_t in_bstride, size_t out_bstride, const Fr* in_scale, const Fr* out_scale, bool scale_by_n_inv, const NttFan* fan) {
    PLONK_REQUIRE(log_n <= PLONK_FR_TWO_ADICITY, PLONK_ERR_ARG, "NTT size 2^%u exceeds the 2-adicity (28) of BN254 Fr", log_n);
    if (!batch) return PLONK_OK;
    const size_t N = (size_t)1 << log_n;
    unsigned wr1, wr2;
    if (fan && fan->count <= 1) fan = nullptr;
    const bool wave_ok = (ctx->ntt_kind == 0 || ctx->ntt_kind >= 5) && ntt_wave_plan(ctx, log_n, false, &wr1, &wr2) &&
                         ((ctx->ntt_single_log >= 11 && ctx->ntt_radix_log >= 10) || ctx->ntt_kind >= 5);
    if (fan && !(wave_ok && !wr2 && wave_fan_ok(*fan, N, out_scale != nullptr))) {  // no single launch for this size / kernel choice: one call per copy
        for (unsigned f = 0; f < fan->count; f++)
            PLONK_TRY(ntt_run(ctx, in + (size_t)f * fan->in_stride, out + (size_t)f * fan->out_stride, log_n, inverse, batch, in_len, in_bstride, out_bstride,
                              in_scale ? in_scale + (size_t)f * fan->scale_stride : nullptr, out_scale ? out_scale + (size_t)f * fan->scale_stride : nullptr,
                              scale_by_n_inv, nullptr));
        return PLONK_OK;
    }
    // The wave kernels serve every size they cover (2^7 .. 2^13 in one launch, 2^14 .. 2^26 in two): measured on MI355X
    // they beat the LDS kernels at every such size and batch (profiles/r02_e_ntt_kinds.json, r02_v_ntt_kinds.json, r03_*;
    // 2^14 / 2^15 since round 4's two-element kernels).  kind 4 = automatic choice among the LDS kernels only (A/B runs); a
    // plonk_ntt_configure with small tiles (the multi-pass tests) also keeps a transform on the LDS kernels.
    if (wave_ok) return ntt_run_wave(ctx, in, out, log_n, inverse, batch, in_len, in_bstride, out_bstride, in_scale, out_scale, scale_by_n_inv, fan);
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
        unsigned nthr = T / 4;
        if (nthr < 64) nthr = 64;
        if (nthr > 1024u) nthr = 1024;
        const unsigned row_pitch = last ? (R + (C > 1 ? 1 : 0)) : 0;
        const size_t t_pad = last ? (size_t)row_pitch * C : T;
        const size_t tw_bytes = 32 * (size_t)(R / 2 ? R / 2 : 1);
        const size_t shmem = 32 * t_pad + tw_bytes;
        const size_t tiles = N / T;
        if (!ctx->ntt_attr_set) {  // a per-device attribute: tracked per context, not per process
            PLONK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ntt_pass_radix2_kernel),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
            ctx->ntt_attr_set = true;
        }
        // algorithmic bytes of a size-N transform: 64 * N (read once, write once), split over its passes
        PLONK_TRY(prof_begin(ctx, "ntt_pass", 64.0 * (double)N * (double)batch / (double)P));
        PLONK_LAUNCH(ntt_pass_radix2_kernel, dim3((unsigned)tiles, (unsigned)batch), dim3(nthr), shmem, ctx->stream, p);
        PLONK_TRY(prof_end(ctx));
        PLONK_CHECK_HIP(hipGetLastError());
        log_h += p.log_r;
    }
    return PLONK_OK;
}
