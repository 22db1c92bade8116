// g1_ntt.hip — inverse DFT over the group: the Lagrange-basis SRS in n log n group operations.
//
// Reference behaviour replaced: Setup.commit = ifft + lincomb over powers_of_x (/root/reference/setup.py:66-72; README.md:166
// names the Lagrange-basis alternative).  With the SRS itself taken to the Lagrange basis,
//     [L_i(tau)]_1 = (1/n) sum_j w^(-ij) [tau^j]_1,        i < n = 2^log_n,
// a commitment of Lagrange values is one MSM with no inverse NTT in front of it.  msm.hip builds that view as n MSMs of size n
// (quadratic: fine at the prover's 2^11 where the lookup table makes an MSM cost microseconds, a wall above 2^12).  Here the same
// points come from a radix-2 decimation-in-frequency transform whose butterflies are group operations:
//     u' = u + v,      v' = (u - v) * w_2m^j                    (stage m = n/2, n/4, .., 1)
// i.e. log n stages of n/2 additions, n/2 subtractions and n/2 SCALAR MULTIPLICATIONS by a twiddle factor, then the factor 1/n
// (one more scalar multiplication per point), a bit-reversal and one batched conversion to affine.  The scalar multiplication is
// a fixed-window method on XYZZ coordinates: signed 4-bit digits, the multiples 1 P .. 8 P of the lane's own point in LDS,
// 4 doublings + at most one addition per digit — every lane of a wave runs the same instruction sequence whatever its scalar.
// A one-off per (SRS, size): the result is cached on the parent SRS like the MSM-built view, and both are bit-identical
// (tests: plonk_srs_lagrange under PLONK_LAGRANGE_SRS=msm / ntt at 2^0 .. 2^12).
#include <stdlib.h>
#include <string.h>

#include "plonk_internal.h"

#define G1NTT_BLOCK 64

// out[i] = XYZZ form of the affine base i
__global__ void g1ntt_load_kernel(const G1Affine* bases, G1Xyzz* out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Affine b;
    b.x = fp_load(&bases[i].x);
    b.y = fp_load(&bases[i].y);
    out[i] = g1_xyzz_from_affine(b);
}

// one DIF stage, additions only: for every pair (p, p + m): a[p] = u + v, a[p + m] = u - v
__global__ void g1ntt_butterfly_kernel(G1Xyzz* a, size_t n, size_t m) {
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n / 2) return;
    const size_t p = (q / m) * 2 * m + (q % m);
    const G1Xyzz u = a[p], v = a[p + m];
    G1Xyzz s = u, d = u, nv = v;
    nv.y = fp_neg(v.y);
    g1_add(s, v);
    g1_add(d, nv);
    a[p] = s;
    a[p + m] = d;
}

// signed 4-bit digits of a canonical 256-bit scalar below 2^254, low digit first: digit j is stored biased (d + 8, in [0, 16))
// in nibble j of the 8 output words.  d in [-8, 8); the top digit needs no carry out because the scalar's top nibble is <= 3.
PLONK_DEV void g1ntt_recode(const uint32_t k[8], uint32_t out[8]) {
    uint32_t carry = 0;
#pragma unroll
    for (int wd = 0; wd < 8; wd++) {
        uint32_t o = 0;
#pragma unroll
        for (int nb = 0; nb < 8; nb++) {
            uint32_t d = ((k[wd] >> (4 * nb)) & 15u) + carry;  // 0 .. 16
            carry = d >= 8 ? 1u : 0u;
            d = (d + 8) & 15u;  // d - 16 carry + 8, reduced to a nibble: [-8, 8) biased by 8
            o |= d << (4 * nb);
        }
        out[wd] = o;
    }
}

// a[p] *= scalar, for the points this stage (or the final scaling) multiplies.
//   uniform == 0: thread q < n / 2 owns p = (q / m) 2 m + m + (q % m), scalar = tw[(q % m) * tw_stride] (the inverse root powers,
//                 Montgomery form); a factor of one is skipped.
//   uniform == 1: thread q < n owns p = q, scalar = tw[0] (the factor 1 / n).
__global__ void __launch_bounds__(G1NTT_BLOCK) g1ntt_scalar_mul_kernel(G1Xyzz* a, size_t n, size_t m, const Fr* tw, size_t tw_stride, int uniform) {
    __shared__ G1Xyzz tab[8 * G1NTT_BLOCK];  // tab[(d - 1) * 64 + lane] = d * P of that lane, d = 1 .. 8
    const size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned lane = threadIdx.x;
    const size_t count = uniform ? n : n / 2;
    const bool live = q < count;
    const size_t j = (!live || uniform) ? 0 : q % m;
    const size_t p = !live ? 0 : (uniform ? q : (q / m) * 2 * m + m + j);
    if (!live || (!uniform && j == 0)) return;  // (no barrier below: every lane only touches its own LDS column)
    const Fr kf = fp_from_mont(fp_load(tw + j * tw_stride));
    G1Xyzz P = a[p];
    if (g1_is_identity(P)) return;
    uint32_t rk[8];
    g1ntt_recode(kf.v, rk);
    {   // 1 P .. 8 P
        G1Xyzz t = P;
        tab[0 * G1NTT_BLOCK + lane] = t;       // 1
        g1_dbl(t);
        tab[1 * G1NTT_BLOCK + lane] = t;       // 2
        G1Xyzz t3 = t;
        g1_add(t3, P);
        tab[2 * G1NTT_BLOCK + lane] = t3;      // 3
        g1_dbl(t);
        tab[3 * G1NTT_BLOCK + lane] = t;       // 4
        G1Xyzz t5 = t;
        g1_add(t5, P);
        tab[4 * G1NTT_BLOCK + lane] = t5;      // 5
        g1_dbl(t3);
        tab[5 * G1NTT_BLOCK + lane] = t3;      // 6
        g1_add(t3, P);
        tab[6 * G1NTT_BLOCK + lane] = t3;      // 7
        g1_dbl(t);
        tab[7 * G1NTT_BLOCK + lane] = t;       // 8
    }
    G1Xyzz acc = g1_xyzz_identity();
#pragma unroll 1
    for (int dj = 63; dj >= 0; dj--) {
        g1_dbl(acc);
        g1_dbl(acc);
        g1_dbl(acc);
        g1_dbl(acc);
        uint32_t word = rk[0];  // rk[dj >> 3] without a dynamically indexed register array
#pragma unroll
        for (int wq = 1; wq < 8; wq++) word = (dj >> 3) == wq ? rk[wq] : word;
        const int d = (int)((word >> (4 * (dj & 7))) & 15u) - 8;
        if (d) {
            const unsigned ad = d < 0 ? (unsigned)-d : (unsigned)d;
            G1Xyzz t = tab[(ad - 1) * G1NTT_BLOCK + lane];
            if (d < 0) t.y = fp_neg(t.y);
            g1_add(acc, t);
        }
    }
    a[p] = acc;
}

// out[i] = in[bit-reverse(i)]
__global__ void g1ntt_bitrev_kernel(const G1Xyzz* in, G1Xyzz* out, unsigned log_n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ((size_t)1 << log_n)) return;
    size_t r = 0;
    for (unsigned b = 0; b < log_n; b++) r |= ((i >> b) & 1) << (log_n - 1 - b);
    out[i] = in[r];
}

static Fr g1ntt_fr_u64(uint64_t x) {  // small integer -> Montgomery form (host)
    Fr a = fp_zero<FrParams>();
    a.v[0] = (uint32_t)x;
    a.v[1] = (uint32_t)(x >> 32);
    return fp_to_mont(a);
}

// d_bases_out[i] = [L_i(tau)]_1 (affine, Montgomery coordinates, identity = (0, 0)) from the first n = 2^log_n bases of `srs`
int g1_lagrange_by_ntt(plonk_ctx* ctx, const plonk_srs* srs, unsigned log_n, G1Affine* d_bases_out) {
    const size_t n = (size_t)1 << log_n;
    G1Xyzz *a = nullptr, *b = nullptr;
    Fr* ninv = nullptr;
    auto cleanup = [&]() {
        if (a) hipFree(a);
        if (b) hipFree(b);
        if (ninv) hipFree(ninv);
    };
    if (!plonk_dev_malloc(&a, n * sizeof(G1Xyzz)) || !plonk_dev_malloc(&b, n * sizeof(G1Xyzz)) || !plonk_dev_malloc(&ninv, sizeof(Fr))) {
        cleanup();
        plonk_set_error("hipMalloc failed while building the Lagrange-basis SRS of size %zu (EC inverse NTT)", n);
        return PLONK_ERR_NOMEM;
    }
    const Fr* roots = nullptr;  // w^(-e), e < n
    int rc = log_n ? ntt_get_roots(ctx, log_n, true, &roots) : PLONK_OK;
    if (rc != PLONK_OK) {
        cleanup();
        return rc;
    }
    const unsigned g256 = (unsigned)((n + 255) / 256);
    PLONK_LAUNCH(g1ntt_load_kernel, dim3(g256), dim3(256), 0, ctx->stream, (const G1Affine*)srs->bases, a, n);
    for (size_t m = n / 2; m >= 1; m >>= 1) {
        const unsigned gb = (unsigned)((n / 2 + 63) / 64);
        PLONK_LAUNCH(g1ntt_butterfly_kernel, dim3(gb), dim3(64), 0, ctx->stream, a, n, m);
        if (m > 1)  // (the last stage's twiddles are all one)
            PLONK_LAUNCH(g1ntt_scalar_mul_kernel, dim3(gb), dim3(G1NTT_BLOCK), 0, ctx->stream, a, n, m, roots, n / (2 * m), 0);
    }
    if (log_n) {  // 1 / n
        Fr h = fp_inv(g1ntt_fr_u64((uint64_t)n));
        if (hipMemcpyAsync(ninv, &h, sizeof(Fr), hipMemcpyHostToDevice, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
            cleanup();
            plonk_set_error("copy of 1/n failed while building the Lagrange-basis SRS");
            return PLONK_ERR_HIP;
        }
        PLONK_LAUNCH(g1ntt_scalar_mul_kernel, dim3((unsigned)((n + 63) / 64)), dim3(G1NTT_BLOCK), 0, ctx->stream, a, n, (size_t)1, (const Fr*)ninv, (size_t)0, 1);
    }
    PLONK_LAUNCH(g1ntt_bitrev_kernel, dim3(g256), dim3(256), 0, ctx->stream, (const G1Xyzz*)a, b, log_n);
    g1_batch_to_affine(ctx, (const G1Xyzz*)b, d_bases_out, n);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
        cleanup();
        plonk_set_error("the EC inverse NTT of the SRS failed on the device");
        return PLONK_ERR_HIP;
    }
    cleanup();
    return PLONK_OK;
}
