// fr_ops.hip — element-wise Fr kernels: Montgomery conversion, pointwise + - * /, scalar
// broadcast, rotation, batch inversion, power tables, barycentric evaluation.
//
// Reference behaviour replaced: /root/reference/poly.py:23-109 (operators, shift) and
// poly.py:181-195 (barycentric_eval).  These are streaming kernels: one 32-byte element per lane
// per access (two global_load_dwordx4), grid-stride over at most 2048 workgroups of 256 lanes;
// bounded by HBM for add/sub and by the integer ALU for mul/div.
#include <string.h>

#include "plonk_internal.h"

static inline dim3 grid_for(size_t n, unsigned block = 256) {
    size_t g = (n + block - 1) / block;
    if (g > 2048) g = 2048;
    if (g == 0) g = 1;
    return dim3((unsigned)g);
}

// ------------------------------------------------------------------------------------------------
__global__ void fr_convert_kernel(const Fr* in, Fr* out, size_t n, int to_mont) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Fr a = fp_load(in + i);
        fp_store(out + i, to_mont ? fp_to_mont(a) : fp_from_mont(a));
    }
}

// canonical-range check fused with the conversion: first offending index (or ~0) lands in *bad
__global__ void fr_to_mont_checked_kernel(const Fr* in, Fr* out, size_t n, unsigned long long* bad) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Fr a = fp_load(in + i);
        uint32_t br = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) (void)fp_sbb(a.v[k], FrParams::mod(k), br);  // a - r borrows  <=>  a < r
        if (!br) atomicMin(bad, (unsigned long long)i);
        fp_store(out + i, fp_to_mont(a));
    }
}

// in-place Montgomery conversion of freshly uploaded values; *first_bad = index of the first element >= r, or ~0
int k_fr_to_mont_checked(plonk_ctx* ctx, Fr* data, size_t n, unsigned long long* d_first_bad) {
    if (!n) return PLONK_OK;
    PLONK_CHECK_HIP(hipMemsetAsync(d_first_bad, 0xff, sizeof(unsigned long long), ctx->stream));
    PLONK_LAUNCH(fr_to_mont_checked_kernel, grid_for(n), dim3(256), 0, ctx->stream, (const Fr*)data, data, n, d_first_bad);
    PLONK_CHECK_HIP(hipGetLastError());
    return PLONK_OK;
}

int k_fr_to_mont(plonk_ctx* ctx, const Fr* in, Fr* out, size_t n) {
    if (!n) return PLONK_OK;
    PLONK_LAUNCH(fr_convert_kernel, grid_for(n), dim3(256), 0, ctx->stream, in, out, n, 1);
    PLONK_CHECK_HIP(hipGetLastError());
    return PLONK_OK;
}
int k_fr_from_mont(plonk_ctx* ctx, const Fr* in, Fr* out, size_t n) {
    if (!n) return PLONK_OK;
    PLONK_LAUNCH(fr_convert_kernel, grid_for(n), dim3(256), 0, ctx->stream, in, out, n, 0);
    PLONK_CHECK_HIP(hipGetLastError());
    return PLONK_OK;
}

// ------------------------------------------------------------------------------------------------
// out[i] = a[i] (op) b[i]        poly.py:23-36, 45-58, 68-77
__global__ void fr_pointwise_kernel(int op, const Fr* a, const Fr* b, Fr* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Fr x = fp_load(a + i), y = fp_load(b + i), r;
        if (op == PLONK_OP_ADD) r = fp_add(x, y);
        else if (op == PLONK_OP_SUB) r = fp_sub(x, y);
        else r = fp_mul(x, y);
        fp_store(out + i, r);
    }
}

// out[i] = a[i] (op) s for i < limit, a[i] otherwise   (limit = 1: MONOMIAL constant-term rule)
__global__ void fr_scalar_kernel(int op, const Fr* a, Fr s, Fr* out, size_t n, size_t limit) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Fr x = fp_load(a + i), r = x;
        if (i < limit) {
            if (op == PLONK_OP_ADD) r = fp_add(x, s);
            else if (op == PLONK_OP_SUB) r = fp_sub(x, s);
            else r = fp_mul(x, s);
        }
        fp_store(out + i, r);
    }
}

int k_fr_pointwise(plonk_ctx* ctx, int op, const Fr* a, const Fr* b, Fr* out, size_t n) {
    if (!n) return PLONK_OK;
    PLONK_LAUNCH(fr_pointwise_kernel, grid_for(n), dim3(256), 0, ctx->stream, op, a, b, out, n);
    PLONK_CHECK_HIP(hipGetLastError());
    return PLONK_OK;
}

int k_fr_pointwise_scalar(plonk_ctx* ctx, int op, const Fr* a, const Fr& s_mont, Fr* out, size_t n, size_t limit) {
    if (!n) return PLONK_OK;
    PLONK_LAUNCH(fr_scalar_kernel, grid_for(n), dim3(256), 0, ctx->stream, op, a, s_mont, out, n, limit);
    PLONK_CHECK_HIP(hipGetLastError());
    return PLONK_OK;
}

// ------------------------------------------------------------------------------------------------
// out[i] = constant + sum_k scalar[k] * term[k][i]: a whole run of `Polynomial * Scalar`, `+`, `-` (the linearisation
// polynomial R and the opening numerators of prover.py:245-288 are such runs: ~40 operator calls, each a launch and a
// pass over 4n elements) as ONE pass — every term read once, one store.
#define FR_LINCOMB_MAX 20
struct FrLincomb {
    const Fr* term[FR_LINCOMB_MAX];
    Fr scalar[FR_LINCOMB_MAX];
    Fr constant;
    unsigned n_terms;
};
__global__ void fr_lincomb_kernel(FrLincomb a, Fr* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Fr acc = a.constant;
        for (unsigned k = 0; k < a.n_terms; k++) acc = fp_add(acc, fp_mul(fp_load(a.term[k] + i), a.scalar[k]));
        fp_store(out + i, acc);
    }
}
int k_fr_lincomb(plonk_ctx* ctx, const Fr* const* terms, const Fr* scalars_mont, unsigned n_terms, const Fr& constant_mont, Fr* out, size_t n) {
    if (!n) return PLONK_OK;
    FrLincomb a;
    memset(&a, 0, sizeof a);
    for (unsigned k = 0; k < n_terms; k++) {
        a.term[k] = terms[k];
        a.scalar[k] = scalars_mont[k];
    }
    a.constant = constant_mont;
    a.n_terms = n_terms;
    PLONK_LAUNCH(fr_lincomb_kernel, grid_for(n), dim3(256), 0, ctx->stream, a, out, n);
    PLONK_CHECK_HIP(hipGetLastError());
    return PLONK_OK;
}

// ------------------------------------------------------------------------------------------------
// out[b][i] = in[b][(i + shift) mod n]        poly.py:102-109
__global__ void fr_rotate_kernel(const Fr* in, Fr* out, size_t n, size_t shift, size_t total) {
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
        size_t b = g / n, i = g - b * n;
        size_t j = i + shift;
        if (j >= n) j -= n;
        fp_store(out + g, fp_load(in + b * n + j));
    }
}

int k_fr_rotate(plonk_ctx* ctx, const Fr* in, Fr* out, size_t n, size_t shift, size_t batch) {
    if (!n || !batch) return PLONK_OK;
    PLONK_LAUNCH(fr_rotate_kernel, grid_for(n * batch), dim3(256), 0, ctx->stream, in, out, n, shift, n * batch);
    PLONK_CHECK_HIP(hipGetLastError());
    return PLONK_OK;
}

// ------------------------------------------------------------------------------------------------
// Batch inversion (Montgomery's trick) — one field inversion (fp_inv) per lane-chunk of INV_CHUNK elements.
// Zeros are skipped in the running product and map to zero (py_ecc: x / 0 == 0).
#define INV_CHUNK 8
__global__ void __launch_bounds__(64) fr_batch_inverse_kernel(const Fr* in, Fr* out, size_t n) {
    size_t nchunks = (n + INV_CHUNK - 1) / INV_CHUNK;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < nchunks; c += (size_t)gridDim.x * blockDim.x) {
        size_t base = c * INV_CHUNK;
        // prefix products in registers (compile-time indices); the values themselves are read again on the way back instead of
        // being held — 64 VGPRs instead of 128, and no scratch (round 3: 528 B per lane).  in == out is fine: element k is
        // written after its second read and before nothing that still needs it.
        Fr pre[INV_CHUNK];
        Fr acc = fp_one<FrParams>();
        wave_for<INV_CHUNK>([&](auto K) {
            constexpr unsigned k = decltype(K)::value;
            const Fr v = (base + k < n) ? fp_load(in + base + k) : fp_zero<FrParams>();
            pre[k] = acc;
            if (!fp_is_zero(v)) acc = fp_mul(acc, v);
        });
        acc = fp_inv(acc);
        wave_for_down<INV_CHUNK>([&](auto K) {
            constexpr unsigned k = decltype(K)::value;
            if (base + k < n) {
                const Fr v = fp_load(in + base + k);
                Fr r = fp_zero<FrParams>();
                if (!fp_is_zero(v)) {
                    r = fp_mul(acc, pre[k]);
                    acc = fp_mul(acc, v);
                }
                fp_store(out + base + k, r);
            }
        });
    }
}

int k_fr_batch_inverse(plonk_ctx* ctx, const Fr* in, Fr* out, size_t n) {
    if (!n) return PLONK_OK;
    size_t nchunks = (n + INV_CHUNK - 1) / INV_CHUNK;
    PLONK_LAUNCH(fr_batch_inverse_kernel, grid_for(nchunks, 64), dim3(64), 0, ctx->stream, in, out, n);
    PLONK_CHECK_HIP(hipGetLastError());
    return PLONK_OK;
}

// ------------------------------------------------------------------------------------------------
// out[i] = first * base^i.  Each lane raises base to its chunk start, then walks POW_CHUNK steps.
#define POW_CHUNK 16
__global__ void fr_powers_kernel(Fr base, Fr first, Fr* out, size_t n) {
    size_t nchunks = (n + POW_CHUNK - 1) / POW_CHUNK;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < nchunks; c += (size_t)gridDim.x * blockDim.x) {
        size_t start = c * POW_CHUNK;
        Fr cur = fp_mul(first, fp_pow_u64(base, (uint64_t)start));
        for (int k = 0; k < POW_CHUNK && start + k < n; k++) {
            fp_store(out + start + k, cur);
            cur = fp_mul(cur, base);
        }
    }
}

// number of positions where a and b differ (b null: where a is non-zero) — the reference's `==` on two Polynomials /
// its `values[k:] == [0] * m` checks (poly.py:20-21, prover.py:205-208, 288, 299) without moving the vectors to the host
__global__ void fr_count_diff_kernel(const Fr* a, const Fr* b, size_t n, unsigned long long* count) {
    unsigned long long mine = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const Fr x = fp_load(a + i);
        mine += b ? !fp_eq(x, fp_load(b + i)) : !fp_is_zero(x);
    }
    if (mine) atomicAdd(count, mine);
}

int k_fr_count_diff(plonk_ctx* ctx, const Fr* a, const Fr* b, size_t n, unsigned long long* d_count) {
    PLONK_CHECK_HIP(hipMemsetAsync(d_count, 0, sizeof *d_count, ctx->stream));
    PLONK_LAUNCH(fr_count_diff_kernel, grid_for(n, 256), dim3(256), 0, ctx->stream, a, b, n, d_count);
    PLONK_CHECK_HIP(hipGetLastError());
    return PLONK_OK;
}

int k_fr_powers(plonk_ctx* ctx, const Fr& base_mont, const Fr& first_mont, Fr* out, size_t n) {
    if (!n) return PLONK_OK;
    size_t nchunks = (n + POW_CHUNK - 1) / POW_CHUNK;
    PLONK_LAUNCH(fr_powers_kernel, grid_for(nchunks, 64), dim3(64), 0, ctx->stream, base_mont, first_mont, out, n);
    PLONK_CHECK_HIP(hipGetLastError());
    return PLONK_OK;
}

// ------------------------------------------------------------------------------------------------
// Barycentric evaluation, poly.py:181-195:
//     P(x) = (x^N - 1)/N * sum_i v_i * w^i / (x - w^i)
// One workgroup per polynomial.  Lane t owns BARY_CHUNK consecutive i: it forms d_i = x - w^i,
// inverts the chunk with Montgomery's trick (a zero d_i contributes 0, matching py_ecc's x/0 == 0),
// accumulates v_i * w^i * d_i^-1 and the block tree-reduces the partial sums through LDS.
// `roots` is the full table w^0..w^(N-1).  x values: one per polynomial (x_stride = 1) or shared (0).
#define BARY_CHUNK 8
#define FR_BARY_MANY_MAX 16
struct FrBaryPtrs { const Fr* p[FR_BARY_MANY_MAX]; };  // vals == null: polynomial b lives at ptrs.p[b] (plonk_fr_barycentric_many)
__global__ void __launch_bounds__(256) fr_barycentric_kernel(const Fr* vals, FrBaryPtrs ptrs, const Fr* roots, unsigned log_n, const Fr* xs, size_t x_stride,
                                      Fr n_inv, Fr* out) {
    const size_t n = (size_t)1 << log_n;
    const Fr* v = vals ? vals + (size_t)blockIdx.x * n : ptrs.p[blockIdx.x];
    const Fr x = fp_load(xs + (size_t)blockIdx.x * x_stride);
    Fr sum = fp_zero<FrParams>();
    size_t nchunks = (n + BARY_CHUNK - 1) / BARY_CHUNK;
    for (size_t c = threadIdx.x; c < nchunks; c += blockDim.x) {
        size_t base = c * BARY_CHUNK;
        Fr pre[BARY_CHUNK];  // (as in fr_batch_inverse_kernel: prefix products in registers, d_i formed again on the way back)
        Fr acc = fp_one<FrParams>();
        wave_for<BARY_CHUNK>([&](auto K) {
            constexpr unsigned k = decltype(K)::value;
            const Fr d = (base + k < n) ? fp_sub(x, fp_load(roots + base + k)) : fp_zero<FrParams>();
            pre[k] = acc;
            if (!fp_is_zero(d)) acc = fp_mul(acc, d);
        });
        acc = fp_inv(acc);
        wave_for_down<BARY_CHUNK>([&](auto K) {
            constexpr unsigned k = decltype(K)::value;
            if (base + k < n) {
                const Fr w = fp_load(roots + base + k), d = fp_sub(x, w);
                if (!fp_is_zero(d)) {
                    const Fr dinv = fp_mul(acc, pre[k]);
                    acc = fp_mul(acc, d);
                    sum = fp_add(sum, fp_mul(fp_mul(fp_load(v + base + k), w), dinv));
                }
            }
        });
    }
    __shared__ Fr red[256];
    red[threadIdx.x] = sum;
    __syncthreads();
    for (unsigned s = blockDim.x / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fp_add(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        Fr xn = x;
        for (unsigned i = 0; i < log_n; i++) xn = fp_sqr(xn);
        Fr lead = fp_mul(fp_sub(xn, fp_one<FrParams>()), n_inv);
        fp_store(out + blockIdx.x, fp_mul(lead, red[0]));
    }
}

int k_fr_barycentric(plonk_ctx* ctx, const Fr* vals, const Fr* roots, unsigned log_n, const Fr* xs_dev,
                     size_t x_stride, const Fr& n_inv_mont, Fr* out_dev, size_t n_polys) {
    if (!n_polys) return PLONK_OK;
    size_t n = (size_t)1 << log_n;
    unsigned block = 256;
    while (block > 64 && (size_t)block * BARY_CHUNK > n) block >>= 1;
    FrBaryPtrs none;
    memset(&none, 0, sizeof none);
    PLONK_LAUNCH(fr_barycentric_kernel, dim3((unsigned)n_polys), dim3(block), 0, ctx->stream, vals, none, roots, log_n,
                 xs_dev, x_stride, n_inv_mont, out_dev);
    PLONK_CHECK_HIP(hipGetLastError());
    return PLONK_OK;
}
// the same for up to FR_BARY_MANY_MAX polynomials that live in separate buffers, each with its own point
int k_fr_barycentric_ptrs(plonk_ctx* ctx, const Fr* const* polys, const Fr* roots, unsigned log_n, const Fr* xs_dev, const Fr& n_inv_mont,
                          Fr* out_dev, size_t n_polys) {
    if (!n_polys) return PLONK_OK;
    size_t n = (size_t)1 << log_n;
    unsigned block = 256;
    while (block > 64 && (size_t)block * BARY_CHUNK > n) block >>= 1;
    FrBaryPtrs ptrs;
    memset(&ptrs, 0, sizeof ptrs);
    for (size_t k = 0; k < n_polys; k++) ptrs.p[k] = polys[k];
    PLONK_LAUNCH(fr_barycentric_kernel, dim3((unsigned)n_polys), dim3(block), 0, ctx->stream, (const Fr*)nullptr, ptrs, roots, log_n, xs_dev,
                 (size_t)1, n_inv_mont, out_dev);
    PLONK_CHECK_HIP(hipGetLastError());
    return PLONK_OK;
}
