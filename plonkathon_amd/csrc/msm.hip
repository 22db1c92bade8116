// msm.hip — batched fixed-base Pippenger multi-scalar multiplication on BN254 G1.
//
// Reference behaviour replaced: ec_lincomb -> lincomb -> multisubset
// (/root/reference/curve.py:38-111), i.e. everything Setup.commit does after its ifft
// (setup.py:66-72).  The reference bit-slices the scalars into 255 subsets and adds affine points
// with one Fq inversion per addition (~109k additions for N = 2^11); the result is a group element,
// so any correct schedule yields the same affine point.
//
// Schedule.  The bases are fixed (the SRS), so a window table
// T[w][i] = 2^(c*w) * P_i is built once per SRS; every window of every scalar then lands in ONE
// shared bucket set and no doublings remain in the per-MSM work:
//   1. msm_digits_kernel     scalar -> canonical -> + sum_w 2^(cw+c-1)  -> W raw digits u_w;
//                            signed digit d_w = u_w - 2^(c-1) in [-2^(c-1), 2^(c-1)).
//   2. msm_accumulate_kernel one workgroup per (MSM, window group).  Per window: LDS counting sort
//                            of the N digits by |d| (LDS atomics), then lane (bucket k, slice l)
//                            walks its share of bucket k's list doing XYZZ += affine mixed adds
//                            (sign folded into y).  Buckets live in registers / LDS, never in HBM.
//                            Afterwards the slices are merged, lane k forms k * B_k by
//                            double-and-add and the workgroup tree-reduces sum_k k*B_k through LDS.
//   3. msm_finalize_kernel   adds the window-group partials, converts to the unique affine
//                            representative (one Fermat inversion) and leaves canonical x||y.
// All table reads hit L2 / Infinity Cache (4 MiB table); the kernel is integer-ALU bound.
#include <string.h>

#include "plonk_internal.h"

#define MSM_BLOCK 256

// ------------------------------------------------------------------------------------------------
// Window table: table[w*n + i] = 2^(c*w) * bases[i], affine.
__global__ void msm_table_kernel(const G1Affine* bases, size_t n, unsigned c, unsigned W, G1Xyzz* tmp) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        G1Affine b;
        b.x = fp_load(&bases[i].x);
        b.y = fp_load(&bases[i].y);
        G1Xyzz p = g1_xyzz_from_affine(b);
        for (unsigned w = 0; w < W; w++) {
            tmp[(size_t)w * n + i] = p;
            if (w + 1 < W)
                for (unsigned k = 0; k < c; k++) g1_dbl(p);
        }
    }
}

// XYZZ -> affine for a whole array, Montgomery's trick over chunks of 8 (identity -> (0,0)).
#define AFF_CHUNK 8
__global__ void g1_batch_to_affine_kernel(const G1Xyzz* in, G1Affine* out, size_t n) {
    size_t nchunks = (n + AFF_CHUNK - 1) / AFF_CHUNK;
    for (size_t ch = (size_t)blockIdx.x * blockDim.x + threadIdx.x; ch < nchunks; ch += (size_t)gridDim.x * blockDim.x) {
        size_t base = ch * AFF_CHUNK;
        Fq den[AFF_CHUNK], pre[AFF_CHUNK];
        Fq acc = fp_one<FqParams>();
        for (int k = 0; k < AFF_CHUNK; k++) {
            den[k] = fp_zero<FqParams>();
            if (base + k < n) {
                const G1Xyzz& p = in[base + k];
                den[k] = fp_mul(fp_load(&p.zz), fp_load(&p.zzz));  // zero <=> identity
            }
            pre[k] = acc;
            if (!fp_is_zero(den[k])) acc = fp_mul(acc, den[k]);
        }
        acc = fp_inv(acc);
        for (int k = AFF_CHUNK - 1; k >= 0; k--) {
            if (base + k >= n) continue;
            G1Affine r = g1_affine_identity();
            if (!fp_is_zero(den[k])) {
                const G1Xyzz& p = in[base + k];
                Fq t = fp_mul(acc, pre[k]);  // 1 / (zz * zzz)
                acc = fp_mul(acc, den[k]);
                Fq zz = fp_load(&p.zz), zzz = fp_load(&p.zzz);
                r.x = fp_mul(fp_load(&p.x), fp_mul(t, zzz));
                r.y = fp_mul(fp_load(&p.y), fp_mul(t, zz));
            }
            fp_store(&out[base + k].x, r.x);
            fp_store(&out[base + k].y, r.y);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// digits[(m*W + w)*n + i] = raw c-bit digit u_w of (scalar_i + K), K = sum_w 2^(c*w + c - 1)
struct MsmRecode { uint32_t k[9]; };

__global__ void msm_digits_kernel(const Fr* scalars, size_t n, size_t M, size_t stride, unsigned c, unsigned W,
                                  MsmRecode rc, uint16_t* digits) {
    const size_t total = n * M;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (size_t)gridDim.x * blockDim.x) {
        const size_t m = g / n, i = g - m * n;
        Fr s = fp_from_mont(fp_load(scalars + m * stride + i));
        uint32_t limb[10];
        uint64_t carry = 0;
#pragma unroll
        for (int j = 0; j < 9; j++) {
            carry += (uint64_t)(j < 8 ? s.v[j] : 0) + rc.k[j];
            limb[j] = (uint32_t)carry;
            carry >>= 32;
        }
        limb[9] = 0;
        for (unsigned w = 0; w < W; w++) {
            unsigned bit = c * w, j = bit >> 5, sh = bit & 31;
            uint64_t two = (uint64_t)limb[j] | ((uint64_t)limb[j + 1] << 32);
            digits[(m * W + w) * n + i] = (uint16_t)((two >> sh) & ((1u << c) - 1));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LDS helpers for XYZZ points (plain layout; used once per window-group, not in the hot loop)
PLONK_DEV void lds_put(G1Xyzz* s, unsigned i, const G1Xyzz& p) { s[i] = p; }

__global__ void __launch_bounds__(MSM_BLOCK) msm_accumulate_kernel(const G1Affine* table, size_t table_n,
                                                                   const uint16_t* digits, size_t n, unsigned c,
                                                                   unsigned W, unsigned G, G1Xyzz* partial) {
    PLONK_DYN_SMEM(smem);
    const unsigned K = 1u << (c - 1);            // buckets 1..K
    const unsigned L = MSM_BLOCK / K ? MSM_BLOCK / K : 1;  // lanes per bucket (K <= MSM_BLOCK)
    const unsigned m = blockIdx.x / G, g = blockIdx.x % G;
    const unsigned w_begin = (unsigned)(((size_t)W * g) / G), w_end = (unsigned)(((size_t)W * (g + 1)) / G);
    const unsigned tid = threadIdx.x;

    // LDS carve-up: hist[K+1] | cursor[K+1] | sorted[n] (u16) | reduction scratch (XYZZ per thread)
    unsigned* hist = reinterpret_cast<unsigned*>(smem);
    unsigned* start = hist + (K + 1);
    unsigned* cursor = start + (K + 1);
    uint16_t* sorted = reinterpret_cast<uint16_t*>(cursor + (K + 1));
    size_t off = (size_t)(3 * (K + 1)) * 4 + n * 2;
    off = (off + 15) & ~(size_t)15;
    G1Xyzz* red = reinterpret_cast<G1Xyzz*>(smem + off);

    const unsigned my_bucket = tid / L + 1;  // 1..K (threads beyond K*L idle in the walk)
    const unsigned my_slice = tid % L;
    const bool walker = tid < K * L;
    G1Xyzz acc = g1_xyzz_identity();

    for (unsigned w = w_begin; w < w_end; w++) {
        const uint16_t* dg = digits + ((size_t)m * W + w) * n;
        for (unsigned k = tid; k <= K; k += MSM_BLOCK) hist[k] = 0;
        __syncthreads();
        for (size_t i = tid; i < n; i += MSM_BLOCK) {
            int d = (int)dg[i] - (int)K;
            unsigned a = d < 0 ? (unsigned)(-d) : (unsigned)d;
            atomicAdd(&hist[a], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            unsigned run = 0;
            for (unsigned k = 0; k <= K; k++) {
                start[k] = run;
                cursor[k] = run;
                run += hist[k];
            }
        }
        __syncthreads();
        for (size_t i = tid; i < n; i += MSM_BLOCK) {
            int d = (int)dg[i] - (int)K;
            unsigned a = d < 0 ? (unsigned)(-d) : (unsigned)d;
            if (a) {
                unsigned pos = atomicAdd(&cursor[a], 1u);
                sorted[pos] = (uint16_t)(i | (d < 0 ? 0x8000u : 0u));
            }
        }
        __syncthreads();
        if (walker) {
            const unsigned b = start[my_bucket], e = b + hist[my_bucket];
            const G1Affine* tw = table + (size_t)w * table_n;
            for (unsigned q = b + my_slice; q < e; q += L) {
                const unsigned ent = sorted[q];
                const G1Affine* src = tw + (ent & 0x7fffu);
                G1Affine pt;
                pt.x = fp_load(&src->x);
                pt.y = fp_load(&src->y);
                if (ent & 0x8000u) pt.y = fp_neg(pt.y);
                g1_madd(acc, pt);
            }
        }
        __syncthreads();
    }

    // merge the L slices of each bucket, weight by the bucket index, tree-reduce the workgroup
    red[tid] = acc;
    __syncthreads();
    if (walker && my_slice == 0) {
        for (unsigned l = 1; l < L; l++) g1_add(acc, red[tid + l]);
        // k * B_k, left-to-right double-and-add on the bucket index
        G1Xyzz r = g1_xyzz_identity();
        for (int bit = (int)c - 1; bit >= 0; bit--) {
            g1_dbl(r);
            if ((my_bucket >> bit) & 1) g1_add(r, acc);
        }
        acc = r;
    } else {
        acc = g1_xyzz_identity();
    }
    __syncthreads();
    red[tid] = acc;
    __syncthreads();
    for (unsigned s = MSM_BLOCK / 2; s > 0; s >>= 1) {
        if (tid < s) {
            G1Xyzz a = red[tid];
            g1_add(a, red[tid + s]);
            red[tid] = a;
        }
        __syncthreads();
    }
    if (tid == 0) partial[(size_t)m * G + g] = red[0];
}

// ------------------------------------------------------------------------------------------------
// out_xy[m] = canonical affine of sum_g partial[m][g]; flags[m] = 1 for the identity
__global__ void msm_finalize_kernel(const G1Xyzz* partial, size_t M, unsigned G, Fq* out_xy, uint8_t* flags) {
    for (size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (size_t)gridDim.x * blockDim.x) {
        G1Xyzz acc = partial[m * G];
        for (unsigned g = 1; g < G; g++) g1_add(acc, partial[m * G + g]);
        G1Affine a = g1_to_affine(acc);
        flags[m] = g1_affine_is_identity(a) ? 1 : 0;
        fp_store(out_xy + 2 * m, fp_from_mont(a.x));
        fp_store(out_xy + 2 * m + 1, fp_from_mont(a.y));
    }
}

// ------------------------------------------------------------------------------------------------
static unsigned windows_for(unsigned c) {
    // smallest W with 2^254 + K < 2^(c*W), K < 2^(c*W) * (1/2 + 2^-c): c*W >= 256 suffices
    return (256 + c - 1) / c;
}

int msm_build_table(plonk_ctx* ctx, plonk_srs* srs, unsigned c) {
    if (srs->table && srs->window_bits == c) return PLONK_OK;
    if (srs->table) {
        hipFree(srs->table);
        srs->table = nullptr;
    }
    const unsigned W = windows_for(c);
    const size_t n = srs->n_points, total = n * W;
    void *tmp = nullptr, *tab = nullptr;
    if (hipMalloc(&tmp, total * sizeof(G1Xyzz)) != hipSuccess || hipMalloc(&tab, total * sizeof(G1Affine)) != hipSuccess) {
        if (tmp) hipFree(tmp);
        plonk_set_error("hipMalloc of the %zu-point window table failed", total);
        return PLONK_ERR_NOMEM;
    }
    unsigned grid = (unsigned)((n + 63) / 64);
    if (grid > 2048) grid = 2048;
    PLONK_LAUNCH(msm_table_kernel, dim3(grid), dim3(64), 0, ctx->stream, srs->bases, n, c, W, (G1Xyzz*)tmp);
    size_t chunks = (total + AFF_CHUNK - 1) / AFF_CHUNK;
    unsigned g2 = (unsigned)((chunks + 63) / 64);
    if (g2 > 4096) g2 = 4096;
    PLONK_LAUNCH(g1_batch_to_affine_kernel, dim3(g2), dim3(64), 0, ctx->stream, (const G1Xyzz*)tmp, (G1Affine*)tab, total);
    PLONK_CHECK_HIP(hipGetLastError());
    PLONK_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    hipFree(tmp);
    srs->table = (G1Affine*)tab;
    srs->window_bits = c;
    srs->n_windows = W;
    return PLONK_OK;
}

// Enqueue a batch of M MSMs; results land in device buffers (d_out_xy: 2*M Fq canonical, d_flags: M bytes).
int msm_run_device(plonk_ctx* ctx, plonk_srs* srs, const Fr* d_scalars, size_t n, size_t M, size_t stride,
                   Fq* d_out_xy, uint8_t* d_flags) {
    PLONK_REQUIRE(n >= 1 && n <= srs->n_points, PLONK_ERR_ARG, "MSM size %zu exceeds the %zu loaded bases", n, srs->n_points);
    PLONK_REQUIRE(n <= 32768, PLONK_ERR_ARG, "MSM size %zu > 32768 is not supported by the LDS sort", n);
    if (!M) return PLONK_OK;
    unsigned c = ctx->msm_window_bits ? ctx->msm_window_bits : 8;
    if (c < 2) c = 2;
    if (c > 9) c = 9;  // K = 2^(c-1) buckets must fit one 256-lane workgroup
    PLONK_TRY(msm_build_table(ctx, srs, c));
    const unsigned W = srs->n_windows;
    unsigned G = ctx->msm_groups;
    if (!G) {
        // enough workgroups to fill 256 CUs a few times over, but no more splitting than needed
        G = 1;
        while (G < W && M * G < 1024) G *= 2;
    }
    if (G > W) G = W;

    const size_t dig_bytes = M * W * n * sizeof(uint16_t);
    const size_t part_bytes = M * G * sizeof(G1Xyzz);
    void* s;
    PLONK_TRY(ctx_scratch(ctx, 1, dig_bytes + 256 + part_bytes, &s));
    uint16_t* digits = (uint16_t*)s;
    G1Xyzz* partial = (G1Xyzz*)((uint8_t*)s + ((dig_bytes + 255) & ~(size_t)255));

    size_t total = n * M;
    unsigned gd = (unsigned)((total + 255) / 256);
    if (gd > 4096) gd = 4096;
    MsmRecode rc;
    memset(&rc, 0, sizeof rc);
    for (unsigned w = 0; w < W; w++) {
        unsigned bit = c * w + c - 1;
        rc.k[bit >> 5] |= 1u << (bit & 31);
    }
    PLONK_LAUNCH(msm_digits_kernel, dim3(gd), dim3(256), 0, ctx->stream, d_scalars, n, M, stride, c, W, rc, digits);

    const unsigned K = 1u << (c - 1);
    size_t shmem = (size_t)(3 * (K + 1)) * 4 + n * 2;
    shmem = ((shmem + 15) & ~(size_t)15) + (size_t)MSM_BLOCK * sizeof(G1Xyzz);
    PLONK_REQUIRE(shmem <= 160 * 1024, PLONK_ERR_ARG, "MSM LDS footprint %zu exceeds 160 KiB", shmem);
    static bool configured = false;
    if (!configured) {
        PLONK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(msm_accumulate_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
        configured = true;
    }
    // algorithmic bytes of an MSM of size n: (64 + 32) * n + 64   
    PLONK_TRY(prof_begin(ctx, "msm_accumulate", (double)M * (96.0 * (double)n + 64.0)));
    PLONK_LAUNCH(msm_accumulate_kernel, dim3((unsigned)(M * G)), dim3(MSM_BLOCK), shmem, ctx->stream,
                 (const G1Affine*)srs->table, srs->n_points, (const uint16_t*)digits, n, c, W, G, partial);
    PLONK_TRY(prof_end(ctx));
    unsigned gf = (unsigned)((M + 63) / 64);
    PLONK_LAUNCH(msm_finalize_kernel, dim3(gf), dim3(64), 0, ctx->stream, (const G1Xyzz*)partial, M, G, d_out_xy, d_flags);
    PLONK_CHECK_HIP(hipGetLastError());
    return PLONK_OK;
}
