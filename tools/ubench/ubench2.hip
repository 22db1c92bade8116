// ubench2.hip — round-2 microbenchmarks on gfx950 for the two alternative multiplier pipes VERDICT r01 #3 asks
// about, and for the cross-lane primitives the NTT kernels use.
//   1. the 254-bit x 254-bit PRODUCT (no reduction) three ways:
//        int29   : 9 x 9 limbs of 29 bits, 81 v_mad_u64_u32 into 64-bit column accumulators (what fpl_mul does)
//        f64     : 6 x 6 limbs of 48 bits on the FP64 pipe, round-toward-zero FMAs: per partial product one chained
//                  FMA for the high part (the accumulator is aligned so its ulp is 2^48: trunc(a*b + H) adds exactly
//                  floor(a*b / 2^48) units), one exact subtraction, one FMA for the low 48 bits, one addition —
//                  4 FP64 operations per partial product, 144 per product; exactness is checked against __int128
//   2. the constant-operand half (Montgomery's q*m) on the matrix pipe: the instruction mix a
//      v_mfma_i32_32x32x32_i8 formulation needs per 64 reductions of one wave — 4 + 16 v_permlane32_swap to lay the
//      operands out / bring the column sums home, 2 MFMAs, ~130 plain VALU operations to split 29-bit limbs into
//      signed bytes and to fold 32 int32 column sums back into limbs — against the 90 v_mad_u64_u32 it replaces
//   3. cross-lane exchange rates: DPP quad_perm / row_ror, ds_swizzle, ds_bpermute (__shfl_xor), v_permlane32_swap
// Prints one JSON object; rates are per-lane operations per second over the whole chip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define ITERS 512

// ---- 1a. integer product: 81 mads, 17 columns ---------------------------------------------------
__global__ void __launch_bounds__(256) k_prod_int29(uint32_t* out, uint32_t seed) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t a[9], b[9];
    for (int i = 0; i < 9; i++) { a[i] = (tid * 2654435761u + i * 40503u + seed) & 0x1fffffffu; b[i] = (tid * 40503u + i * 2654435761u + 7u) & 0x1fffffffu; }
    uint32_t s = 0;
    for (int it = 0; it < ITERS; it++) {
        uint64_t acc = 0;
        uint32_t r[9];
#pragma unroll
        for (int k = 0; k < 17; k++) {
#pragma unroll
            for (int i = 0; i < 9; i++) {
                const int j = k - i;
                if (j >= 0 && j < 9) acc += (uint64_t)a[i] * b[j];
            }
            if (k >= 8) r[k - 8] = (uint32_t)acc & 0x1fffffffu;
            acc >>= 29;
        }
#pragma unroll
        for (int i = 0; i < 9; i++) { a[i] = r[i]; s ^= r[i]; }
    }
    out[tid] = s;
}

// ---- 1b. FP64 product: 36 partial products x 4 DP ops --------------------------------------------
// 6 limbs of 48 bits held as doubles (exact integers < 2^48).  Column k: H runs a chain of round-toward-zero FMAs
// from the constant C = 2^100 (ulp 2^48), L sums the exact low parts (each < 2^48; at most 6 per column: < 2^51).
__device__ __forceinline__ void f64_product(const double a[6], const double b[6], double hi[11], double lo[11]) {
    const double C = 1267650600228229401496703205376.0;  // 2^100
#pragma unroll
    for (int k = 0; k < 11; k++) {
        double H = C, L = 0.0;
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const int j = k - i;
            if (j < 0 || j > 5) continue;
            const double Hn = __builtin_fma(a[i], b[j], H);   // H + floor(a b / 2^48) 2^48   (RTZ, H multiple of 2^48)
            const double t = H - Hn;                          // exact: -(hi part)
            L += __builtin_fma(a[i], b[j], t);                // exact low 48 bits
            H = Hn;
        }
        hi[k] = H - C;  // multiple of 2^48, < 6 * 2^96
        lo[k] = L;
    }
}

__global__ void __launch_bounds__(256) k_prod_f64(double* out, uint32_t seed, int check) {
    __builtin_amdgcn_s_setreg(1 | (2 << 6) | (1 << 11), 3);  // MODE.FP_ROUND[3:2] (f64/f16) = round toward zero
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    double a[6], b[6];
    for (int i = 0; i < 6; i++) {
        a[i] = (double)((((uint64_t)(tid * 2654435761u + i * 40503u + seed) << 16) ^ (tid * 97u + i)) & 0xffffffffffffull);
        b[i] = (double)((((uint64_t)(tid * 40503u + i * 2654435761u + 7u) << 16) ^ (tid * 131u + 3 * i)) & 0xffffffffffffull);
    }
    if (check) {  // one product, raw outputs for the host to verify against __int128
        double hi[11], lo[11];
        f64_product(a, b, hi, lo);
        for (int k = 0; k < 11; k++) { out[tid * 34 + k] = hi[k]; out[tid * 34 + 11 + k] = lo[k]; }
        for (int i = 0; i < 6; i++) { out[tid * 34 + 22 + i] = a[i]; out[tid * 34 + 28 + i] = b[i]; }
        return;
    }
    double s = 0;
    for (int it = 0; it < ITERS; it++) {
        double hi[11], lo[11];
        f64_product(a, b, hi, lo);
#pragma unroll
        for (int i = 0; i < 6; i++) {  // feed something product-dependent back (keeps the chain honest, stays < 2^48)
            a[i] = lo[i] * 0.125;
            a[i] = a[i] - (double)(long long)(a[i] * (1.0 / 281474976710656.0)) * 281474976710656.0;
            s += hi[i + 5] * 1e-40;
        }
    }
    out[tid] = s + a[0];
}

// ---- 2. matrix-pipe reduction: instruction mix -------------------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// mode 0: 90 mads (the q*m half of fpl_mul);  mode 1: the MFMA formulation's mix;  mode 2: mode 1 without the MFMAs
// (what the VALU alone pays);  mode 3: MFMAs only
template <int MODE>
__global__ void __launch_bounds__(256) k_reduce_mix(uint32_t* out, uint32_t seed) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t x[9];
    for (int i = 0; i < 9; i++) x[i] = (tid * 2654435761u + i * 40503u + seed) & 0x1fffffffu;
    uint32_t s = 0;
    v16i acc0 = {0}, acc1 = {0};
    for (int it = 0; it < ITERS; it++) {
        if (MODE == 0) {
            uint64_t acc = x[0];
            uint32_t q[9];
#pragma unroll
            for (int k = 0; k < 9; k++) {  // 9 mul_lo + 81 mads, shaped like the reduction half of fpl_mul
#pragma unroll
                for (int i = 0; i < k; i++) acc += (uint64_t)q[i] * (0x12345u + 77u * (k - i));
                q[k] = ((uint32_t)acc * 0x0fffffffu) & 0x1fffffffu;
                acc += (uint64_t)q[k] * 0x10000001u;
                acc >>= 29;
            }
#pragma unroll
            for (int k = 9; k < 17; k++) {
#pragma unroll
                for (int i = k - 8; i < 9; i++) acc += (uint64_t)q[i] * (0x54321u + 31u * (k - i));
                x[k - 9] = ((uint32_t)acc & 0x1fffffffu) ^ x[k - 9];
                acc >>= 29;
            }
            x[8] ^= (uint32_t)acc;
        } else {
            // (a) 9 x 29-bit limbs -> 33 bytes, recoded to signed digits: ~24 shifts/ors + 8 add-with-carry + 8 xor
            uint32_t w[9];
            if (MODE != 3) {
#pragma unroll
                for (int i = 0; i < 8; i++) w[i] = (x[i] >> (3 * i)) | (x[i + 1] << (29 - 3 * i));
                w[8] = x[8] >> 24;
                uint32_t c = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) { uint64_t t = (uint64_t)w[i] + 0x80808080u + c; w[i] = (uint32_t)t ^ 0x80808080u; c = (uint32_t)(t >> 32); }
                w[8] += c;
            } else {
#pragma unroll
                for (int i = 0; i < 9; i++) w[i] = x[i];
            }
            // (b) operands to the MFMA layout: lanes 0-31 <-> 32-63 exchange of 4 dwords
            if (MODE != 3) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    auto r = __builtin_amdgcn_permlane32_swap(w[i], w[4 + i], false, false);
                    w[i] = r[0];
                    w[4 + i] = r[1];
                }
            }
            // (c) two 32x32x32 i8 MFMAs (K = 32 digits, N = 32 elements each)
            if (MODE != 2) {
                v4i a0 = {(int)w[0], (int)w[1], (int)w[2], (int)w[3]}, a1 = {(int)w[4], (int)w[5], (int)w[6], (int)w[7]};
                v4i bm = {0x01020304, 0x05060708, 0x090a0b0c, 0x0d0e0f10};
                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(bm, a0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(bm, a1, acc1, 0, 0, 0);
            }
            if (MODE != 3) {
                // (d) column sums home: 16 half-exchanges
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    auto r = __builtin_amdgcn_permlane32_swap((uint32_t)acc0[i], (uint32_t)acc1[i], false, false);
                    acc0[i] = (int)r[0];
                    acc1[i] = (int)r[1];
                }
                // (e) 32 int32 column sums (|.| < 2^20) -> 9 limbs: per column sign-split + shift-add into its limb (~4 ops)
                uint32_t l[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int c8 = 0; c8 < 32; c8++) {
                    const int v = c8 < 16 ? acc0[c8] : acc1[c8 - 16];
                    const int bit = 8 * c8, li = bit / 29, sh = bit % 29;
                    const uint32_t lo = ((uint32_t)v << sh) & 0x1fffffffu;
                    const int hi = v >> (29 - sh);
                    l[li] += lo;
                    if (li + 1 < 9) l[li + 1] += (uint32_t)hi;
                }
                uint32_t c = 0;
#pragma unroll
                for (int i = 0; i < 9; i++) { uint32_t t = l[i] + c; x[i] = (t & 0x1fffffffu) ^ (x[i] >> 1); c = (uint32_t)((int)t >> 29); }
                // keep the accumulators small and data-dependent
#pragma unroll
                for (int i = 0; i < 16; i++) { acc0[i] &= 0xfffff; acc1[i] &= 0xfffff; }
            }
        }
    }
    for (int i = 0; i < 9; i++) s ^= x[i];
    for (int i = 0; i < 16; i++) s ^= (uint32_t)(acc0[i] ^ acc1[i]);
    out[tid] = s;
}

// ---- 3. cross-lane exchange rates (one dword per lane per op) ---------------------------------------------
template <int OP>
__global__ void __launch_bounds__(256) k_xlane(uint32_t* out, uint32_t seed) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t v[8];
    for (int i = 0; i < 8; i++) v[i] = tid * 2654435761u + i + seed;
    for (int it = 0; it < ITERS * 4; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (OP == 0) v[i] += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v[i], 0xB1, 0xf, 0xf, false);    // quad_perm [1,0,3,2]: lane ^ 1
            if (OP == 1) v[i] += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v[i], 0x128, 0xf, 0xf, false);   // row_ror:8: lane ^ 8
            if (OP == 2) v[i] += (uint32_t)__builtin_amdgcn_ds_swizzle((int)v[i], (4 << 10) | 0x1f);            // xor 4 within 32
            if (OP == 3) v[i] += (uint32_t)__shfl_xor((int)v[i], 16);                                           // ds_bpermute
            if (OP == 4) { auto r = __builtin_amdgcn_permlane32_swap(v[i], v[(i + 1) & 7], false, false); v[i] += r[0]; v[(i + 1) & 7] ^= r[1]; }
            if (OP == 5) v[i] += v[(i + 1) & 7] ^ 0x9e3779b9u;                                                  // plain VALU reference (2 ops)
        }
    }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= v[i];
    out[tid] = s;
}

template <class F> static double time_ms(F launch, int reps = 5) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * 8, threads = 256;
    const double lanes = (double)blocks * threads;
    uint32_t* out; hipMalloc(&out, (size_t)blocks * threads * 34 * 8);
    printf("{\"device\": \"%s\", \"cus\": %d", prop.gcnArchName, prop.multiProcessorCount);

    // exactness of the FP64 product first
    {
        const int nchk = 256;
        hipLaunchKernelGGL(k_prod_f64, dim3(1), dim3(nchk), 0, 0, (double*)out, 5u, 1);
        static double h[256 * 34];
        hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
        int bad = 0, neg_lo = 0;
        for (int t = 0; t < nchk; t++) {
            const double* r = h + t * 34;
            for (int k = 0; k < 11; k++) {
                __int128 want = 0;
                for (int i = 0; i < 6; i++) { int j = k - i; if (j >= 0 && j < 6) want += (__int128)(uint64_t)r[22 + i] * (uint64_t)r[28 + j]; }
                // r[k] is a multiple of 2^48 below 2^99: r[k] / 2^48 is an exact integer below 2^51; the low sum is an
                // exact integer too (non-negative when the FMAs really truncate)
                __int128 got = (__int128)(int64_t)(r[k] / 281474976710656.0) * ((__int128)1 << 48) + (__int128)(int64_t)r[11 + k];
                if (got != want) bad++;
                if (r[11 + k] < 0) neg_lo++;
            }
        }
        printf(", \"f64_product_exact\": %s, \"f64_round_toward_zero_in_effect\": %s", bad ? "false" : "true", neg_lo ? "false" : "true");
    }
    double t_int = time_ms([&] { hipLaunchKernelGGL(k_prod_int29, dim3(blocks), dim3(threads), 0, 0, out, 1u); });
    double t_f64 = time_ms([&] { hipLaunchKernelGGL(k_prod_f64, dim3(blocks), dim3(threads), 0, 0, (double*)out, 1u, 0); });
    printf(", \"product_int29_81mad_Gops\": %.2f, \"product_f64_144op_Gops\": %.2f", lanes * ITERS / (t_int * 1e-3) / 1e9,
           lanes * ITERS / (t_f64 * 1e-3) / 1e9);
    double r0 = time_ms([&] { hipLaunchKernelGGL(k_reduce_mix<0>, dim3(blocks), dim3(threads), 0, 0, out, 1u); });
    double r1 = time_ms([&] { hipLaunchKernelGGL(k_reduce_mix<1>, dim3(blocks), dim3(threads), 0, 0, out, 1u); });
    double r2 = time_ms([&] { hipLaunchKernelGGL(k_reduce_mix<2>, dim3(blocks), dim3(threads), 0, 0, out, 1u); });
    double r3 = time_ms([&] { hipLaunchKernelGGL(k_reduce_mix<3>, dim3(blocks), dim3(threads), 0, 0, out, 1u); });
    printf(", \"reduce_90mad_Gops\": %.2f, \"reduce_mfma_mix_Gops\": %.2f, \"reduce_mfma_mix_valu_only_Gops\": %.2f, \"reduce_mfma_only_Gops\": %.2f",
           lanes * ITERS / (r0 * 1e-3) / 1e9, lanes * ITERS / (r1 * 1e-3) / 1e9, lanes * ITERS / (r2 * 1e-3) / 1e9, lanes * ITERS / (r3 * 1e-3) / 1e9);
    const char* names[6] = {"dpp_quad_perm", "dpp_row_ror8", "ds_swizzle", "ds_bpermute", "permlane32_swap", "valu_add_xor"};
    double x[6];
    x[0] = time_ms([&] { hipLaunchKernelGGL(k_xlane<0>, dim3(blocks), dim3(threads), 0, 0, out, 1u); });
    x[1] = time_ms([&] { hipLaunchKernelGGL(k_xlane<1>, dim3(blocks), dim3(threads), 0, 0, out, 1u); });
    x[2] = time_ms([&] { hipLaunchKernelGGL(k_xlane<2>, dim3(blocks), dim3(threads), 0, 0, out, 1u); });
    x[3] = time_ms([&] { hipLaunchKernelGGL(k_xlane<3>, dim3(blocks), dim3(threads), 0, 0, out, 1u); });
    x[4] = time_ms([&] { hipLaunchKernelGGL(k_xlane<4>, dim3(blocks), dim3(threads), 0, 0, out, 1u); });
    x[5] = time_ms([&] { hipLaunchKernelGGL(k_xlane<5>, dim3(blocks), dim3(threads), 0, 0, out, 1u); });
    for (int i = 0; i < 6; i++) printf(", \"xlane_%s_Gops\": %.1f", names[i], lanes * ITERS * 4 * 8 / (x[i] * 1e-3) / 1e9);
    printf("}\n");
    hipFree(out);
    return 0;
}
