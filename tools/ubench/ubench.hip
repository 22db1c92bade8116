// ubench.hip — instruction-rate microbenchmarks on gfx950 for the integer / fp64 primitives a
// 254-bit Montgomery multiplication can be built from, plus the library's own fp_mul.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../plonkathon_amd/csrc ubench.hip -o ubench.bin
// Prints one JSON object; rates are per-lane operations per second over the whole chip.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "fp.h"
#include "g1.h"

#define ITERS 2048
#define CHAINS 8

template <int OP>
__global__ void __launch_bounds__(256) k_op(uint32_t* out, uint32_t seed) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t a[CHAINS];
    uint32_t x = tid * 2654435761u + seed, y = x ^ 0x9e3779b9u;
    double d[CHAINS];
    for (int c = 0; c < CHAINS; c++) { a[c] = ((uint64_t)(x + c) << 32) | (y * (c + 3)); d[c] = 1.0 + c * 1e-3 + (tid & 7) * 1e-6; }
    double m1 = 1.0000001 + (seed & 1) * 1e-9, m2 = 1e-9;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            if (OP == 0) a[c] = (uint64_t)(uint32_t)a[c] * y + a[c];                       // v_mad_u64_u32
            if (OP == 1) a[c] = (uint32_t)a[c] * y + x;                                     // v_mul_lo_u32 (+add)
            if (OP == 2) a[c] = __umulhi((uint32_t)a[c], y) + x;                            // v_mul_hi_u32 (+add)
            if (OP == 3) a[c] = a[c] + (((uint64_t)x << 32) | y);                           // 64-bit add
            if (OP == 4) a[c] = __umul24((uint32_t)a[c], y) + x;                            // v_mul_u32_u24 / mad_u32_u24
            if (OP == 5) d[c] = fma(d[c], m1, m2);                                          // v_fma_f64
            if (OP == 6) a[c] = (uint32_t)a[c] + y + ((uint32_t)a[c] < x);                  // 32-bit add chain
        }
    }
    uint64_t s = 0;
    for (int c = 0; c < CHAINS; c++) s += a[c] + (uint64_t)d[c];
    out[tid] = (uint32_t)s ^ (uint32_t)(s >> 32);
}

template <class P>
__global__ void __launch_bounds__(256) k_fpmul(uint32_t* out, uint32_t seed, int iters) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fp<P> a, b;
    for (int i = 0; i < 8; i++) { a.v[i] = tid * 2654435761u + i * 40503u + seed; b.v[i] = tid * 40503u + i * 2654435761u + 7u; }
    a.v[7] &= 0x1fffffffu; b.v[7] &= 0x1fffffffu;
    for (int i = 0; i < iters; i++) { a = fp_mul(a, b); b = fp_mul(b, a); }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= a.v[i] ^ b.v[i];
    out[tid] = s;
}

template <class P>
__global__ void __launch_bounds__(256) k_fpadd(uint32_t* out, uint32_t seed, int iters) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fp<P> a, b;
    for (int i = 0; i < 8; i++) { a.v[i] = tid * 2654435761u + i * 40503u + seed; b.v[i] = tid * 40503u + i * 2654435761u + 7u; }
    a.v[7] &= 0x1fffffffu; b.v[7] &= 0x1fffffffu;
    for (int i = 0; i < iters; i++) { a = fp_add(a, b); b = fp_sub(b, a); }
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s ^= a.v[i] ^ b.v[i];
    out[tid] = s;
}

// the MSM inner loop's own units: lazy 9x29-bit multiplication and the lazy mixed addition, register-resident operands
__global__ void __launch_bounds__(256) k_fplmul(uint32_t* out, uint32_t seed, int iters) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    FqL a, b;
    for (int i = 0; i < 9; i++) { a.l[i] = (tid * 2654435761u + i * 40503u + seed) & FP29_MASK; b.l[i] = (tid * 40503u + i * 2654435761u + 7u) & FP29_MASK; }
    a.l[8] &= 0xfffff; b.l[8] &= 0xfffff;
    for (int i = 0; i < iters; i++) { a = fpl_mul(a, b); b = fpl_mul(b, a); }
    uint32_t s = 0;
    for (int i = 0; i < 9; i++) s ^= a.l[i] ^ b.l[i];
    out[tid] = s;
}
// the NTT's unit: multiplication by a constant held as a Shoup pair (w, floor(w 2^261 / r)), operand and result on lazy limbs
__global__ void __launch_bounds__(256) k_fplshoup(uint32_t* out, uint32_t seed, int iters) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    FpL<FrParams> a, b;
    FpLS<FrParams> c, d;
    for (int i = 0; i < 9; i++) {
        a.l[i] = (tid * 2654435761u + i * 40503u + seed) & FP29_MASK;
        b.l[i] = (tid * 40503u + i * 2654435761u + 7u) & FP29_MASK;
        c.w[i] = (int32_t)((tid * 97u + i * 7919u + seed) & FP29_MASK);
        c.wp[i] = (int32_t)((tid * 31u + i * 104729u + 3u) & FP29_MASK);
        d.w[i] = c.wp[i] ^ 0x5555;
        d.wp[i] = c.w[i] ^ 0x3333;
    }
    a.l[8] &= 0xfffff; b.l[8] &= 0xfffff; c.w[8] &= 0xfffff; d.w[8] &= 0xfffff;
    for (int i = 0; i < iters; i++) { a = fpl_mul_shoup(a, c); b = fpl_mul_shoup(b, d); }
    uint32_t s = 0;
    for (int i = 0; i < 9; i++) s ^= a.l[i] ^ b.l[i];
    out[tid] = s;
}
__global__ void __launch_bounds__(256, 4) k_g1lmadd(uint32_t* out, uint32_t seed, int iters) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    // not curve points: the formulas do not care, and the exceptional-case filter almost never fires on random data
    Fq x, y;
    for (int i = 0; i < 8; i++) { x.v[i] = tid * 2654435761u + i * 40503u + seed; y.v[i] = tid * 40503u + i * 2654435761u + 7u; }
    x.v[7] &= 0x0fffffffu; y.v[7] &= 0x0fffffffu;
    G1XyzzL acc = g1l_identity();
    uint32_t skipped = 0;
    for (int i = 0; i < iters; i++) {
        skipped += !g1l_madd_fast(acc, x, y);
        x.v[0] += 0x9e3779b9u;  // a different base every step
        y.v[1] ^= x.v[0];
    }
    uint32_t s = skipped;
    for (int i = 0; i < 9; i++) s ^= acc.x.l[i] ^ acc.y.l[i] ^ acc.zz.l[i] ^ acc.zzz.l[i];
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
    uint32_t* out; hipMalloc(&out, (size_t)prop.multiProcessorCount * 32 * threads * 4);
    const double lanes = (double)blocks * threads;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d", prop.name, prop.multiProcessorCount, prop.clockRate / 1000);
    const char* names[7] = {"mad_u64_u32", "mul_lo_u32", "mul_hi_u32", "add_u64", "mul_u32_u24", "fma_f64", "add_u32_carry"};
    double ms[7];
    ms[0] = time_ms([&] { hipLaunchKernelGGL(k_op<0>, dim3(blocks), dim3(threads), 0, 0, out, 1u); });
    ms[1] = time_ms([&] { hipLaunchKernelGGL(k_op<1>, dim3(blocks), dim3(threads), 0, 0, out, 1u); });
    ms[2] = time_ms([&] { hipLaunchKernelGGL(k_op<2>, dim3(blocks), dim3(threads), 0, 0, out, 1u); });
    ms[3] = time_ms([&] { hipLaunchKernelGGL(k_op<3>, dim3(blocks), dim3(threads), 0, 0, out, 1u); });
    ms[4] = time_ms([&] { hipLaunchKernelGGL(k_op<4>, dim3(blocks), dim3(threads), 0, 0, out, 1u); });
    ms[5] = time_ms([&] { hipLaunchKernelGGL(k_op<5>, dim3(blocks), dim3(threads), 0, 0, out, 1u); });
    ms[6] = time_ms([&] { hipLaunchKernelGGL(k_op<6>, dim3(blocks), dim3(threads), 0, 0, out, 1u); });
    for (int i = 0; i < 7; i++)
        printf(", \"%s_Gops\": %.1f", names[i], lanes * ITERS * CHAINS / (ms[i] * 1e-3) / 1e9);
    const int it = 256;
    double m1 = time_ms([&] { hipLaunchKernelGGL(k_fpmul<FrParams>, dim3(blocks), dim3(threads), 0, 0, out, 1u, it); });
    double m2 = time_ms([&] { hipLaunchKernelGGL(k_fpmul<FqParams>, dim3(blocks), dim3(threads), 0, 0, out, 1u, it); });
    double m3 = time_ms([&] { hipLaunchKernelGGL(k_fpadd<FrParams>, dim3(blocks), dim3(threads), 0, 0, out, 1u, it * 8); });
    printf(", \"fr_mul_Gops\": %.2f, \"fq_mul_Gops\": %.2f, \"fr_addsub_Gops\": %.2f", lanes * it * 2 / (m1 * 1e-3) / 1e9,
           lanes * it * 2 / (m2 * 1e-3) / 1e9, lanes * it * 8 * 2 / (m3 * 1e-3) / 1e9);
    double m4 = time_ms([&] { hipLaunchKernelGGL(k_fplmul, dim3(blocks), dim3(threads), 0, 0, out, 1u, it); });
    double m5 = time_ms([&] { hipLaunchKernelGGL(k_g1lmadd, dim3(blocks), dim3(threads), 0, 0, out, 1u, it); });
    printf(", \"fq_lazy_mul_Gops\": %.2f, \"g1_lazy_madd_Gops\": %.3f", lanes * it * 2 / (m4 * 1e-3) / 1e9, lanes * it / (m5 * 1e-3) / 1e9);
    double m6 = time_ms([&] { hipLaunchKernelGGL(k_fplshoup, dim3(blocks), dim3(threads), 0, 0, out, 1u, it); });
    printf(", \"fr_shoup_mul_Gops\": %.2f", lanes * it * 2 / (m6 * 1e-3) / 1e9);
    // occupancy sweep for fr_mul: 1,2,4 blocks per CU
    for (int bpc : {1, 2, 4, 8, 16, 32}) {
        int bl = prop.multiProcessorCount * bpc;
        double m = time_ms([&] { hipLaunchKernelGGL(k_fpmul<FrParams>, dim3(bl), dim3(threads), 0, 0, out, 1u, it); });
        printf(", \"fr_mul_Gops_%dbpc\": %.2f", bpc, (double)bl * threads * it * 2 / (m * 1e-3) / 1e9);
    }
    printf("}\n");
    hipFree(out);
    return 0;
}
