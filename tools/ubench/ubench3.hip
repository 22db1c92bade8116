// ubench3.hip — rates of the non-multiplier instructions of the 29-bit-limb multiplication on gfx950: the 64-bit right
// shift (v_lshrrev_b64) and 64-bit add (v_lshl_add_u64) the compiler emits for `acc >>= 29` / partial-sum merges, against
// their 32-bit replacements (v_alignbit_b32 + v_lshrrev_b32; v_add_co + v_addc_co).  Prints G lane-ops/s, whole chip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define ITERS 4096
#define CH 8
template <int OP> __global__ void __launch_bounds__(256) k(uint32_t* out, uint32_t seed) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t a[CH];
    for (int c = 0; c < CH; c++) a[c] = ((uint64_t)(tid * 2654435761u + c + seed) << 32) | (tid * 40503u + c * 977u);
    const uint64_t add = ((uint64_t)seed << 33) | 0x9e3779b9u;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
            if (OP == 0) { uint64_t t; asm volatile("v_lshrrev_b64 %0, 29, %1" : "=v"(t) : "v"(a[c])); a[c] = t ^ add; }
            if (OP == 1) { uint32_t lo = (uint32_t)a[c], hi = (uint32_t)(a[c] >> 32), nl, nh;
                           asm volatile("v_alignbit_b32 %0, %1, %2, 29" : "=v"(nl) : "v"(hi), "v"(lo));
                           asm volatile("v_lshrrev_b32 %0, 29, %1" : "=v"(nh) : "v"(hi));
                           a[c] = (((uint64_t)nh << 32) | nl) ^ add; }
            if (OP == 2) { uint64_t t; asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(t) : "v"(a[c]), "v"(add)); a[c] = t; }
            if (OP == 3) { uint64_t t = a[c] ^ add; a[c] = t; }  // the xor alone (2 x v_xor_b32): subtract from 0 / 1
            if (OP == 5) { uint64_t t; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(t) : "v"((uint32_t)a[c]), "v"((uint32_t)add), "v"(a[c]) : "vcc"); a[c] = t; }
            if (OP == 6) { uint64_t t; asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %3" : "=v"(t) : "v"((uint32_t)a[c]), "v"((uint32_t)add), "v"(a[c]) : "vcc"); a[c] = t; }
            if (OP == 7) { uint64_t t; asm volatile("v_ashrrev_i64 %0, 29, %1" : "=v"(t) : "v"(a[c])); a[c] = t + add; }
            if (OP == 4) { uint32_t lo = (uint32_t)a[c], hi = (uint32_t)(a[c] >> 32), nl, nh;
                           asm volatile("v_add_co_u32 %0, vcc, %2, %3\n\tv_addc_co_u32 %1, vcc, %4, %5, vcc" : "=&v"(nl), "=v"(nh) : "v"(lo), "v"((uint32_t)add), "v"(hi), "v"((uint32_t)(add >> 32)) : "vcc");
                           a[c] = ((uint64_t)nh << 32) | nl; }
        }
    }
    uint64_t s = 0;
    for (int c = 0; c < CH; c++) s += a[c];
    out[tid] = (uint32_t)s ^ (uint32_t)(s >> 32);
}
template <class F> static double time_ms(F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; r++) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    return best;
}
int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * 8, threads = 256;
    uint32_t* out; hipMalloc(&out, (size_t)blocks * threads * 4);
    const double lanes = (double)blocks * threads;
    const char* names[8] = {"lshrrev_b64_plus_xor", "alignbit_lshr32_plus_xor", "lshl_add_u64", "xor64_only", "add_co_addc_co", "mad_u64_u32", "mad_i64_i32", "ashrrev_i64_plus_add64"};
    double ms[8];
    ms[0] = time_ms([&] { hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, out, 1u); });
    ms[1] = time_ms([&] { hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, out, 1u); });
    ms[2] = time_ms([&] { hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(threads), 0, 0, out, 1u); });
    ms[3] = time_ms([&] { hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(threads), 0, 0, out, 1u); });
    ms[4] = time_ms([&] { hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(threads), 0, 0, out, 1u); });
    ms[5] = time_ms([&] { hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(threads), 0, 0, out, 1u); });
    ms[6] = time_ms([&] { hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(threads), 0, 0, out, 1u); });
    ms[7] = time_ms([&] { hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(threads), 0, 0, out, 1u); });
    printf("{");
    for (int i = 0; i < 8; i++) printf("%s\"%s_Gops\": %.1f", i ? ", " : "", names[i], lanes * ITERS * CH / (ms[i] * 1e-3) / 1e9);
    printf("}\n");
    return 0;
}
