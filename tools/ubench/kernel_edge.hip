// kernel_edge.hip — what one ROUND of workgroups costs besides its arithmetic (round 6: why a lone 2^20 transform, two launches of
// 1024 workgroups each, keeps its SIMDs busy only 60 % of its duration).  One launch = 1024 workgroups of 256 threads, four
// 32-byte elements per thread (32 MiB in, 32 MiB out: a pass of a 2^20-point transform), averaged over back-to-back launches:
//   empty      nothing                                       -> dispatch + completion
//   store      4 x 32 B stores per thread, contiguous        -> + write drain / end-of-kernel write-back
//   store_nt   the same with non-temporal stores
//   copy       load 4, store 4 (contiguous)
//   gather     load 4 with a stride of 32 KiB (a column of a 1024 x 1024 matrix), store contiguous
//   gather_x   the same with the transform kernels' XCD-aware column order (four adjacent columns per XCD share 128-byte lines)
//   scatter_x  load contiguous, store 4 with a stride of 32 KiB, XCD-aware (the row pass's natural-order output)
//   gather_scatter_x  both strided (the row pass as it is: its input is the transposed intermediate)
//   spin       ~30 us of dependent FMAs, no memory           -> the arithmetic stand-in
//   copy_spin  load 4, spin, store 4                         -> what overlaps and what does not
// Prints one JSON line of microseconds per launch.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ float spin(float v, int iters) {
    for (int i = 0; i < iters; i++) v = __builtin_fmaf(v, 1.0000001f, 0.5f);
    return v;
}
template <int MODE> __global__ void __launch_bounds__(256) k(const u32x4* in, u32x4* out, int iters) {
    const unsigned t = blockIdx.x * 256 + threadIdx.x;  // element index base: 4 elements of 2 x u32x4 each
    u32x4 a[8];
    if (MODE == 0) return;
    if (MODE == 3 || MODE == 6 || MODE == 8) {
#pragma unroll
        for (int j = 0; j < 4; j++) { a[2 * j] = in[2 * (j * 262144 + t)]; a[2 * j + 1] = in[2 * (j * 262144 + t) + 1]; }
    } else if (MODE == 4 || MODE == 7 || MODE == 9) {  // column blockIdx.x of a 1024 x 1024 matrix: element (j * 256 + threadIdx.x, column)
        const unsigned b = blockIdx.x, col = MODE == 4 ? b : ((b & ~31u) | ((b & 7u) << 2) | ((b >> 3) & 3u));
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned g = (j * 256 + threadIdx.x) * 1024 + col;
            a[2 * j] = in[2 * g]; a[2 * j + 1] = in[2 * g + 1];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) a[j] = u32x4{t, (unsigned)j, 3u, 4u};
    }
    if (MODE == 5 || MODE == 6) {
        float v = spin((float)a[0].x, iters);
        a[0].x = (unsigned)v;
        if (MODE == 5) { if (v == 12345.0f) out[t] = a[0]; return; }
    }
    if (MODE == 8 || MODE == 9) {
        const unsigned b = blockIdx.x, col = (b & ~31u) | ((b & 7u) << 2) | ((b >> 3) & 3u);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            u32x4* o = out + 2 * ((j * 256 + threadIdx.x) * 1024 + col);
            o[0] = a[2 * j]; o[1] = a[2 * j + 1];
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        u32x4* o = out + 2 * (j * 262144 + t);
        if (MODE == 2) { __builtin_nontemporal_store(a[2 * j], o); __builtin_nontemporal_store(a[2 * j + 1], o + 1); }
        else { o[0] = a[2 * j]; o[1] = a[2 * j + 1]; }
    }
}
template <int MODE> static double run(const u32x4* in, u32x4* out, int iters, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(256), 0, 0, in, out, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(256), 0, 0, in, out, iters);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return 1e3 * ms / reps;
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 15000, reps = 50;
    u32x4 *in, *out;
    CHECK(hipMalloc(&in, 32u << 20)); CHECK(hipMalloc(&out, 32u << 20));
    CHECK(hipMemset(in, 1, 32u << 20));
    printf("{\"what\": \"one round of 1024 workgroups x 256 threads, 4 x 32 B per thread; us per launch, back to back\", \"spin_iters\": %d, ", iters);
    printf("\"empty\": %.2f, ", run<0>(in, out, iters, reps));
    printf("\"store\": %.2f, ", run<1>(in, out, iters, reps));
    printf("\"store_nt\": %.2f, ", run<2>(in, out, iters, reps));
    printf("\"copy\": %.2f, ", run<3>(in, out, iters, reps));
    printf("\"gather\": %.2f, ", run<4>(in, out, iters, reps));
    printf("\"gather_x\": %.2f, ", run<7>(in, out, iters, reps));
    printf("\"scatter_x\": %.2f, ", run<8>(in, out, iters, reps));
    printf("\"gather_scatter_x\": %.2f, ", run<9>(in, out, iters, reps));
    printf("\"spin\": %.2f, ", run<5>(in, out, iters, reps));
    printf("\"copy_spin\": %.2f}\n", run<6>(in, out, iters, reps));
    return 0;
}
