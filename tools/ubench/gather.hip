// gather.hip — random 64-byte reads over a table of T GiB (T = argv[1], default 64): the access pattern of a
// direct-lookup fixed-base MSM (one affine G1 point per lane per step, no locality).  Prints JSON with
// G reads/s and useful GB/s for several table sizes.  Build: hipcc --offload-arch=gfx950 -O3 gather.hip -o gather.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

struct alignas(16) u4 { uint32_t x, y, z, w; };

__global__ void __launch_bounds__(256) k_gather(const u4* table, uint64_t n_slots, int iters, uint32_t seed, uint32_t* out) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t s = (uint64_t)tid * 0x9e3779b97f4a7c15ull + seed;
    uint32_t acc = 0;
    for (int i = 0; i < iters; i += 4) {
        u4 v[8];
#pragma unroll
        for (int k = 0; k < 4; k++) {  // four independent 64-byte reads in flight per lane
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            const uint64_t slot = (s >> 20) % n_slots;
            const u4* p = table + slot * 4;
            v[2 * k] = p[0];
            v[2 * k + 1] = p[1];   // first 32 bytes only would be x; read x and half of y: 2 x 16 B ... plus
            acc ^= p[2].x ^ p[3].w; // ... the rest of the 64-byte slot
        }
#pragma unroll
        for (int k = 0; k < 8; k++) acc ^= v[k].x + v[k].w;
    }
    out[tid] = acc;
}

int main(int argc, char** argv) {
    const double max_gib = argc > 1 ? atof(argv[1]) : 64.0;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * 16, threads = 256, iters = 256;
    uint32_t* out;
    hipMalloc(&out, (size_t)blocks * threads * 4);
    printf("{\"cus\": %d", prop.multiProcessorCount);
    for (double gib : {0.0039, 0.25, 2.0, 8.0, 32.0, 64.0, 128.0, 160.0}) {
        if (gib > max_gib) break;
        const size_t bytes = (size_t)(gib * 1024.0 * 1024.0 * 1024.0) & ~(size_t)63;
        void* t = nullptr;
        if (hipMalloc(&t, bytes) != hipSuccess) { printf(", \"alloc_failed_gib\": %.2f", gib); break; }
        hipMemset(t, 1, bytes);
        hipDeviceSynchronize();
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e30f;
        for (int r = 0; r < 3; r++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(threads), 0, 0, (const u4*)t, (uint64_t)(bytes / 64), iters, (uint32_t)r, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double reads = (double)blocks * threads * iters;
        printf(", \"%.4g_GiB\": {\"Greads_per_s\": %.2f, \"useful_GBps\": %.1f}", gib, reads / (best * 1e-3) / 1e9, reads * 64 / (best * 1e-3) / 1e9);
        fflush(stdout);
        hipFree(t);
    }
    printf("}\n");
    return 0;
}
