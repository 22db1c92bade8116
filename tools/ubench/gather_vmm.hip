// gather_vmm.hip — the random 64-byte read rate of gather.hip over tables allocated three ways: hipMalloc, the virtual-memory API
// with the recommended granularity, and the same with a 1 GiB-aligned address range (does a larger alignment buy larger
// page-table fragments, i.e. more translation reach?).  Prints one JSON object.  Build: hipcc --offload-arch=gfx950 -O3
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
        for (int k = 0; k < 4; k++) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            const uint64_t slot = (s >> 20) % n_slots;
            const u4* p = table + slot * 4;
            v[2 * k] = p[0];
            v[2 * k + 1] = p[1];
            acc ^= p[2].x ^ p[3].w;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) acc ^= v[k].x + v[k].w;
    }
    out[tid] = acc;
}

static double rate(const void* t, size_t bytes, uint32_t* out, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    const int iters = 256;
    for (int r = 0; r < 3; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, (const u4*)t, (uint64_t)(bytes / 64), iters, (uint32_t)r, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return (double)blocks * 256 * iters / (best * 1e-3) / 1e9;
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * 16;
    uint32_t* out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipMemAllocationProp ap = {};
    ap.type = hipMemAllocationTypePinned;
    ap.location.type = hipMemLocationTypeDevice;
    ap.location.id = 0;
    size_t gmin = 0, grec = 0;
    hipMemGetAllocationGranularity(&gmin, &ap, hipMemAllocationGranularityMinimum);
    hipMemGetAllocationGranularity(&grec, &ap, hipMemAllocationGranularityRecommended);
    printf("{\"granularity_min\": %zu, \"granularity_recommended\": %zu", gmin, grec);
    for (double gib : {8.0, 64.0}) {
        const size_t bytes = (size_t)(gib * 1024.0 * 1024.0 * 1024.0);
        void* t = nullptr;
        if (hipMalloc(&t, bytes) == hipSuccess) {
            hipMemset(t, 1, bytes);
            hipDeviceSynchronize();
            printf(", \"hipMalloc_%g_GiB\": {\"Greads_per_s\": %.2f, \"address_mod_1GiB\": %zu}", gib, rate(t, bytes, out, blocks), (size_t)t & ((1ull << 30) - 1));
            fflush(stdout);
            hipFree(t);
        }
        for (size_t align : {(size_t)0, (size_t)1 << 30}) {
            void* va = nullptr;
            hipMemGenericAllocationHandle_t h;
            if (hipMemAddressReserve(&va, bytes, align, nullptr, 0) != hipSuccess) { printf(", \"reserve_failed\": %zu", align); continue; }
            if (hipMemCreate(&h, bytes, &ap, 0) != hipSuccess) { printf(", \"create_failed\": %g", gib); hipMemAddressFree(va, bytes); continue; }
            hipMemAccessDesc ad = {};
            ad.location = ap.location;
            ad.flags = hipMemAccessFlagsProtReadWrite;
            if (hipMemMap(va, bytes, 0, h, 0) != hipSuccess || hipMemSetAccess(va, bytes, &ad, 1) != hipSuccess) { printf(", \"map_failed\": %g", gib); continue; }
            hipMemset(va, 1, bytes);
            hipDeviceSynchronize();
            printf(", \"vmm_align_%zu_%g_GiB\": {\"Greads_per_s\": %.2f, \"address_mod_1GiB\": %zu}", align, gib, rate(va, bytes, out, blocks), (size_t)va & ((1ull << 30) - 1));
            fflush(stdout);
            hipMemUnmap(va, bytes);
            hipMemRelease(h);
            hipMemAddressFree(va, bytes);
        }
    }
    printf("}\n");
    return 0;
}
