// hip_emu.h — TEST INFRASTRUCTURE ONLY.
// A minimal single-threaded emulation of the HIP execution model (grid of workgroups, threads as
// cooperative fibers, __syncthreads, static/dynamic __shared__, wave64 shuffles/ballot, atomics,
// and the slice of the hip* host API the library uses) so the SAME kernel sources that hipcc
// compiles for gfx950 can be compiled with g++ and run in the CPU-only build container.  Its only
// purpose is to let `pytest -m "not gpu"` exercise kernel index/barrier logic before spending
// GPU-minutes; parity claims are made by the `-m gpu` tests on real hardware, never through this.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__
#define PLONK_HD inline
#define PLONK_FP_CALL inline
#define PLONK_HD_NOINLINE inline
#define PLONK_SCHED_FENCE() ((void)0)
#define PLONK_DEV inline
#define PLONK_LAMBDA_INLINE
#define PLONK_KERNEL(...) __VA_ARGS__
#define PLONK_DYN_SMEM(name) unsigned char* name = ::hipemu::g_dyn_smem

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace hipemu {
struct Idx3 { unsigned x, y, z; };
extern Idx3 g_threadIdx, g_blockIdx;
extern dim3 g_blockDim, g_gridDim;
extern unsigned char* g_dyn_smem;
void syncthreads();
uint64_t shfl64(uint64_t v, int src_lane);
uint64_t ballot(int pred);
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
}  // namespace hipemu

#define threadIdx (::hipemu::g_threadIdx)
#define blockIdx (::hipemu::g_blockIdx)
#define blockDim (::hipemu::g_blockDim)
#define gridDim (::hipemu::g_gridDim)
#define warpSize 64

inline void __syncthreads() { ::hipemu::syncthreads(); }
inline void __threadfence() {}
inline void __threadfence_block() {}
template <class T> inline T __shfl(T v, int lane, int = 64) {
    static_assert(sizeof(T) <= 8, "");
    uint64_t b = 0; memcpy(&b, &v, sizeof(T)); b = ::hipemu::shfl64(b, lane); T r; memcpy(&r, &b, sizeof(T)); return r;
}
template <class T> inline T __shfl_xor(T v, int mask, int = 64) { return __shfl(v, (int)((threadIdx.x & 63) ^ mask)); }
template <class T> inline T __shfl_down(T v, unsigned d, int = 64) {
    int l = (int)(threadIdx.x & 63) + (int)d; return __shfl(v, l < 64 ? l : (int)(threadIdx.x & 63));
}
template <class T> inline T __shfl_up(T v, unsigned d, int = 64) {
    int l = (int)(threadIdx.x & 63) - (int)d; return __shfl(v, l >= 0 ? l : (int)(threadIdx.x & 63));
}
// cross-lane builtins of the NTT wave kernels (csrc/ntt_wave.hip), emulated on top of the wave shuffle: the three DPP
// controls in use (quad_perm, row_ror), ds_swizzle in bit mode, and the v_permlane32_swap half exchange
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    const int lane = (int)(threadIdx.x & 63), in_row = lane & 15;
    int from = lane;
    bool valid = true;
    if (ctrl < 0x100) from = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);        // quad_perm
    else if (ctrl >= 0x101 && ctrl <= 0x10f) { from = lane + (ctrl - 0x100); valid = in_row + (ctrl - 0x100) <= 15; }  // row_shl:n reads lane + n
    else if (ctrl >= 0x111 && ctrl <= 0x11f) { from = lane - (ctrl - 0x110); valid = in_row >= (ctrl - 0x110); }       // row_shr:n reads lane - n
    else if (ctrl >= 0x121 && ctrl <= 0x12f) from = (lane & ~15) | ((lane + (ctrl - 0x120)) & 15);  // row_ror:n (n = 8: lane ^ 8)
    else abort();
    const int v = __shfl(src, valid ? from : lane);  // every lane takes part in the exchange
    const bool enabled = ((row_mask >> (lane >> 4)) & 1) && ((bank_mask >> ((lane >> 2) & 3)) & 1);
    if (!enabled) return old;
    if (!valid) return bound_ctrl ? 0 : old;
    return v;
}
inline int __builtin_amdgcn_ds_swizzle(int src, int pattern) {
    const int lane = (int)(threadIdx.x & 63);
    if (pattern & 0x8000) abort();  // only bit mode is used
    const int a = pattern & 31, o = (pattern >> 5) & 31, x = (pattern >> 10) & 31;
    return __shfl(src, (lane & 32) | ((((lane & 31) & a) | o) ^ x));
}
struct hipemu_u2 { unsigned v[2]; unsigned operator[](int i) const { return v[i]; } };
inline hipemu_u2 __builtin_amdgcn_permlane32_swap(unsigned vdst, unsigned src, bool, bool) {
    const int lane = (int)(threadIdx.x & 63);
    const unsigned src_lo = __shfl(src, lane & 31), dst_hi = __shfl(vdst, lane | 32);
    hipemu_u2 r;
    r.v[0] = lane < 32 ? vdst : src_lo;   // new vdst: lanes 32-63 take src[0..31]
    r.v[1] = lane < 32 ? dst_hi : src;    // new src:  lanes 0-31 take vdst[32..63]
    return r;
}
inline hipemu_u2 __builtin_amdgcn_permlane16_swap(unsigned vdst, unsigned src, bool, bool) {
    const int lane = (int)(threadIdx.x & 63);
    const unsigned src_even = __shfl(src, lane & ~16), dst_odd = __shfl(vdst, lane | 16);
    hipemu_u2 r;
    r.v[0] = (lane & 16) ? src_even : vdst;  // new vdst: odd rows take the even row below them of src
    r.v[1] = (lane & 16) ? src : dst_odd;    // new src:  even rows take the odd row above them of vdst
    return r;
}
inline unsigned long long __ballot(int pred) { return ::hipemu::ballot(pred); }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
inline unsigned __brev(unsigned x) {
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
    return (x >> 16) | (x << 16);
}
template <class T> inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <class T> inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }

#define PLONK_LAUNCH(kern, grid, block, shmem, stream, ...) \
    ::hipemu::launch((grid), (block), (shmem), [=]() { kern(__VA_ARGS__); })

// ---- host API slice -------------------------------------------------------------------------
typedef int hipError_t;
typedef struct hipemuStream* hipStream_t;
typedef struct hipemuEvent* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost, hipMemcpyDefault };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t { char name[256]; int multiProcessorCount; size_t totalGlobalMem; char gcnArchName[256]; };
hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int d);
hipError_t hipGetDevice(int* d);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int d);
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
inline hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) { *free_b = *total_b = (size_t)1 << 34; return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags = 0);
hipError_t hipHostFree(void* p);
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemset(void* d, int v, size_t n);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st);
hipError_t hipStreamCreate(hipStream_t* s);
enum { hipStreamNonBlocking = 1 };
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipDeviceSynchronize();
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipGetLastError();
const char* hipGetErrorString(hipError_t e);
template <class F> inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }
