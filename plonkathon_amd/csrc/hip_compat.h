// Build-mode switch for the kernel sources.
//   default      : real HIP for gfx950 (hipcc --offload-arch=gfx950) — the product.
//   -DPLONK_EMU  : tests/emu/hip_emu.h, a fiber-based single-source emulation of the HIP execution
//                  model on the host, used ONLY by the CPU test-suite (tests/emu/) to check kernel
//                  index/barrier logic in a container without a GPU.  Never built into
//                  libplonk_hip.so and never loaded by the plonkathon_amd package.
#pragma once
#ifdef PLONK_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#define PLONK_HD __host__ __device__ __forceinline__
#define PLONK_DEV __device__ __forceinline__
#define PLONK_KERNEL(...) HIP_KERNEL_NAME(__VA_ARGS__)
#define PLONK_LAUNCH(kern, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(kern, grid, block, shmem, stream, __VA_ARGS__)
#define PLONK_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif
