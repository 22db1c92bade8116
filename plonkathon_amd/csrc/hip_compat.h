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
// The 254-bit multiply is ~330 instructions; inlining every call makes the hot kernels 60-300 KB of
// code against a 64 KB instruction cache.  PLONK_FP_CALL = out-of-line (one copy per kernel image).
// out-of-line helper for large, rarely-hot bodies (Keccak-f: inlining it at every sponge call site made
// transcript_kernel 694 KB)
#define PLONK_HD_NOINLINE __host__ __device__ inline __attribute__((noinline))
#ifdef PLONK_FP_OUTLINE
#define PLONK_FP_CALL __host__ __device__ __attribute__((noinline))
#else
#define PLONK_FP_CALL __host__ __device__ __forceinline__
#endif
#define PLONK_DEV __device__ __forceinline__
// a lambda whose body must be inlined at every call (large bodies otherwise become calls, and the arrays they touch move to scratch)
#define PLONK_LAMBDA_INLINE __attribute__((always_inline))
#define PLONK_KERNEL(...) HIP_KERNEL_NAME(__VA_ARGS__)
#define PLONK_LAUNCH(kern, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(kern, grid, block, shmem, stream, __VA_ARGS__)
// stops the scheduler from interleaving independent big-integer ops (which multiplies live registers)
#ifdef __HIP_DEVICE_COMPILE__
#define PLONK_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define PLONK_SCHED_FENCE() ((void)0)
#endif
#define PLONK_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#endif

// PLONK_CHAIN(acc): an empty statement that merely USES the running 64-bit column sum of a product-scanning
// multiplication.  The second use makes every partial sum a leaf for LLVM's reassociation, which otherwise sums a
// column's products from zero and adds the carry of the previous column last — one v_lshl_add_u64 per column, 16 per
// multiplication (5-7 % of the instructions of every ALU-bound kernel here).  With the chain kept as written the carry is
// the addend of the column's first v_mad_u64_u32.  Emits no instruction.  The statements of one multiplication are
// threaded through a scalar token (BEGIN ... END ties it to a result word) instead of being `volatile`, so that the
// scheduler may still interleave independent multiplications: volatile statements keep program order, which cost the
// 3-waves-per-SIMD NTT kernel 4 % at saturation, and were no faster for the MSM loop (profiles/r02_p_chain_ab.txt, r02_q).
#if defined(__HIP_DEVICE_COMPILE__)
#define PLONK_CHAIN_BEGIN() uint32_t plonk_chain_tok = 0
#define PLONK_CHAIN(acc) asm("" : "=s"(plonk_chain_tok) : "v"(acc), "0"(plonk_chain_tok))
#define PLONK_CHAIN_END(word) asm("" : "+v"(word) : "s"(plonk_chain_tok))
#else
#define PLONK_CHAIN_BEGIN() ((void)0)
#define PLONK_CHAIN(acc) ((void)0)
#define PLONK_CHAIN_END(word) ((void)0)
#endif
