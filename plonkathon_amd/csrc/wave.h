// wave.h — cross-lane primitives of a 64-lane wavefront on gfx950, without LDS round trips or barriers.
//
//   wave_lane_xor<MASK>(v, lane)  the value lane ^ MASK holds, for MASK = 1, 2, 4, 8, 16, 32:
//       MASK 1, 2   DPP quad_perm          (full-rate VALU move; tools/ubench/ubench2: 37 T lane-ops/s)
//       MASK 8      DPP row_ror:8          (same rate)
//       MASK 4, 16  ds_swizzle, bit mode   (the LDS crossbar without memory: 17 T/s)
//       MASK 32     v_permlane32_swap      (the gfx950 half-wave exchange: 8.7 T/s; ds_bpermute manages 6.5 T/s)
//   g1_wave_reduce(p, lane)       butterfly sum of one G1 point per lane: afterwards every lane holds the total
//                                 ("wave-reduced bucket sum": the last six levels of the MSM reductions)
// Used by the NTT wave kernels (ntt.hip) for their in-register digit exchanges and by the MSM kernels (msm.hip).
#pragma once
#include "g1.h"

template <unsigned MASK> PLONK_DEV uint32_t wave_lane_xor(uint32_t v, unsigned lane) {
    if (MASK == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
    if (MASK == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);   // quad_perm [2,3,0,1]
    if (MASK == 4) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (4 << 10) | 0x1f);           // bit mode: lane ^ 4
    if (MASK == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);  // row_ror:8
    if (MASK == 16) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (16 << 10) | 0x1f);         // bit mode: lane ^ 16
    // lane ^ 32: v_permlane32_swap exchanges the upper half of its first operand with the lower half of the second
    auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return lane < 32 ? r[1] : r[0];
}

template <unsigned MASK> PLONK_DEV Fq fq_wave_xor(const Fq& a, unsigned lane) {
    Fq r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = wave_lane_xor<MASK>(a.v[i], lane);
    return r;
}

template <unsigned MASK> PLONK_DEV void g1_wave_reduce_step(G1Xyzz& p, unsigned lane) {
    G1Xyzz o;
    o.x = fq_wave_xor<MASK>(p.x, lane);
    o.y = fq_wave_xor<MASK>(p.y, lane);
    o.zz = fq_wave_xor<MASK>(p.zz, lane);
    o.zzz = fq_wave_xor<MASK>(p.zzz, lane);
    g1_add(p, o);
}

// all 64 lanes of the wave must call this together
PLONK_DEV void g1_wave_reduce(G1Xyzz& p, unsigned lane) {
    g1_wave_reduce_step<32>(p, lane);
    g1_wave_reduce_step<16>(p, lane);
    g1_wave_reduce_step<8>(p, lane);
    g1_wave_reduce_step<4>(p, lane);
    g1_wave_reduce_step<2>(p, lane);
    g1_wave_reduce_step<1>(p, lane);
}

