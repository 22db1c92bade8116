// wave.h — cross-lane primitives of a 64-lane wavefront on gfx950, without LDS round trips or barriers.
//
//   wave_lane_xor<MASK>(v, lane)  the value lane ^ MASK holds, for MASK = 1, 2, 4, 8, 16, 32:
//       MASK 1, 2   DPP quad_perm          (full-rate VALU move; tools/ubench/ubench2: 37 T lane-ops/s)
//       MASK 8      DPP row_ror:8          (same rate)
//       MASK 4, 16  ds_swizzle, bit mode   (the LDS crossbar without memory: 17 T/s)
//       MASK 32     v_permlane32_swap      (the gfx950 half-wave exchange: 8.7 T/s; ds_bpermute manages 6.5 T/s)
//   wave_swap_words<MASK>(a, b)   swap a register-index bit with a lane bit (the NTT wave kernels' transposes)
//   g1_wave_suffix_scan(p, lane)  lane l <- p_l + .. + p_63 (the weighting of the bucket reduction)
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

// Swap one register-index bit with the lane bit of MASK, for one word of the two registers involved: lanes with the bit
// clear keep a and give b to their partner lane ^ MASK (receiving its a into b), lanes with the bit set keep b and give a.
//   MASK 32, 16  v_permlane32_swap / v_permlane16_swap: the instruction exchanges exactly the halves (rows) this trade
//                moves — one instruction, no select
//   MASK 8, 4    two DPP moves whose bank_mask writes only the receiving lanes (row_ror:8; row_shr:4 / row_shl:4): the
//                select is part of the move
//   MASK 2, 1    a quad_perm DPP move and a select per direction (bank_mask cannot split a quad)
template <unsigned MASK> PLONK_DEV void wave_swap_words(int32_t& a, int32_t& b, bool hi, unsigned lane) {
    if constexpr (MASK == 32) {
        auto r = __builtin_amdgcn_permlane32_swap((uint32_t)a, (uint32_t)b, false, false);
        a = (int32_t)r[0];
        b = (int32_t)r[1];
        return;
    } else if constexpr (MASK == 16) {
        auto r = __builtin_amdgcn_permlane16_swap((uint32_t)a, (uint32_t)b, false, false);
        a = (int32_t)r[0];
        b = (int32_t)r[1];
        return;
    } else if constexpr (MASK == 8) {
        const int na = __builtin_amdgcn_update_dpp(a, b, 0x128, 0xf, 0xC, false);  // lanes 8..15 of a row: a <- b of lane - 8
        const int nb = __builtin_amdgcn_update_dpp(b, a, 0x128, 0xf, 0x3, false);  // lanes 0..7 of a row:  b <- a of lane + 8
        a = na;
        b = nb;
        return;
    } else if constexpr (MASK == 4) {
        const int na = __builtin_amdgcn_update_dpp(a, b, 0x114, 0xf, 0xA, false);  // row_shr:4, banks 1 and 3: a <- b of lane - 4
        const int nb = __builtin_amdgcn_update_dpp(b, a, 0x104, 0xf, 0x5, false);  // row_shl:4, banks 0 and 2: b <- a of lane + 4
        a = na;
        b = nb;
        return;
    } else {
        const int32_t pa = (int32_t)wave_lane_xor<MASK>((uint32_t)a, lane), pb = (int32_t)wave_lane_xor<MASK>((uint32_t)b, lane);
        const int32_t na = hi ? pb : a, nb = hi ? b : pa;
        a = na;
        b = nb;
        return;
    }
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

// the same on lazy limbs (g1l_add: ~2 700 instructions against ~4 600): the Horner step of the comb MSM (msm_comb.h)
// returns false (p untouched in that lane) where the two operands were equal or opposite: g1l_add_fast
template <unsigned MASK, bool OPPOSITE_OK = false> PLONK_DEV bool g1l_wave_reduce_step(G1XyzzL& p, unsigned lane) {
    G1XyzzL o;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        o.x.l[i] = (int32_t)wave_lane_xor<MASK>((uint32_t)p.x.l[i], lane);
        o.y.l[i] = (int32_t)wave_lane_xor<MASK>((uint32_t)p.y.l[i], lane);
        o.zz.l[i] = (int32_t)wave_lane_xor<MASK>((uint32_t)p.zz.l[i], lane);
        o.zzz.l[i] = (int32_t)wave_lane_xor<MASK>((uint32_t)p.zzz.l[i], lane);
    }
    o.inf = wave_lane_xor<MASK>(p.inf ? 1u : 0u, lane) != 0;
    return g1l_add_fast<OPPOSITE_OK>(p, o);
}

// Suffix sums over the lanes of a wave: afterwards lane l holds p_l + p_(l+1) + .. + p_63 (Hillis-Steele; the value of
// lane + d comes through ds_bpermute — 32 words per step beside a general addition of ~4 000 instructions).
// All 64 lanes of the wave must call this together.
PLONK_DEV void g1_wave_suffix_scan(G1Xyzz& p, unsigned lane) {
    for (unsigned d = 1; d < 64; d <<= 1) {
        G1Xyzz o;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            o.x.v[i] = (uint32_t)__shfl_down((int)p.x.v[i], d, 64);
            o.y.v[i] = (uint32_t)__shfl_down((int)p.y.v[i], d, 64);
            o.zz.v[i] = (uint32_t)__shfl_down((int)p.zz.v[i], d, 64);
            o.zzz.v[i] = (uint32_t)__shfl_down((int)p.zzz.v[i], d, 64);
        }
        if (lane + d < 64) g1_add(p, o);
    }
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

