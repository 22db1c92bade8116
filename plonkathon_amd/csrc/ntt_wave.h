// ntt_wave.h — the in-register "wave" NTT kernels, templated on the field (BN254 Fr for the prover and its parity tests:
// ntt.hip; BLS12-381 Fr for the standalone transform north_star names: ntt_bls.hip).  Device code only; tables, plans and
// launches live in the .hip files.
#pragma once
#include "plonk_internal.h"
#include "wave.h"
#include "fpl.h"

#define NTT_TW_LO_LOG 10

// (wave_for<N>: fp.h — every loop over the element array is expanded at compile time)

// ------------------------------------------------------------------------------------------------
// Variant C ("wave" kernels): the whole transform in registers, exchanges INSIDE a wave by cross-lane moves.
// The default wherever it applies: every size 2^8 .. 2^13 in one launch, 2^16 .. 2^26 as two passes of those.
//
// N = 2^(LOG_E + 6 + 2 L) points, L = 0, 1, 2; N / E threads (64, 256, 1024), each holding E = 2^LOG_E elements in
// registers as 9 signed 29-bit limbs (fpl.h) from the first load to the last store:
//   E = 8: N = 2^9, 2^11, 2^13   radix 8, L x radix 4, radix 8, radix 8     (3 waves per SIMD; 4 at 1024 threads)
//   E = 4: N = 2^8, 2^10, 2^12   radix 4, L x radix 4, 3 x radix 4          (half the registers: 4+ waves per SIMD, and
//                                 twice the workgroups for a lone transform — round 3)
//   E = 8, 512 threads: N = 2^12  radix 8, ONE radix-8 stage on the three wave bits, radix 8, radix 8 (round 4; template argument
//                                 NLDS = 3).  The 1024-thread kernels leave one workgroup per CU — nothing fills the ALUs while its
//                                 sixteen waves exchange through LDS; this form fits two (72 KiB of LDS, 128 registers)
//   E = 2: N = 2^7, 2^9          radix 2 throughout (round 4): the LATENCY kernels.  A lone small transform is bound by the
//                                 dependent instruction chain of ONE wave (a 2^8 transform at E = 4 is ~5 500 instructions per
//                                 lane: 12-14 us by rocprofv3, with 3 of every 4 SIMDs idle at 2^16); two elements per lane cut
//                                 the chain to ~2 300 and put four times the waves on the chip.  Same multiplication count
//                                 per element (radix 2, 4, 8 all pay half a twiddle per element and level), ~10 % more
//                                 additions / range reductions.  They serve 2^14 = 2^7 x 2^7, 2^15 = 2^7 x 2^8 (sizes no pair
//                                 of the larger kernels reaches) and lone 2^16 .. 2^18 (2^7 x 2^9, 2^8 x 2^9, 2^9 x 2^9).
// Decimation in frequency by digits.  The bits of the element index live in three places — the register index (LOG_E
// bits), the lane (6 bits) and, for L > 0, the wave (2 L bits).  A stage works on the digit currently held in the
// register index; between stages that digit is swapped with
//   * two WAVE bits: the only exchange that needs LDS, in rounds of 4 elements per thread (36 B x 4 x threads);
//   * LANE bits: single-bit swaps of a register-index bit with a lane bit.  lane ^ 32 and lane ^ 16 are ONE instruction
//     per pair of words (v_permlane32_swap / v_permlane16_swap exchange exactly the halves / rows a bit swap trades);
//     lane ^ 1, 2, 8 are two selects whose moved operand comes through DPP (quad_perm / row_ror); lane ^ 4 goes
//     through ds_swizzle — no LDS memory, no barrier for the last six levels of every transform.
// After a stage on a digit of a sub-transform of size S (remaining points indexed by `low`), output f is multiplied by
// w_S^(low f) = roots[(N / S) low f]  (the Cooley-Tukey twiddle between the digit DFT and the remaining sub-transforms);
// the root table is stored as Shoup pairs of limbs (no unpacking in the loop).  Outputs appear at frequency
// k = d_A + r_A d_B + ... (first digit least significant), which the final store turns into a natural-order write.
// Coset scaling, zero padding n -> 4n, 1/N and the inverse-coset scaling are fused into the first load / last store.
//
// Range discipline (m = the modulus; "N-form" = limbs 0..7 in [0, 2^29), limb 8 signed and small).  Every stage receives
// N-form elements with |value| < 2.8 m: loads are canonical, twiddle multiplications (fpl_mul_shoup: the factor is a known
// constant, so the product needs 143 multiply-adds instead of fpl_mul's 171) return N-form in (-1.8 m, 2.8 m), and the one
// output of each butterfly group that carries no twiddle factor (index 0) goes through fpl_reduce_small (N-form,
// |value| < 0.51 m).  Inside a radix-8 butterfly five carry sweeps keep every limb inside int32 and every multiplicand
// inside the multiplications' operand bound (|limb| < 1.27 * 2^30): the bounds are written on each line of dft8l / dft4l.
// |value| never exceeds 22.4 m (a sum of eight inputs): fpl_reduce_small's table reaches 23 m, the multiplications 128 m.
#define NTT_SHOUP_STRIDE 20  // int32 words per entry of a root table: w (9), floor(w 2^261 / m) (9), 2 unused
template <class P> struct NttWaveT {
    const Fp<P>* in;
    Fp<P>* out;
    size_t in_bstride, out_bstride;
    unsigned in_len;
    // mode 0: the whole transform, blockIdx.x = batch index.  Two-pass transforms N = R1 R2 (index i1 R2 + c -> frequency
    // k1 + R1 k2): mode 1 = R1-point transforms down the R2 columns (element i1 of column c at in[i1 R2 + c], output k1
    // times w_N^(c k1) to out[k1 R2 + c]); mode 2 = R2-point transforms along the R1 rows (row k1 at in[k1 R2 ..],
    // output k2 to out[k1 + R1 k2]).  blockIdx.x = column / row (XCD-aware order), blockIdx.y = batch index.
    unsigned mode, log_n, log_other;  // log2 of the whole transform and of the stride between successive positions
    // Distributed (multi-GPU) transforms run the same two passes on a slice: rank g of W owns R2 / W columns for the
    // column pass (twiddle column = sub + sub_base) and R1 / W rows for the row pass, whose input arrives from the
    // all-to-all as W chunks [source rank][local row][source's columns]: position c of a row sits at
    // (c >> chunk_log) * chunk_stride + row * 2^chunk_log + (c & (2^chunk_log - 1)).  chunk_stride = 0 means contiguous rows.
    unsigned sub_base, chunk_log, chunk_stride;
    const int32_t* tw_lo;  // inter-pass twiddles w_N^e = tw_lo[e & 1023] * tw_hi[e >> 10], Shoup pairs: applied one after the other (mode 1)
    const int32_t* tw_hi;  //   (for an inverse transform tw_hi carries the factor 1/N as well: tw_always)
    unsigned tw_always;    // 1: multiply even when e == 0 (tw_hi[0] = 1/N); 2: tw_lo is the FULL table of this pass in usage
                           //   order (ntt_interpass_table_kernel; the host launches ntt_wavel_column_kernel): one multiplication per element
    const int32_t* roots;  // the twiddles of this kernel's transform size and direction, Shoup pairs in program order (wavel_tw_*)
    const Fp<P>* in_scale;   // per-element factor at load (coset offset powers) or null
    const Fp<P>* out_scale;  // per-element factor at store or null
    Fp<P> out_scalar;
    unsigned has_out_scalar;
    FpLS<P> w8[3];           // w_8, w_8^2 (= w_4), w_8^3 for the transform direction, Shoup pairs: kernel arguments live in SGPRs
    const int32_t* jm;     // fpl_reduce_small's table of j * m
    // mode 0 only: blockIdx.y = "fan" index f — the same input transformed several times under different scalings (the
    // prover evaluates one coefficient vector on three cosets) or adjacent slices of one buffer.  Bit 0: in += f N, bit 1:
    // out += f N, bit 2: in_scale / out_scale += f N (N = this kernel's transform size; one argument instead of three strides:
    // the 8-element kernels have no scalar registers to spare)
    unsigned fan;
};
#define NTT_FAN_IN 1u
#define NTT_FAN_OUT 2u
#define NTT_FAN_SCALE 4u
#define NTT_TMP_TRANSPOSED 8u  // modes 1 and 2: see wavel_transform

// entry idx of a root table: the Shoup pair of w^idx, 80 bytes as five 16-byte loads
template <class P> PLONK_DEV FpLS<P> wavel_ld_root(const int32_t* tab, unsigned idx) {
    const u32x4* t = reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(tab) + idx * (unsigned)(NTT_SHOUP_STRIDE * sizeof(int32_t)));
    const u32x4 a = t[0], b = t[1], c = t[2], d = t[3], e = t[4];
    FpLS<P> r;
    r.w[0] = (int32_t)a.x; r.w[1] = (int32_t)a.y; r.w[2] = (int32_t)a.z; r.w[3] = (int32_t)a.w;
    r.w[4] = (int32_t)b.x; r.w[5] = (int32_t)b.y; r.w[6] = (int32_t)b.z; r.w[7] = (int32_t)b.w;
    r.w[8] = (int32_t)c.x; r.wp[0] = (int32_t)c.y; r.wp[1] = (int32_t)c.z; r.wp[2] = (int32_t)c.w;
    r.wp[3] = (int32_t)d.x; r.wp[4] = (int32_t)d.y; r.wp[5] = (int32_t)d.z; r.wp[6] = (int32_t)d.w;
    r.wp[7] = (int32_t)e.x; r.wp[8] = (int32_t)e.y;
    return r;
}

// swap register-index bit RB with the lane bit of MASK: lanes with the bit clear keep x[r] and trade x[r | 1 << RB],
// lanes with the bit set keep x[r | 1 << RB] and trade x[r]
template <unsigned E, unsigned RB, unsigned MASK, class P> PLONK_DEV void wavel_swap_bit(FpL<P> (&x)[E], unsigned lane) {
    const bool hi = (lane & MASK) != 0;
    wave_for<E / 2>([&](auto I) {
        constexpr unsigned ih = decltype(I)::value;
        constexpr unsigned r = ((ih >> RB) << (RB + 1)) | (ih & ((1u << RB) - 1)), r1 = r | (1u << RB);  // the indices with bit RB clear
        wave_for<9>([&](auto W) {
            constexpr unsigned i = decltype(W)::value;
            wave_swap_words<MASK>(x[r].l[i], x[r1].l[i], hi, lane);
        });
    });
}

// elements per thread (log2) of the kernel that serves a 2^log_r transform: 2^7 only exists with two; odd sizes take eight,
// even sizes four; `latency` picks the two-element form of 2^9 (a lone small transform: see the top of this file)
PLONK_HD constexpr unsigned wavel_log_e(unsigned log_r, bool latency = false) {
    return (log_r == 7 || (latency && log_r == 9)) ? 1u : ((log_r & 1u) ? 3u : 2u);
}

// ---- twiddle tables in the order the kernel consumes them ("program order") ----------------------------------------------
// Stage s of a wave kernel multiplies register f (f = 1 .. count) by w_R^(low f mult), low < nb: the table holds, stage after
// stage and f after f, one BLOCK of nb Shoup pairs indexed by low — stored as five planes of nb x 16 bytes, so that a load
// instruction of 64 lanes with consecutive `low` reads 1 KB of consecutive bytes (8 cache lines).  The natural-order table
// (80-byte entries at index low f mult) made every one of the five loads of a twiddle touch 40 .. 120 different lines and
// use a fifth to a fifteenth of each: at 2^10 .. 2^13, whose tables do not fit the 32 KB L1, that was ~0.5 MB of L2 -> L1
// traffic per 64 KB transform.  Entries: ~N per kernel size (1020 at 2^10, 2040 at 2^11).
//   twiddled stages: A (digit in the registers at load), the L wave-bit stages, the lane stages except the last
//   E = 2: every stage but the last is twiddled, stage s on the 2^(6 + 2L - s) values of the thread bits below the one it
//   will trade next: blocks of NT, NT / 2, .., 2 entries, factor w^(low 2^s)
// thread-index bits above the lane: two per radix-4 LDS stage; nlds = 3 names the 512-thread form with ONE radix-8 stage on three
PLONK_HD constexpr unsigned wavel_log_t(unsigned nlds) { return 6 + (nlds == 3 ? 3u : 2u * nlds); }
// the nlds of the kernel that serves 2^log_r with 2^log_e elements per thread
PLONK_HD constexpr unsigned wavel_nlds(unsigned log_r, unsigned log_e) { return (log_r == 12 && log_e == 3) ? 3u : (log_r - 6 - log_e) / 2; }
PLONK_HD constexpr unsigned wavel_tw_stages(unsigned log_e, unsigned nlds) {
    return log_e == 1 ? 6 + 2 * nlds : (nlds == 3 ? 3u : 1 + nlds + (log_e == 3 ? 1 : 2));
}
PLONK_HD constexpr unsigned wavel_tw_nb(unsigned log_e, unsigned nlds, unsigned s) {  // distinct values of `low` in stage s
    if (nlds == 3) return s == 0 ? 512u : (s == 1 ? 64u : 8u);
    if (log_e == 1) return 1u << (6 + 2 * nlds - s);
    if (s == 0) return 64u << (2 * nlds);
    if (s <= nlds) return 1u << (6 + 2 * (nlds - s));
    return log_e == 3 ? 8u : (s == nlds + 1 ? 16u : 4u);
}
PLONK_HD constexpr unsigned wavel_tw_count(unsigned log_e, unsigned nlds, unsigned s) {  // factors f = 1 .. count
    return log_e == 1 ? 1u : ((nlds != 3 && s >= 1 && s <= nlds) ? 3u : (1u << log_e) - 1);
}
PLONK_HD constexpr unsigned wavel_tw_mult(unsigned log_e, unsigned nlds, unsigned s) {  // N / S of stage s
    if (nlds == 3) return s == 0 ? 1u : (s == 1 ? 8u : 64u);
    const unsigned log_n = log_e + 6 + 2 * nlds;
    if (log_e == 1) return 1u << s;
    if (s == 0) return 1;
    if (s <= nlds) return 1u << (log_n - (6 + 2 * (nlds - s) + 2));
    return log_e == 3 ? 1u << (log_n - 6) : (s == nlds + 1 ? 1u << (log_n - 6) : 1u << (log_n - 4));
}
PLONK_HD constexpr unsigned wavel_tw_offset(unsigned log_e, unsigned nlds, unsigned s) {  // first entry of stage s's blocks
    unsigned o = 0;
    for (unsigned t = 0; t < s; t++) o += wavel_tw_nb(log_e, nlds, t) * wavel_tw_count(log_e, nlds, t);
    return o;
}
#define NTT_PLANE_WORDS 4  // a plane holds 16 bytes of every entry of its block; five planes per block
#ifndef NTT_TW_PREFETCH
#define NTT_TW_PREFETCH 1  // 0: twiddle factors loaded where they are used (rounds 2 - 5; A/B builds)
#endif

// entry `low` of a block of nb entries
template <class P> PLONK_DEV FpLS<P> wavel_ld_root_planar(const int32_t* block, unsigned nb, unsigned low) {
    // (each plane as "uniform base + 32-bit lane offset": SGPR-base addressing, no 64-bit address arithmetic per lane)
    const unsigned off = low * 16u;
    const auto plane = [&](unsigned pl) PLONK_LAMBDA_INLINE {
        return *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(block + pl * nb * NTT_PLANE_WORDS) + off);
    };
    const u32x4 a = plane(0), b = plane(1), c = plane(2), d = plane(3), e = plane(4);
    FpLS<P> r;
    r.w[0] = (int32_t)a.x; r.w[1] = (int32_t)a.y; r.w[2] = (int32_t)a.z; r.w[3] = (int32_t)a.w;
    r.w[4] = (int32_t)b.x; r.w[5] = (int32_t)b.y; r.w[6] = (int32_t)b.z; r.w[7] = (int32_t)b.w;
    r.w[8] = (int32_t)c.x; r.wp[0] = (int32_t)c.y; r.wp[1] = (int32_t)c.z; r.wp[2] = (int32_t)c.w;
    r.wp[3] = (int32_t)d.x; r.wp[4] = (int32_t)d.y; r.wp[5] = (int32_t)d.z; r.wp[6] = (int32_t)d.w;
    r.wp[7] = (int32_t)e.x; r.wp[8] = (int32_t)e.y;
    return r;
}

// the kernels compiled for 128 VGPRs with 8 elements per thread (WavelCfg::TIGHT): 2^13 and 2^11
#define WAVEL_TIGHT_LOG_N(log_n) ((log_n) == 13 || (log_n) == 11 || (log_n) == 9)
// x[BASE + f] *= w^(low f mult), f = 1 .. COUNT-1, from stage STAGE's blocks of the program-order table;  x[BASE] (no
// factor) is range-reduced instead
template <unsigned LOG_E, unsigned NLDS, unsigned STAGE, unsigned BASE, unsigned COUNT, unsigned E, class P>
PLONK_DEV void wavel_twiddle(FpL<P> (&x)[E], unsigned low, const int32_t* roots, const int32_t* jm) {
    constexpr unsigned LOG_N = LOG_E + wavel_log_t(NLDS), NB = wavel_tw_nb(LOG_E, NLDS, STAGE);
    static_assert(COUNT - 1 == wavel_tw_count(LOG_E, NLDS, STAGE), "twiddle layout");
    const int32_t* blocks = roots + (size_t)wavel_tw_offset(LOG_E, NLDS, STAGE) * NTT_SHOUP_STRIDE;
    constexpr bool TIGHT_MUL = LOG_E == 3 && WAVEL_TIGHT_LOG_N(LOG_N);
    if constexpr (TIGHT_MUL || LOG_E == 3 || !NTT_TW_PREFETCH) {  // (every 8-element kernel sits at 128 registers)
        x[BASE] = fpl_reduce_small(x[BASE], jm);
        wave_for<COUNT - 1>([&](auto F) {
            constexpr unsigned f = decltype(F)::value + 1;
            x[BASE + f] = fpl_mul_shoup<P, TIGHT_MUL>(x[BASE + f], wavel_ld_root_planar<P>(blocks + (f - 1) * NB * NTT_SHOUP_STRIDE, NB, low));
            if constexpr (TIGHT_MUL) PLONK_SCHED_FENCE();  // 128 VGPRs: keeps the scheduler from holding several twiddles in flight
        });
    } else {
        // Round 6: the compiler placed each factor's five loads a dozen instructions before the multiplication that needs them — an
        // L2 round trip in front of every one of the ~14 twiddle multiplications of a transform, which a batch hides behind other
        // workgroups and a lone transform (one round of workgroups, all in the same phase) does not.  Here the factor of
        // multiplication f + 1 is requested BEFORE multiplication f starts (the fences pin the order: the scheduler would sink
        // the loads again to save registers), the first one before the range reduction of the factor-free output: one factor
        // (18 registers) in flight beside the one in use.  Not in the 128-register kernels (TIGHT_MUL).
        FpLS<P> tw = wavel_ld_root_planar<P>(blocks, NB, low);
        PLONK_SCHED_FENCE();
        x[BASE] = fpl_reduce_small(x[BASE], jm);
        wave_for<COUNT - 1>([&](auto F) {
            constexpr unsigned f = decltype(F)::value + 1;
            FpLS<P> nxt = tw;
            if constexpr (f + 1 < COUNT) {
                nxt = wavel_ld_root_planar<P>(blocks + f * NB * NTT_SHOUP_STRIDE, NB, low);
                PLONK_SCHED_FENCE();
            }
            x[BASE + f] = fpl_mul_shoup<P, false>(x[BASE + f], tw);
            tw = nxt;
        });
    }
}
// inputs N-form, |value| < 2.8.  Outputs: x0 in [0, 2^31) (for fpl_reduce_small), x1..x3 multiplicands; |value| < 11.2
template <class P> PLONK_DEV void dft4l(FpL<P>& x0, FpL<P>& x1, FpL<P>& x2, FpL<P>& x3, const FpLS<P>& w2) {
    const FpL<P> a0 = fpl_add(x0, x2), a1 = fpl_add(x1, x3);                 // [0, 2^30)
    const FpL<P> d0 = fpl_sub(x0, x2);                                       // (-2^29, 2^29)
    const FpL<P> d1 = fpl_mul_shoup(fpl_sub(x1, x3), w2);                    // N-form, (-1.8 m, 2.8 m)
    x0 = fpl_add(a0, a1);                                                 // [0, 2^31)
    x2 = fpl_sub(a0, a1);                                                 // (-2^30, 2^30)
    x1 = fpl_add(d0, d1);                                                 // (-2^29, 2^30)
    x3 = fpl_sub(d0, d1);                                                 // (-2^30, 2^29)
}
// inputs N-form, |value| < 2.8.  Outputs: every limb within (-2^30, 2^30] (multiplicands, and fit for fpl_reduce_small);
// |value| <= 22.4
template <class P> PLONK_DEV void dft8l(FpL<P> (&x)[8], const FpLS<P>& w1, const FpLS<P>& w2, const FpLS<P>& w3) {
    const FpL<P> a0 = fpl_add(x[0], x[4]), a1 = fpl_add(x[1], x[5]), a2 = fpl_add(x[2], x[6]), a3 = fpl_add(x[3], x[7]);  // [0, 2^30)
    const FpL<P> b0 = fpl_norm(fpl_sub(x[0], x[4]));                         // N-form (sweep 1)
    const FpL<P> b1 = fpl_mul_shoup(fpl_sub(x[1], x[5]), w1), b2 = fpl_mul_shoup(fpl_sub(x[2], x[6]), w2), b3 = fpl_mul_shoup(fpl_sub(x[3], x[7]), w3);  // operands (-2^29, 2^29)
    const FpL<P> c0 = fpl_norm(fpl_add(a0, a2)), c1 = fpl_norm(fpl_add(a1, a3));  // sums [0, 2^31) -> N-form (sweeps 2, 3)
    const FpL<P> d0 = fpl_norm(fpl_sub(a0, a2));                             // (-2^30, 2^30) -> N-form (sweep 4)
    const FpL<P> d1 = fpl_mul_shoup(fpl_sub(a1, a3), w2);                    // operand (-2^30, 2^30)
    const FpL<P> e0 = fpl_add(b0, b2), e1 = fpl_add(b1, b3);                 // [0, 2^30)
    const FpL<P> f0 = fpl_sub(b0, b2);                                       // (-2^29, 2^29)
    const FpL<P> f1 = fpl_mul_shoup(fpl_sub(b1, b3), w2);                    // operand (-2^29, 2^29)
    x[0] = fpl_add(c0, c1);                                               // [0, 2^30)
    x[4] = fpl_sub(c0, c1);                                               // (-2^29, 2^29)
    x[2] = fpl_add(d0, d1);                                               // [0, 2^30)
    x[6] = fpl_sub(d0, d1);                                               // (-2^29, 2^29)
    x[1] = fpl_norm(fpl_add(e0, e1));                                     // [0, 2^31) -> N-form (sweep 5)
    x[5] = fpl_sub(e0, e1);                                               // (-2^30, 2^30)
    x[3] = fpl_add(f0, f1);                                               // (-2^29, 2^30)
    x[7] = fpl_sub(f0, f1);                                               // (-2^30, 2^29)
}
// inputs N-form, |value| < 2.8.  Outputs: x0 in [0, 2^30) (for fpl_reduce_small), x1 in (-2^29, 2^29) (a multiplicand); |value| < 5.6
template <class P> PLONK_DEV void dft2l(FpL<P>& x0, FpL<P>& x1) {
    const FpL<P> a = fpl_add(x0, x1), d = fpl_sub(x0, x1);
    x0 = a;
    x1 = d;
}
// the digit DFT on the register index: radix 8 (E = 8), radix 4 (E = 4) or radix 2 (E = 2)
template <unsigned E, class P> PLONK_DEV void wavel_dft(FpL<P> (&x)[E], const FpLS<P>& w1, const FpLS<P>& w2, const FpLS<P>& w3) {
    if constexpr (E == 8) dft8l(x, w1, w2, w3);
    else if constexpr (E == 4) dft4l(x[0], x[1], x[2], x[3], w2);
    else dft2l(x[0], x[1]);
}
template <class P> PLONK_DEV void wavel_lds_st(u32x4* lo, u32x4* hi, uint32_t* top, unsigned i, const FpL<P>& a) {
    lo[i] = u32x4{(uint32_t)a.l[0], (uint32_t)a.l[1], (uint32_t)a.l[2], (uint32_t)a.l[3]};
    hi[i] = u32x4{(uint32_t)a.l[4], (uint32_t)a.l[5], (uint32_t)a.l[6], (uint32_t)a.l[7]};
    top[i] = (uint32_t)a.l[8];
}
template <class P> PLONK_DEV FpL<P> wavel_lds_ld(const u32x4* lo, const u32x4* hi, const uint32_t* top, unsigned i) {
    const u32x4 a = lo[i], b = hi[i];
    FpL<P> r;
    r.l[0] = (int32_t)a.x; r.l[1] = (int32_t)a.y; r.l[2] = (int32_t)a.z; r.l[3] = (int32_t)a.w;
    r.l[4] = (int32_t)b.x; r.l[5] = (int32_t)b.y; r.l[6] = (int32_t)b.z; r.l[7] = (int32_t)b.w;
    r.l[8] = (int32_t)top[i];
    return r;
}

// threadIdx.x again, as a value the compiler cannot connect to earlier reads: in the 1024-thread kernel (128 VGPRs) the
// per-thread LDS addresses, lane masks and twiddle indices of later stages were otherwise computed at the top of the
// kernel and carried — spilled — through the first stages
template <bool OPAQUE> PLONK_DEV unsigned wavel_tid() {
    unsigned t = threadIdx.x;
#if defined(__HIP_DEVICE_COMPILE__)
    if (OPAQUE) asm volatile("" : "+v"(t));
#endif
    return t;
}
// the same, pinned behind a value the previous stage produces last (the compiler moved the plain form up to the last barrier)
template <bool OPAQUE> PLONK_DEV unsigned wavel_tid_after(int32_t dep) {
    unsigned t = threadIdx.x;
#if defined(__HIP_DEVICE_COMPILE__)
    if (OPAQUE) asm volatile("" : "+v"(t) : "v"(dep));
#else
    (void)dep;
#endif
    return t;
}

// element g of a uniform base as a 32-bit byte offset (g < 2^27: the wave kernels' transforms have at most 2^26 points):
// SGPR-base addressing, one VGPR per address instead of two and no 64-bit address arithmetic
template <class T> PLONK_DEV const T* wavel_at(const T* base, unsigned g) { return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (g << 5)); }
template <class T> PLONK_DEV T* wavel_at(T* base, unsigned g) { return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + (g << 5)); }

// waves per SIMD the register allocation aims at: 1024-thread workgroups must fit 128 VGPRs (4); the E = 8 forms run
// faster without spills at 3 (measured in round 2: 18.1 vs 16.8 G elements/s at 2^11 x 2048); E = 4 fits 4 without spills
template <unsigned LOG_E, unsigned NLDS> struct WavelCfg {
    static constexpr unsigned E = 1u << LOG_E, LOG_N = LOG_E + wavel_log_t(NLDS), NT = 1u << wavel_log_t(NLDS);
    // TIGHT: the kernel is compiled for 128 VGPRs with the register-saving measures of the 1024-thread kernel (opaque
    // threadIdx re-reads per stage, scheduling fences around the twiddle multiplications)
    static constexpr bool TIGHT = NLDS == 2 || LOG_E == 3;
    static constexpr unsigned WAVES = (NLDS == 2 || LOG_E <= 2) ? 4 : (TIGHT ? 4 : 3);
};

// One transform (or one column / row of a two-pass transform) by one workgroup: the body of both kernels below.
// FULL: the column pass of a two-pass transform with its inter-pass twiddles in ONE table (p.tw_lo; see the end of this file)
// — a kernel of its own, for E = 4 only: the E = 8 kernels sit at 128 registers and any change to this epilogue spills
template <class P, unsigned LOG_E, unsigned NLDS, bool FULL = false>
PLONK_DEV void wavel_transform(const NttWaveT<P>& p, unsigned char* smem) {
    constexpr unsigned E = 1u << LOG_E, LOG_T = wavel_log_t(NLDS), LOG_N = LOG_E + LOG_T, NT = 1u << LOG_T;
    constexpr unsigned LSLOTS = E >= 4 ? 4 : 1;    // elements per thread and exchange round
    u32x4* l_lo = reinterpret_cast<u32x4*>(smem);  // LSLOTS * NT elements as two 16-byte planes and one 4-byte plane
    u32x4* l_hi = l_lo + LSLOTS * NT;
    uint32_t* l_top = reinterpret_cast<uint32_t*>(l_hi + LSLOTS * NT);
    const unsigned tid0 = threadIdx.x;
    const unsigned bidx = p.mode ? blockIdx.y : blockIdx.x, fan = p.mode ? 0u : (blockIdx.y << LOG_N);  // f N
    const Fp<P>* in = p.in + (size_t)bidx * p.in_bstride + ((p.fan & NTT_FAN_IN) ? fan : 0u);
    Fp<P>* out = p.out + (size_t)bidx * p.out_bstride + ((p.fan & NTT_FAN_OUT) ? fan : 0u);
    // column / row of a two-pass transform.  Workgroup b runs on XCD b % 8 (each XCD has its own L2): the remap gives
    // every XCD four ADJACENT columns (rows) per group of 32, so the 32-byte elements it touches share 128-byte lines.
    const unsigned b = blockIdx.x;
    const unsigned sub = !p.mode ? 0 : ((gridDim.x & 31u) ? b : ((b & ~31u) | ((b & 7u) << 2) | ((b >> 3) & 3u)));
    // global index of sub-transform position pos on the input side, of frequency o on the output side
    // The intermediate buffer of a two-pass transform is kept TRANSPOSED (NTT_TMP_TRANSPOSED in `fan`, which modes 1 and 2 do not
    // otherwise use): the column pass stores column c as one contiguous run tmp[c R1 + k1], the row pass gathers row k1 with stride
    // R1 (chunk_log = 0, chunk_stride = R1).  The strided access moves from the stores — which write partial 128-byte lines back
    // once the L2 of an XCD cannot hold every column in flight: 1.3 - 1.4 x the bytes at 2^22 / 2^24 — to the loads, which do not care.
    // (not in the 1024-thread 8-element kernel — with this selection it spills two registers: a 2^13 column pass stores as before)
    const bool tmp_t = !(LOG_E == 3 && NLDS == 2) && p.mode == 1 && (p.fan & NTT_TMP_TRANSPOSED);
    const unsigned in_shift = p.mode == 1 ? p.log_other : 0, out_shift = (p.mode && !tmp_t) ? p.log_other : 0;
    const unsigned in_off = p.mode == 1 ? sub : (p.mode == 2 ? (p.chunk_stride ? sub << p.chunk_log : sub << LOG_N) : 0);
    const unsigned out_off = p.mode ? (tmp_t ? sub << LOG_N : sub) : 0;
    const unsigned chunk_mask = (1u << p.chunk_log) - 1;
    const int32_t* jm = p.jm;
    const FpLS<P> &w8_1 = p.w8[0], &w8_2 = p.w8[1], &w8_3 = p.w8[2];  // kernel arguments: scalar registers

    FpL<P> x[E];
    {
        // The loads of a group of LB elements are issued before the first of them is waited for (round 6): with a bounds check
        // around each load the compiler emitted load - wait - unpack per element, E memory round trips in a row at the start of
        // every workgroup — hidden in a batch, exposed in a lone transform, whose workgroups all start together.  Out-of-range
        // positions (zero padding n -> 4n) read element 0 instead and are masked to zero afterwards.  LB = E up to four elements
        // per thread; the 8-element kernels sit at 128 registers and take two groups of four.
        constexpr unsigned LB = E < 4 ? E : 4;
        wave_for<E / LB>([&](auto GRP) {
            constexpr unsigned j0 = decltype(GRP)::value * LB;
            Fp<P> raw[LB];
            bool inside[LB];
            wave_for<LB>([&](auto J) {  // position j * NT + tid: consecutive lanes read consecutive positions
                constexpr unsigned j = j0 + decltype(J)::value;
                const unsigned pos = j * NT + tid0;
                const unsigned g = p.chunk_stride ? (pos >> p.chunk_log) * p.chunk_stride + (pos & chunk_mask) + in_off : (pos << in_shift) + in_off;
                inside[j - j0] = g < p.in_len;
                raw[j - j0] = fp_load(wavel_at(in, inside[j - j0] ? g : 0u));  // [0, 2m): canonical input, or the column pass's redundant residues
            });
            wave_for<LB>([&](auto J) {
                constexpr unsigned j = j0 + decltype(J)::value;
                wave_for<8>([&](auto W) { raw[j - j0].v[decltype(W)::value] = inside[j - j0] ? raw[j - j0].v[decltype(W)::value] : 0u; });
                x[j] = fpl_from_fp(raw[j - j0]);
            });
        });
    }
    if (p.in_scale) {  // the factor of element j + 1 is requested before element j is multiplied (one load in flight: registers)
        const Fp<P>* in_scale = p.in_scale + ((p.fan & NTT_FAN_SCALE) ? fan : 0u);
        const auto scale_at = [&](unsigned j) PLONK_LAMBDA_INLINE {
            const unsigned g = ((j * NT + tid0) << in_shift) + in_off;
            return fp_load(wavel_at(in_scale, g < p.in_len ? g : 0u));  // (a padded position is zero: any factor will do)
        };
        Fp<P> sc = scale_at(0);
        wave_for<E>([&](auto J) {
            constexpr unsigned j = decltype(J)::value;
            const Fp<P> cur = sc;
            if constexpr (j + 1 < E) sc = scale_at(j + 1);
            x[j] = fpl_mul(x[j], fpl_from_fp(cur));
        });
    }
    if constexpr (E == 2) {
        // LOG_T + 1 radix-2 stages.  Stage s works on index bit LOG_T - s (in the register index), multiplies its odd output by
        // w^(low 2^s), low = the thread bits below, and then trades the register bit for thread bit LOG_T - 1 - s: through LDS
        // (one element per thread) while that is a wave bit, across lanes for the last six
        wave_for<LOG_T>([&](auto S) {
            constexpr unsigned s = decltype(S)::value, tb = LOG_T - 1 - s;
            dft2l(x[0], x[1]);
            const unsigned tid = s ? wavel_tid<false>() : tid0;
            wavel_twiddle<LOG_E, NLDS, s, 0, 2>(x, tid & ((2u << tb) - 1u), p.roots, jm);
            if constexpr (tb >= 6) {
                const bool hi = (tid >> tb) & 1u;  // keeps x[1], gives x[0]; the others keep x[0], give x[1]
                FpL<P> give;
                wave_for<9>([&](auto W) { give.l[decltype(W)::value] = hi ? x[0].l[decltype(W)::value] : x[1].l[decltype(W)::value]; });
                wavel_lds_st(l_lo, l_hi, l_top, tid, give);
                __syncthreads();
                const FpL<P> got = wavel_lds_ld<P>(l_lo, l_hi, l_top, tid ^ (1u << tb));
                __syncthreads();
                wave_for<9>([&](auto W) {
                    constexpr unsigned i = decltype(W)::value;
                    x[0].l[i] = hi ? got.l[i] : x[0].l[i];
                    x[1].l[i] = hi ? x[1].l[i] : got.l[i];
                });
            } else {
                wavel_swap_bit<E, 0, (1u << tb)>(x, tid & 63u);
            }
        });
        dft2l(x[0], x[1]);  // [0, 2^30), (-2^29, 2^29): within the optional multiplications' operand bound
    }
    if constexpr (E >= 4) {
        // stage A: digit = the top LOG_E index bits, low = tid0
        wavel_dft<E>(x, w8_1, w8_2, w8_3);
        wavel_twiddle<LOG_E, NLDS, 0, 0, E>(x, tid0, p.roots, jm);
        if constexpr (NLDS == 3) {
            // ONE radix-8 stage on the three wave bits: register bits (1, 0) <-> thread bits (7, 6) in two rounds of four elements
            // (the radix-4 stages' exchange), then register bit 2 <-> thread bit 8 in a third: a thread keeps four elements and
            // trades four with thread tid ^ 256
            const unsigned tid = wavel_tid<true>();
            const unsigned mine = (tid >> 6) & 3u, rest = tid & ~(3u << 6);
            wave_for<2>([&](auto R2) {
                constexpr unsigned r2 = decltype(R2)::value;
                wave_for<4>([&](auto Q) { wavel_lds_st(l_lo, l_hi, l_top, decltype(Q)::value * NT + tid, x[4 * r2 + decltype(Q)::value]); });
                __syncthreads();
                wave_for<4>([&](auto Q) { x[4 * r2 + decltype(Q)::value] = wavel_lds_ld<P>(l_lo, l_hi, l_top, mine * NT + (rest | (decltype(Q)::value << 6))); });
                __syncthreads();
            });
            const bool hi = (tid >> 8) & 1u;  // keeps x[4 .. 7], gives x[0 .. 3]; the others keep x[0 .. 3], give x[4 .. 7]
            wave_for<4>([&](auto Q) {
                constexpr unsigned q = decltype(Q)::value;
                FpL<P> give;
                wave_for<9>([&](auto W) { give.l[decltype(W)::value] = hi ? x[q].l[decltype(W)::value] : x[q + 4].l[decltype(W)::value]; });
                wavel_lds_st(l_lo, l_hi, l_top, q * NT + tid, give);
            });
            __syncthreads();
            wave_for<4>([&](auto Q) {
                constexpr unsigned q = decltype(Q)::value;
                const FpL<P> got = wavel_lds_ld<P>(l_lo, l_hi, l_top, q * NT + (tid ^ 256u));
                wave_for<9>([&](auto W) {
                    constexpr unsigned i = decltype(W)::value;
                    x[q].l[i] = hi ? got.l[i] : x[q].l[i];
                    x[q + 4].l[i] = hi ? x[q + 4].l[i] : got.l[i];
                });
            });
            __syncthreads();
            dft8l(x, w8_1, w8_2, w8_3);
            wavel_twiddle<LOG_E, NLDS, 1, 0, 8>(x, tid & 63u, p.roots, jm);
        }
        // L radix-4 stages on the wave bits: swap register bits (1, 0) with thread bits (tb + 1, tb)
        wave_for<(NLDS == 3 ? 0u : NLDS)>([&](auto S) {
            constexpr unsigned s = decltype(S)::value;
            constexpr unsigned tb = 6 + 2 * (NLDS - 1 - s);
            const unsigned tid = wavel_tid<WavelCfg<LOG_E, NLDS>::TIGHT>();
            const unsigned mine = (tid >> tb) & 3u, rest = tid & ~(3u << tb);
            wave_for<E / 4>([&](auto R2) {
                constexpr unsigned r2 = decltype(R2)::value;
                wave_for<4>([&](auto Q) { wavel_lds_st(l_lo, l_hi, l_top, decltype(Q)::value * NT + tid, x[4 * r2 + decltype(Q)::value]); });
                __syncthreads();
                wave_for<4>([&](auto Q) { x[4 * r2 + decltype(Q)::value] = wavel_lds_ld<P>(l_lo, l_hi, l_top, mine * NT + (rest | (decltype(Q)::value << tb))); });
                __syncthreads();
            });
            const unsigned low = tid & ((1u << tb) - 1);
            wave_for<E / 4>([&](auto R2) {
                constexpr unsigned r2 = decltype(R2)::value;
                dft4l(x[4 * r2], x[4 * r2 + 1], x[4 * r2 + 2], x[4 * r2 + 3], w8_2);
                wavel_twiddle<LOG_E, NLDS, 1 + s, 4 * r2, 4>(x, low, p.roots, jm);
            });
        });
        const unsigned lane = wavel_tid_after<WavelCfg<LOG_E, NLDS>::TIGHT>(x[E - 1].l[8]) & 63u;
        if constexpr (E == 8) {
            // stage on lane bits 5..3
            wavel_swap_bit<E, 2, 32>(x, lane);
            wavel_swap_bit<E, 1, 16>(x, lane);
            wavel_swap_bit<E, 0, 8>(x, lane);
            dft8l(x, w8_1, w8_2, w8_3);
            wavel_twiddle<LOG_E, NLDS, (NLDS == 3 ? 2u : NLDS + 1), 0, 8>(x, lane & 7u, p.roots, jm);
            // stage on lane bits 2..0
            wavel_swap_bit<E, 2, 4>(x, lane);
            wavel_swap_bit<E, 1, 2>(x, lane);
            wavel_swap_bit<E, 0, 1>(x, lane);
            dft8l(x, w8_1, w8_2, w8_3);
        } else {
            // stages on lane bits (5, 4), (3, 2), (1, 0)
            wavel_swap_bit<E, 1, 32>(x, lane);
            wavel_swap_bit<E, 0, 16>(x, lane);
            dft4l(x[0], x[1], x[2], x[3], w8_2);
            wavel_twiddle<LOG_E, NLDS, NLDS + 1, 0, 4>(x, lane & 15u, p.roots, jm);
            wavel_swap_bit<E, 1, 8>(x, lane);
            wavel_swap_bit<E, 0, 4>(x, lane);
            dft4l(x[0], x[1], x[2], x[3], w8_2);
            wavel_twiddle<LOG_E, NLDS, NLDS + 2, 0, 4>(x, lane & 3u, p.roots, jm);
            wavel_swap_bit<E, 1, 2>(x, lane);
            wavel_swap_bit<E, 0, 1>(x, lane);
            dft4l(x[0], x[1], x[2], x[3], w8_2);
            x[0] = fpl_norm(x[0]);  // [0, 2^31) -> N-form: the optional multiplications below take limbs within (-2^30, 2^30]
        }
    }
    // frequency of register j: digits in processing order, first digit least significant
    unsigned k, shift;
    const unsigned tid = wavel_tid<WavelCfg<LOG_E, NLDS>::TIGHT>();
    const unsigned lane = tid & 63u;
    if constexpr (E == 2) {
        //   one bit per digit: the thread bits from the top down, then j
        k = __brev(tid) >> (32 - LOG_T);
        shift = LOG_T;
    } else if constexpr (E == 8) {
        //   d_A = (lane bit 5) * 4 + thread bits (top pair);  then the remaining wave pairs;  (lane bits 4, 3);  (lane bits 2..0);  j
        //   (without wave stages the first digit is simply lane bits 5..3)
        shift = 3;
        if (NLDS == 3) {  // d_A = thread bits (8, 7, 6); then (lane bits 5..3); (lane bits 2..0); j
            k = ((tid >> 6) & 7u) | (((lane >> 3) & 7u) << 3);
            shift = 6;
        } else if (NLDS) {
            k = (((lane >> 5) & 1u) << 2) | ((tid >> (6 + 2 * (NLDS > 0 ? NLDS - 1 : 0))) & 3u);
            for (unsigned s = 1; s < NLDS; s++) {
                k |= ((tid >> (6 + 2 * (NLDS - 1 - s))) & 3u) << shift;
                shift += 2;
            }
            k |= ((lane >> 3) & 3u) << shift;
            shift += 2;
        } else {
            k = (lane >> 3) & 7u;
        }
        k |= (lane & 7u) << shift;
        shift += 3;
    } else {
        //   every digit has two bits: the thread-index pairs from the top down hold d_A, d_B, ..; j is the last digit
        k = 0;
        wave_for<LOG_T / 2>([&](auto I) {
            constexpr unsigned i = decltype(I)::value;
            k |= ((tid >> (LOG_T - 2 - 2 * i)) & 3u) << (2 * i);
        });
        shift = LOG_T;
    }
    if constexpr (FULL) {  // inter-pass twiddle w_N^(column * frequency) from the table in usage order (ntt_interpass_table_kernel)
        const int32_t* blocks = p.tw_lo + (size_t)sub * (E * NT * NTT_SHOUP_STRIDE);  // this column's E blocks of NT entries: uniform
        wave_for<E>([&](auto J) {
            constexpr unsigned j = decltype(J)::value;
            x[j] = fpl_mul_shoup(x[j], wavel_ld_root_planar<P>(blocks + j * (NT * NTT_SHOUP_STRIDE), NT, tid));
        });
    } else if (p.mode == 1) {  // ... or as two factors from the small tables
        wave_for<E>([&](auto J) {
            constexpr unsigned j = decltype(J)::value;
            const unsigned e = (sub + p.sub_base) * (k | (j << shift));  // < N
            if (e || p.tw_always) {  // two multiplications by table constants (380 instructions) instead of forming their product first (434)
                // (the 512-thread kernel is never a column pass — wave_launch_as refuses — and keeps only the second factor here: with
                // both it spills 7 registers, without the section 35; the allocation it gets this way has none)
                if constexpr (NLDS != 3) x[j] = fpl_mul_shoup(x[j], wavel_ld_root<P>(p.tw_lo, e & ((1u << NTT_TW_LO_LOG) - 1)));
                if (p.log_n > NTT_TW_LO_LOG) x[j] = fpl_mul_shoup(x[j], wavel_ld_root<P>(p.tw_hi, e >> NTT_TW_LO_LOG));
            }
        });
    }
    if (p.out_scale) {
        // (the 1024-thread 8-element kernel has no register left for a per-copy out_scale — 12 B of scratch per lane with it:
        // wave_fan_ok sends such a call down the one-launch-per-copy route)
        constexpr bool FAN_OUT_SCALE = !(LOG_E == 3 && NLDS == 2);
        const Fp<P>* out_scale = p.out_scale + ((FAN_OUT_SCALE && !p.mode && (p.fan & NTT_FAN_SCALE)) ? (blockIdx.y << LOG_N) : 0u);
        wave_for<E>([&](auto J) {
            constexpr unsigned j = decltype(J)::value;
            x[j] = fpl_mul(x[j], fpl_from_fp(fp_load(wavel_at(out_scale, ((k | (j << shift)) << out_shift) + out_off))));
        });
    }
    if (p.has_out_scalar) {
        const FpL<P> sc = fpl_from_fp_uniform(p.out_scalar);
        wave_for<E>([&](auto J) { x[decltype(J)::value] = fpl_mul(x[decltype(J)::value], sc); });
    }
    // |value| <= 22.4 m whatever happened above -> (0.49 m, 1.51 m) -> canonical (the column pass skips that last step).  The table
    // entries of all E values are requested first (round 6: the entry is a dependent load in front of every store otherwise); the
    // 8-element kernels have no registers for that and reduce one value at a time
    if constexpr (E <= 4) {
        FplJmEntry je[E];
        wave_for<E>([&](auto J) { je[decltype(J)::value] = fpl_reduce_small_lookup<P, 1>(x[decltype(J)::value], jm); });
        wave_for<E>([&](auto J) {
            constexpr unsigned j = decltype(J)::value;
            fp_store(wavel_at(out, ((k | (j << shift)) << out_shift) + out_off), fpl_pack_positive(fpl_reduce_small_apply<P, 1>(x[j], je[j]), p.mode != 1));
        });
    } else {
        wave_for<E>([&](auto J) {
            constexpr unsigned j = decltype(J)::value;
            fp_store(wavel_at(out, ((k | (j << shift)) << out_shift) + out_off), fpl_pack_positive(fpl_reduce_small<P, 1>(x[j], jm), p.mode != 1));
        });
    }
}

template <class P, unsigned LOG_E, unsigned NLDS>
__global__ void __launch_bounds__(1u << wavel_log_t(NLDS), (WavelCfg<LOG_E, NLDS>::WAVES)) ntt_wavel_kernel(NttWaveT<P> p) {
    PLONK_DYN_SMEM(smem);
    wavel_transform<P, LOG_E, NLDS>(p, smem);
}
// (E = 4 only.  The E = 8 forms spill at 128 registers with every variant of the table epilogue; compiled for three waves per
// SIMD — 132-135 registers, no scratch — they were measured: batches gain 4-5 %, but a lone 2^21 loses 8 % and a lone 2^24
// 8 %, profiles/r03_q_ntt_e8_column_table_ab.jsonl)
template <class P, unsigned LOG_E, unsigned NLDS>
__global__ void __launch_bounds__(1u << wavel_log_t(NLDS), (WavelCfg<LOG_E, NLDS>::WAVES)) ntt_wavel_column_kernel(NttWaveT<P> p) {
    PLONK_DYN_SMEM(smem);
    wavel_transform<P, LOG_E, NLDS, true>(p, smem);
}



// the Shoup pair of every entry of a packed table, NTT_SHOUP_STRIDE words per entry (what wavel_ld_root reads)
struct Ninv261 { uint32_t l[9]; };
template <class P> __global__ void ntt_limb_table_kernel(const Fp<P>* in, int32_t* out, size_t n, Ninv261 ninv) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const FpLS<P> a = fpl_shoup_from_mont(fp_load(in + i), ninv.l);
    int32_t* o = out + i * NTT_SHOUP_STRIDE;
    for (int w = 0; w < 9; w++) {
        o[w] = a.w[w];
        o[9 + w] = a.wp[w];
    }
    o[18] = o[19] = 0;
}


// one block of a program-order twiddle table (wavel_tw_*): entry low = the Shoup pair of roots[(low f mult) mod N], five planes
template <class P> __global__ void ntt_program_block_kernel(const Fp<P>* roots, unsigned log_n, unsigned nb, unsigned f, unsigned mult, int32_t* block, Ninv261 ninv) {
    const unsigned low = blockIdx.x * blockDim.x + threadIdx.x;
    if (low >= nb) return;
    const FpLS<P> a = fpl_shoup_from_mont(fp_load(roots + ((low * f * mult) & ((1u << log_n) - 1))), ninv.l);
    int32_t e[NTT_SHOUP_STRIDE];
    for (int w = 0; w < 9; w++) {
        e[w] = a.w[w];
        e[9 + w] = a.wp[w];
    }
    e[18] = e[19] = 0;
    for (unsigned pl = 0; pl < 5; pl++)
        for (unsigned w = 0; w < NTT_PLANE_WORDS; w++) block[((size_t)pl * nb + low) * NTT_PLANE_WORDS + w] = e[pl * NTT_PLANE_WORDS + w];
}

// ---- the inter-pass twiddles of a column pass as ONE table in usage order ------------------------------------------------
// The column pass of N = R1 R2 multiplies output k1 of column c by w_N^(c k1).  From the two small tables that is two
// multiplications per element (lo[e & 1023] * hi[e >> 10]); with 288 GB of HBM at a tenth of its bandwidth in these
// ALU-bound kernels, a table of all N products read as one coalesced 80-byte stream trades a multiplication (9 % of a
// two-pass transform's instructions) for bytes.  Layout: for column c, register j: a block of NT Shoup pairs indexed by
// thread (five planes of NT x 16 bytes, wavel_ld_root_planar) at ((c E + j) NT) entries.
// frequency (without the register digit) of thread tid's outputs and the bit position of the register digit: the
// run-time restatement of the index arithmetic at the end of wavel_transform (the parity tests compare both paths)
PLONK_HD unsigned wavel_freq(unsigned log_e, unsigned nlds, unsigned tid, unsigned* shift_out) {
    const unsigned lane = tid & 63u, log_t = wavel_log_t(nlds);
    unsigned k = 0, shift;
    if (log_e == 1) {
        for (unsigned i = 0; i < log_t; i++) k |= ((tid >> (log_t - 1 - i)) & 1u) << i;
        shift = log_t;
    } else if (log_e == 3) {
        shift = 3;
        if (nlds == 3) {
            k = ((tid >> 6) & 7u) | (((lane >> 3) & 7u) << 3);
            shift = 6;
        } else if (nlds) {
            k = (((lane >> 5) & 1u) << 2) | ((tid >> (6 + 2 * (nlds - 1))) & 3u);
            for (unsigned s = 1; s < nlds; s++) {
                k |= ((tid >> (6 + 2 * (nlds - 1 - s))) & 3u) << shift;
                shift += 2;
            }
            k |= ((lane >> 3) & 3u) << shift;
            shift += 2;
        } else {
            k = (lane >> 3) & 7u;
        }
        k |= (lane & 7u) << shift;
        shift += 3;
    } else {
        for (unsigned i = 0; i < log_t / 2; i++) k |= ((tid >> (log_t - 2 - 2 * i)) & 3u) << (2 * i);
        shift = log_t;
    }
    *shift_out = shift;
    return k;
}

// entry i = (c E + j) NT + tid of that table: the Shoup pair of scale * lo[e & 1023] * hi[e >> 10], e = c * k1(tid, j)
template <class P>
__global__ void ntt_interpass_table_kernel(const Fp<P>* lo, const Fp<P>* hi, Fp<P> scale, unsigned log_n, unsigned log_r1, unsigned log_e, int32_t* out,
                                           Ninv261 ninv) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >> log_n) return;
    const unsigned nlds = wavel_nlds(log_r1, log_e), log_t = wavel_log_t(nlds);
    const unsigned tid = (unsigned)i & ((1u << log_t) - 1), j = (unsigned)(i >> log_t) & ((1u << log_e) - 1), c = (unsigned)(i >> log_r1);
    unsigned shift;
    const unsigned k1 = wavel_freq(log_e, nlds, tid, &shift) | (j << shift);
    const unsigned e = c * k1;  // < N
    Fp<P> w = fp_mul(fp_load(lo + (e & ((1u << NTT_TW_LO_LOG) - 1))), scale);
    if (log_n > NTT_TW_LO_LOG) w = fp_mul(w, fp_load(hi + (e >> NTT_TW_LO_LOG)));
    const FpLS<P> a = fpl_shoup_from_mont(w, ninv.l);
    int32_t v[NTT_SHOUP_STRIDE];
    for (int q = 0; q < 9; q++) {
        v[q] = a.w[q];
        v[9 + q] = a.wp[q];
    }
    v[18] = v[19] = 0;
    int32_t* block = out + (i >> log_t << log_t) * NTT_SHOUP_STRIDE;  // the block of (c, j)
    for (unsigned pl = 0; pl < 5; pl++)
        for (unsigned q = 0; q < NTT_PLANE_WORDS; q++) block[((size_t)pl << log_t | tid) * NTT_PLANE_WORDS + q] = v[pl * NTT_PLANE_WORDS + q];
}
