// msm.hip — batched fixed-base multi-scalar multiplication on BN254 G1.
//
// Reference behaviour replaced: ec_lincomb -> lincomb -> multisubset
// (/root/reference/curve.py:38-111), i.e. everything Setup.commit does after its ifft
// (setup.py:66-72).  The reference bit-slices the scalars into 255 subsets and adds affine points
// with one Fq inversion per addition (~109k additions for N = 2^11); the result is a group element,
// so any correct schedule yields the same affine point.  The schedules:
//
// A. TABLE MSM — for a reusable SRS (plonk_srs_load_ptau), within the HBM budget the caller grants (1/16 of the device's memory by
//    default; bench.py opts into 180 GB).  ONE table per (device, base set, layout, bits), shared by every context.
//    A1. comb tables (msm_comb.h, the default): 2^(h-1) entries per base, N * ceil(254 / h) mixed additions per MSM — 13 per base
//        from 68.7 GB, 15 from 8.6 GB for 2^11 points — and ceil(254 / h) - 1 doublings per MSM; with TOP TABLES (round 6: floor(254 / h)
//        columns + a joint table per g bases for the bits left over) 12.15 per base from 157.6 GB (h = 21, g = 7).
//    A2. window tables (rounds 2 - 5; plonk_msm_lookup_configure mode | 16): every multiple L[w][i][d] = d * 2^(c w) * P_i a signed
//        c-bit digit can select (128.8 GB at c = 17 for 2^11 points), N * ceil(255 / c) mixed additions of looked-up points:
//      msm_lookup_kernel           lanes walk flat ranges of (scalar, window) items, 64 random bytes per item
//      msm_lookup_finalize_kernel  sum of the workgroup partials + deferred additions -> canonical affine
//    See the section "Lookup MSM" below.
//
// B. BUCKET METHOD (Pippenger) — arbitrary bases (plonk_srs_load_affine), or when no table fits.
//    A window table T[w][i] = 2^(c*w) * P_i is built once per base set; every window of every scalar then lands
//    in ONE shared bucket set and no doublings remain in the per-MSM work:
//   1. msm_sort_kernel        (one workgroup per MSM) scalar -> canonical -> + sum_w 2^(cw+c-1), signed
//                             digits d_w in [-2^(c-1), 2^(c-1)); LDS counting sort (LDS atomics) of all
//                             W*N (point, window) entries by bucket |d|; the sorted entry list and the
//                             bucket boundaries go to HBM (~210 KiB per MSM).
//   2. msm_accumulate_kernel  (G workgroups per MSM) the sorted list is cut into EQUAL flat ranges, one per
//                             lane, so every lane performs the same number of mixed additions whatever the
//                             bucket sizes.  A lane sums its range top-down and stores one partial sum
//                             ("piece") per bucket it touches: a bucket boundary costs a 128-byte store,
//                             never a group operation, so the wave does not serialise on boundaries that
//                             its lanes cross at different steps.  >= 80 % of this method's time.
//   3. msm_bucket_reduce_kernel (two waves per MSM) lane l owns K/128 consecutive buckets: walking them top-down,
//                             run += pieces of bucket k, tot += run; its share is tot + (first bucket - 1) * run, the
//                             second term from a cross-lane suffix scan of the runs (round 4);
//                             shares are summed across waves through LDS and then inside wave 0 by a cross-lane
//                             butterfly (wave.h: DPP / ds_swizzle / v_permlane32_swap — the "wave-reduced bucket
//                             sum"); lane 0 converts the result to the unique affine representative, canonical x||y.
//                             Buckets never exist in memory.
//    Window-table reads hit L2 / Infinity Cache (3.4 MiB at c = 10).
//
// Both inner loops keep the accumulator as 9 x 29-bit limbs with lazy reductions (fpl.h, g1l_madd_fast) and run
// at the rate of a bare mixed-addition loop (13.4 G additions/s chip-wide): the kernels are integer-ALU bound.
// Steps the fast formulas cannot take are deferred to a 256-slot list per MSM; an MSM that overflows it is redone by
// msm_slow_kernel.  msm_lagrange_srs builds the Lagrange-basis view of an SRS out of the same kernels.
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <utility>

#include "plonk_internal.h"
#include "wave.h"

#ifndef MSM_BLOCK
#define MSM_BLOCK 256
#endif
#define MSM_DEFAULT_WINDOW_BITS 10
#define MSM_MAX_WINDOW_BITS 13
#ifndef MSM_ACC_WAVES
#define MSM_ACC_WAVES 4  // waves per SIMD the accumulate kernel is compiled for (register budget 512 / waves)
#endif
// Additions the fast formulas cannot take (accumulator == +-addend: duplicate bases, or the 2^-25 false positive
// of the cheap filter) are deferred to a per-MSM list of this many slots.  An MSM that overflows it (pathological
// input: many equal bases) is recomputed from scratch with the general formulas by msm_*_slow_kernel.
#define MSM_DEFER_CAP 256

// ------------------------------------------------------------------------------------------------
// Window table: table[w*n + i] = 2^(c*w) * bases[i], affine.
__global__ void msm_table_kernel(const G1Affine* bases, size_t n, unsigned c, unsigned W, G1Xyzz* tmp) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        G1Affine b;
        b.x = fp_load(&bases[i].x);
        b.y = fp_load(&bases[i].y);
        G1Xyzz p = g1_xyzz_from_affine(b);
        for (unsigned w = 0; w < W; w++) {
            tmp[(size_t)w * n + i] = p;
            if (w + 1 < W)
                for (unsigned k = 0; k < c; k++) g1_dbl(p);
        }
    }
}

// XYZZ -> affine for a whole array, Montgomery's trick over chunks of 8 (identity -> (0,0)).
#define AFF_CHUNK 8
__global__ void __launch_bounds__(64) g1_batch_to_affine_kernel(const G1Xyzz* in, G1Affine* out, size_t n) {
    size_t nchunks = (n + AFF_CHUNK - 1) / AFF_CHUNK;
    for (size_t ch = (size_t)blockIdx.x * blockDim.x + threadIdx.x; ch < nchunks; ch += (size_t)gridDim.x * blockDim.x) {
        size_t base = ch * AFF_CHUNK;
        Fq pre[AFF_CHUNK];  // prefix products in registers (compile-time indices); zz zzz is formed again on the way back
        Fq acc = fp_one<FqParams>();
        wave_for<AFF_CHUNK>([&](auto K) {
            constexpr unsigned k = decltype(K)::value;
            Fq den = fp_zero<FqParams>();
            if (base + k < n) den = fp_mul(fp_load(&in[base + k].zz), fp_load(&in[base + k].zzz));  // zero <=> identity
            pre[k] = acc;
            if (!fp_is_zero(den)) acc = fp_mul(acc, den);
        });
        acc = fp_inv(acc);
        wave_for_down<AFF_CHUNK>([&](auto K) {
            constexpr unsigned k = decltype(K)::value;
            if (base + k < n) {
                const G1Xyzz& p = in[base + k];
                const Fq zz = fp_load(&p.zz), zzz = fp_load(&p.zzz), den = fp_mul(zz, zzz);
                G1Affine r = g1_affine_identity();
                if (!fp_is_zero(den)) {
                    const Fq t = fp_mul(acc, pre[k]);  // 1 / (zz * zzz)
                    acc = fp_mul(acc, den);
                    r.x = fp_mul(fp_load(&p.x), fp_mul(t, zzz));
                    r.y = fp_mul(fp_load(&p.y), fp_mul(t, zz));
                }
                fp_store(&out[base + k].x, r.x);
                fp_store(&out[base + k].y, r.y);
            }
        });
    }
}

void g1_batch_to_affine(plonk_ctx* ctx, const G1Xyzz* in, G1Affine* out, size_t n) {
    const size_t chunks = (n + AFF_CHUNK - 1) / AFF_CHUNK;
    unsigned g = (unsigned)((chunks + 63) / 64);
    if (g > 65536) g = 65536;
    if (!g) return;
    PLONK_LAUNCH(g1_batch_to_affine_kernel, dim3(g), dim3(64), 0, ctx->stream, in, out, n);
}

// ------------------------------------------------------------------------------------------------
// Sorting.  Entry encoding: bits 0..14 base index, bit 15 sign, bits 16.. window.
struct MsmRecode { uint32_t k[9]; };
struct alignas(8) MsmDeferred { uint32_t bucket, entry; };  // an addition left to msm_bucket_reduce_kernel

PLONK_DEV void msm_recode(const Fr* scalars, size_t idx, const MsmRecode& rc, uint32_t limb[10]) {
    Fr s = fp_from_mont(fp_load(scalars + idx));
    uint64_t carry = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) {
        carry += (uint64_t)(j < 8 ? s.v[j] : 0) + rc.k[j];
        limb[j] = (uint32_t)carry;
        carry >>= 32;
    }
    limb[9] = 0;
}

// Calls emit(w, d) for the W signed c-bit digits d of the recoded scalar, low window first.  The limbs are
// consumed through a 64-bit bit buffer with compile-time limb indices (a dynamically indexed register
// array would live in scratch memory).
template <class F> PLONK_DEV void msm_for_each_digit(const uint32_t limb[10], unsigned c, unsigned W, F emit) {
    const uint32_t mask = (1u << c) - 1, half = 1u << (c - 1);
    uint64_t buf = 0;
    unsigned nb = 0, w = 0;
#pragma unroll
    for (int j = 0; j < 10; j++) {
        buf |= (uint64_t)limb[j] << nb;
        nb += 32;
        while (nb >= c && w < W) {
            emit(w, (int)((uint32_t)buf & mask) - (int)half);
            buf >>= c;
            nb -= c;
            w++;
        }
    }
}

// starts[m][k] (k = 0..K+1): starts[k] = number of entries in buckets 1..k-1, starts[K+1] = total.
// Scalar vector of MSM m: scalars + (m % inner) * stride + (m / inner) * outer_stride  (lets one call commit
// several slices of each row of a [batch][4n] array, e.g. the three quotient parts).
__global__ void __launch_bounds__(MSM_BLOCK) msm_sort_kernel(const Fr* scalars, size_t n, size_t stride, size_t inner,
                                                             size_t outer_stride, unsigned c, unsigned W, MsmRecode rc,
                                                             uint32_t* entries, size_t entry_stride, uint32_t* starts,
                                                             uint32_t* n_deferred) {
    PLONK_DYN_SMEM(smem);
    __shared__ uint32_t chunk_tot[MSM_BLOCK];
    const unsigned K = 1u << (c - 1);
    uint32_t* cnt = reinterpret_cast<uint32_t*>(smem);  // K + 2 counters; cnt[0] collects the zero digits
    const unsigned tid = threadIdx.x;
    const size_t m = blockIdx.x;
    const Fr* sc = scalars + (m % inner) * stride + (m / inner) * outer_stride;
    uint32_t* out = entries + m * entry_stride;
    uint32_t* st = starts + m * (size_t)(K + 2);

    for (unsigned k = tid; k < K + 2; k += MSM_BLOCK) cnt[k] = 0;
    __syncthreads();
    for (size_t i = tid; i < n; i += MSM_BLOCK) {
        uint32_t limb[10];
        msm_recode(sc, i, rc, limb);
        msm_for_each_digit(limb, c, W, [&](unsigned, int d) { atomicAdd(&cnt[d < 0 ? -d : d], 1u); });
    }
    __syncthreads();
    // exclusive scan of cnt[1..K] -> bucket starts (bucket 0 = zero digits, dropped)
    const unsigned per = (K + MSM_BLOCK - 1) / MSM_BLOCK;
    const unsigned lo = 1 + tid * per, hi = (lo + per < K + 1) ? lo + per : K + 1;
    uint32_t sum = 0;
    for (unsigned k = lo; k < hi; k++) sum += cnt[k];
    chunk_tot[tid] = sum;
    __syncthreads();
    for (unsigned off = 1; off < MSM_BLOCK; off <<= 1) {
        uint32_t v = chunk_tot[tid];
        if (tid >= off) v += chunk_tot[tid - off];
        __syncthreads();
        chunk_tot[tid] = v;
        __syncthreads();
    }
    uint32_t run = tid ? chunk_tot[tid - 1] : 0;
    for (unsigned k = lo; k < hi; k++) {
        uint32_t v = cnt[k];
        cnt[k] = run;  // becomes the scatter cursor
        st[k] = run;
        run += v;
    }
    if (tid == MSM_BLOCK - 1) {
        st[K + 1] = chunk_tot[MSM_BLOCK - 1];
        st[0] = 0;
        n_deferred[m] = 0;
    }
    __syncthreads();
    for (size_t i = tid; i < n; i += MSM_BLOCK) {
        uint32_t limb[10];
        msm_recode(sc, i, rc, limb);
        msm_for_each_digit(limb, c, W, [&](unsigned w, int d) {
            if (d) {
                uint32_t pos = atomicAdd(&cnt[d < 0 ? -d : d], 1u);
                out[pos] = (uint32_t)i | (d < 0 ? 0x8000u : 0u) | (w << 16);
            }
        });
    }
}

// ------------------------------------------------------------------------------------------------
// Entries per accumulate lane when E sorted entries are cut into `lanes` equal flat ranges (multiple of 4:
// the entry list is read with 16-byte loads).  Used identically by the two kernels below.
PLONK_HD uint32_t msm_lane_span(uint32_t E, uint32_t lanes) {
    uint32_t per = (E + lanes - 1) / lanes;
    per = (per + 3) & ~3u;
    return per ? per : 4;
}

// Lane t (0 .. 256*G-1 within its MSM) sums its flat range [t*per, (t+1)*per) of the sorted entry list,
// walking from the top entry down.  Whenever the walk leaves a bucket the partial sum of that bucket is
// stored ("piece") and the accumulator restarts: no weighting, no cross-lane reduction, and a bucket
// boundary costs eight 16-byte stores instead of a group addition, so lanes of a wave that cross
// boundaries at different steps do not serialise anything expensive.  Piece slot: t + k - 1 — lanes and
// the buckets they touch are both monotone, so the slot is unique, and msm_bucket_reduce_kernel can
// recompute which lanes touched bucket k from the bucket starts alone.
__global__ void __launch_bounds__(MSM_BLOCK, MSM_ACC_WAVES) msm_accumulate_kernel(const G1Affine* table, size_t table_n,
                                                                   const uint32_t* entries, size_t entry_stride,
                                                                   const uint32_t* starts, unsigned c, unsigned G,
                                                                   G1Xyzz* pieces, size_t piece_stride,
                                                                   MsmDeferred* deferred, uint32_t* n_deferred) {
    PLONK_DYN_SMEM(smem);
    const unsigned K = 1u << (c - 1);
    const unsigned m = blockIdx.x / G, g = blockIdx.x % G;
    const unsigned tid = threadIdx.x;
    uint32_t* st = reinterpret_cast<uint32_t*>(smem);  // K + 2 bucket starts
    const uint32_t* gst = starts + (size_t)m * (K + 2);
    for (unsigned k = tid; k < K + 2; k += MSM_BLOCK) st[k] = gst[k];
    __syncthreads();
    const uint32_t E = st[K + 1];
    const uint32_t* ent = entries + (size_t)m * entry_stride;
    const uint32_t per = msm_lane_span(E, G * MSM_BLOCK);
    const uint32_t t = g * MSM_BLOCK + tid;
    const uint64_t lo64 = (uint64_t)t * per;
    if (lo64 >= E) return;
    const uint32_t lo = (uint32_t)lo64;
    const uint32_t hi = (lo64 + per < E) ? (uint32_t)(lo64 + per) : E;

    // bucket of the top entry: largest k in [1, K] with st[k] <= hi - 1
    unsigned a = 1, b = K;
    while (a < b) {
        unsigned mid = (a + b + 1) >> 1;
        if (st[mid] <= hi - 1) a = mid;
        else b = mid - 1;
    }
    unsigned k = a;
    G1Xyzz* out = pieces + (size_t)m * piece_stride + t - 1;  // out[k] = slot t + k - 1
    // Accumulator kept as 9 signed 29-bit limbs with lazy reductions (fpl.h / g1l_madd_fast): the same ~1550
    // multiplier instructions per mixed addition as the packed canonical form but ~3x fewer of everything
    // else.  The rare steps the fast formulas cannot take (the accumulator equals +-the table point, i.e.
    // duplicate bases) are not resolved here — a call or an inlined general addition in this loop costs
    // 25 % of its speed — they are appended to the MSM's deferred list with their bucket, and
    // msm_bucket_reduce_kernel adds them to that bucket with the general formulas.
    G1XyzzL run = g1l_identity();
    auto flush = [&](unsigned kk) {
        out[kk] = g1l_to_piece(run);
        run.inf = true;
    };
    auto accumulate = [&](const Fq& x, const Fq& y, uint32_t en) {
        if (!g1l_madd_fast(run, x, y, (en & 0x8000u) != 0) && !(fp_is_zero(x) && fp_is_zero(y))) {
            const uint32_t slot = atomicAdd(n_deferred + m, 1u);
            if (slot < MSM_DEFER_CAP) deferred[(size_t)m * MSM_DEFER_CAP + slot] = MsmDeferred{k, en};
        }
    };
    auto step = [&](uint32_t e, uint32_t en) {
        if (e >= hi || e < lo) return;
        if (e < st[k]) {  // left bucket k: its partial sum is complete
            flush(k);
            do k--;
            while (e < st[k]);
        }
        const G1Affine* src = table + (size_t)(en >> 16) * table_n + (en & 0x7fffu);
        const Fq x = fp_load(&src->x), y = fp_load(&src->y);
        accumulate(x, y, en);
    };
    for (uint32_t base = (hi - 1) & ~3u;; base -= 4) {
        const u32x4 q = *reinterpret_cast<const u32x4*>(ent + base);
        step(base + 3, q.w);
        step(base + 2, q.z);
        step(base + 1, q.y);
        step(base, q.x);
        if (base <= lo) break;
    }
    flush(k);
}

// sum_k k * B_k for one MSM from the pieces.  Lane l owns the buckets (l*pb, (l+1)*pb]: walking them from
// the top, run += (pieces of bucket k), tot += run, gives tot = sum (k - l*pb) B_k and run = sum B_k, so
// the lane's share is tot + (l*pb) * run; the shares are tree-reduced through LDS.  Every lane adds into
// tot once per bucket, so the wave stays converged; only the (1-3 piece) inner loop varies.
__global__ void __launch_bounds__(256) msm_bucket_reduce_kernel(const uint32_t* starts, unsigned c, unsigned acc_lanes,
                                                                 const G1Xyzz* pieces, size_t piece_stride,
                                                                 const G1Affine* table, size_t table_n, const MsmDeferred* deferred,
                                                                 size_t deferred_stride, const uint32_t* n_deferred,
                                                                 Fq* out_xy, uint8_t* flags) {
    PLONK_DYN_SMEM(smem);
    G1Xyzz* red = reinterpret_cast<G1Xyzz*>(smem);
    const unsigned K = 1u << (c - 1);
    const unsigned m = blockIdx.x, tid = threadIdx.x, nl = blockDim.x;
    const uint32_t* gst = starts + (size_t)m * (K + 2);
    const uint32_t per = msm_lane_span(gst[K + 1], acc_lanes);
    const unsigned pb = (K + nl - 1) / nl;
    const unsigned b_lo = tid * pb < K ? tid * pb : K;
    const unsigned b_hi = b_lo + pb < K ? b_lo + pb : K;
    const G1Xyzz* pc = pieces + (size_t)m * piece_stride - 1;  // pc[t + k] = piece of lane t for bucket k
    G1Xyzz run = g1_xyzz_identity(), tot = g1_xyzz_identity();
    const MsmDeferred* dfr = deferred + (size_t)m * deferred_stride;
    // additions msm_accumulate_kernel left to the general formulas (normally 0); past the cap the MSM is redone by
    // msm_bucket_slow_kernel, which overwrites this kernel's output
    const uint32_t n_dfr = n_deferred[m] < MSM_DEFER_CAP ? n_deferred[m] : MSM_DEFER_CAP;
    uint32_t s_hi = gst[b_hi + 1];
    for (unsigned k = b_hi; k > b_lo; k--) {
        const uint32_t s_lo = gst[k];
        if (s_hi > s_lo) {
            const uint32_t t_last = (s_hi - 1) / per;
            for (uint32_t t = s_lo / per; t <= t_last; t++) g1_add(run, g1_piece_load(pc + (size_t)t + k));
            for (uint32_t i = 0; i < n_dfr; i++) {
                const MsmDeferred d = dfr[i];
                if (d.bucket != k) continue;
                const G1Affine* src = table + (size_t)(d.entry >> 16) * table_n + (d.entry & 0x7fffu);
                G1Affine pt;
                pt.x = fp_load(&src->x);
                pt.y = fp_load(&src->y);
                if (d.entry & 0x8000u) pt.y = fp_neg(pt.y);
                g1_madd(run, pt);
            }
        }
        g1_add(tot, run);
        s_hi = s_lo;
    }
    // The buckets of lane l weigh b_lo(l) = l pb more than its local walk gave them: sum_l l pb run_l = pb sum_{j >= 1} S_j with
    // S_j = sum_{l >= j} run_l — a suffix scan over the lanes (six general additions inside a wave, by cross-lane moves; wave
    // totals through LDS) and log2 pb doublings, where round 3 ran a c-bit double-and-add of (b_lo, run) per lane: c doublings
    // plus, because a wave executes every branch one of its lanes takes, c - 1 additions.
    {
        const unsigned lane = tid & 63u, wave = tid >> 6, nw = nl >> 6;
        G1Xyzz S = run;
        g1_wave_suffix_scan(S, lane);
        if (nw > 1) {
            if (lane == 0) red[wave] = S;  // lane 0 holds its wave's total
            __syncthreads();
            for (unsigned w = nw - 1; w > wave; w--) g1_add(S, red[w]);
            __syncthreads();
        }
        if (tid) {
            for (unsigned q = pb; q > 1; q >>= 1) g1_dbl(S);
            g1_add(tot, S);
        }
    }
    // shares: across waves through LDS, then the last six levels inside wave 0 by cross-lane moves (wave.h)
    red[tid] = tot;
    __syncthreads();
    for (unsigned s = nl / 2; s >= 64; s >>= 1) {
        if (tid < s) {
            G1Xyzz x = red[tid];
            g1_add(x, red[tid + s]);
            red[tid] = x;
        }
        __syncthreads();
    }
    if (tid >= 64) return;
    G1Xyzz total = red[tid];
    g1_wave_reduce(total, tid);
    if (tid == 0) {  // the unique affine representative, canonical x||y; identity reported out of band
        G1Affine a = g1_to_affine(total);
        flags[m] = g1_affine_is_identity(a) ? 1 : 0;
        fp_store(out_xy + 2 * m, fp_from_mont(a.x));
        fp_store(out_xy + 2 * m + 1, fp_from_mont(a.y));
    }
}

// ------------------------------------------------------------------------------------------------
// Lookup MSM.  MI355X has 288 GB of HBM; a reusable SRS of 2^11 points affords the table of EVERY multiple
//     L[w][i][d] = d * 2^(c w) * P_i,   d = 1 .. 2^(c-1)      (128.8 GB at c = 17, 68.7 GB at c = 16)
// so that an MSM is just N * ceil(255 / c) mixed additions of looked-up points (30 720 at c = 17 against
// 53 248 sorted bucket additions plus the bucket reduction): 64 random bytes from HBM per addition — the chip
// sustains 20 G such reads/s (tools/ubench/gather.hip) against the 16-19 G additions/s its ALUs can do.
// Signed digits as in the bucket method; a lane walks a flat range of (scalar, window) items.

// tmp[i * half + d - 1] = d * wbase[w * n + i] for one window w, XYZZ (converted by g1_batch_to_affine_kernel)
__global__ void __launch_bounds__(64) msm_lookup_fill_kernel(const G1Affine* wbase, size_t n, unsigned c, unsigned w, G1Xyzz* tmp) {
    const size_t half = (size_t)1 << (c - 1);
    const size_t seg_len = half < 256 ? half : 256, nseg = half / seg_len;
    for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < n * nseg; id += (size_t)gridDim.x * blockDim.x) {
        const size_t i = id / nseg, k = (id % nseg) * seg_len;  // this lane fills multiples k+1 .. k+seg_len
        G1Affine b;
        b.x = fp_load(&wbase[(size_t)w * n + i].x);
        b.y = fp_load(&wbase[(size_t)w * n + i].y);
        G1Xyzz acc = g1_xyzz_identity();
#pragma unroll 1
        for (int bit = (int)c - 1; bit >= 0; bit--) {  // acc = k * b
            g1_dbl(acc);
            if ((k >> bit) & 1) g1_madd<true>(acc, b);
        }
        G1Xyzz* out = tmp + i * half + k;
#pragma unroll 1
        for (size_t j = 0; j < seg_len; j++) {
            g1_madd<true>(acc, b);
            out[j] = acc;
        }
    }
}

// item = i * W + w.  Which items a lane adds (`strided`, chosen by the host):
//   strided (n >= lanes: every batch of the prover)  lane t of the MSM's 256 * G lanes takes the scalars i = t, t + lanes, ..,
//       all W windows of one scalar before the next.  At any moment the lanes of a workgroup — and, because every workgroup
//       of a launch walks the same sequence at the same pace, the lanes of the whole chip — read the table slabs of ONE
//       window and `lanes` CONSECUTIVE bases: a contiguous 1 - 2 GB of the 128.8 GB table.  The table look-ups are random
//       64-byte reads; what they cost is address translation, not bandwidth (round 5, profiles/r05_valu_summary.json: random
//       64-byte reads run at 40 G/s over a span of <= 2 GiB and at 20 G/s from 8 GiB up, where 88 - 94 % of the UTCL1 requests
//       miss and the UTCL2 is busy 99.5 % of the time; this kernel with round 4's order — each lane 8 consecutive scalars, the
//       chip spread over the whole table — had 92.5 % UTCL1 misses and the UTCL2 busy 92.6 % of its duration).
//   flat (a lone MSM cut into more lanes than it has scalars)  lane t adds items [t * per, (t + 1) * per).
__global__ void __launch_bounds__(MSM_BLOCK, MSM_ACC_WAVES) msm_lookup_kernel(
    const G1Affine* lookup, size_t table_n, unsigned c, unsigned W, const Fr* scalars, size_t n, size_t stride, size_t inner,
    size_t outer_stride, MsmRecode rc, unsigned G, G1Xyzz* partial, MsmDeferred* deferred, size_t deferred_stride,
    uint32_t* n_deferred, unsigned strided) {
    PLONK_DYN_SMEM(smem);  // MSM_BLOCK x 128 B: first each lane's recoded scalar (10 words), then the tree reduction
    const unsigned m = blockIdx.x / G, g = blockIdx.x % G, tid = threadIdx.x;
    uint32_t* row = reinterpret_cast<uint32_t*>(smem) + tid * 10;
    G1Xyzz* red = reinterpret_cast<G1Xyzz*>(smem);
    const Fr* sc = scalars + (m % inner) * stride + (m / inner) * outer_stride;
    const uint32_t items = (uint32_t)(n * W), lanes = G * MSM_BLOCK, t = g * MSM_BLOCK + tid;
    const uint32_t mask = (1u << c) - 1, half = 1u << (c - 1);
    uint32_t i, w, count, step;
    if (strided) {
        i = t;
        w = 0;
        count = t < n ? (((uint32_t)n - 1 - t) / lanes + 1) * W : 0;
        step = lanes;
    } else {
        const uint32_t per = (items + lanes - 1) / lanes;
        const uint64_t lo64 = (uint64_t)t * per;
        const uint32_t lo = lo64 < items ? (uint32_t)lo64 : items;
        const uint32_t hi = lo64 + per < items ? (uint32_t)(lo64 + per) : items;
        i = lo / W;
        w = lo - i * W;
        count = hi - lo;
        step = 1;
    }

    G1XyzzL run = g1l_identity();
    bool fresh = true;
    for (uint32_t k = 0; k < count; k++) {
        if (fresh) {  // new scalar: canonical value + recoding constant, parked in this lane's LDS row
            uint32_t limb[10];
            msm_recode(sc, i, rc, limb);
#pragma unroll
            for (int j = 0; j < 10; j++) row[j] = limb[j];
            fresh = false;
        }
        const unsigned bit = c * w, j = bit >> 5, sh = bit & 31;
        const uint64_t two = (uint64_t)row[j] | ((uint64_t)row[j + 1] << 32);
        const int d = (int)((uint32_t)(two >> sh) & mask) - (int)half;
        if (d) {
            const uint32_t ad = d < 0 ? (uint32_t)-d : (uint32_t)d;
            const G1Affine* src = lookup + ((((size_t)w * table_n + i) << (c - 1)) + (ad - 1));
            const Fq x = fp_load(&src->x), y = fp_load(&src->y);
            if (!g1l_madd_fast(run, x, y, d < 0) && !(fp_is_zero(x) && fp_is_zero(y))) {  // see msm_accumulate_kernel
                const uint32_t slot = atomicAdd(n_deferred + m, 1u);
                if (slot < MSM_DEFER_CAP) deferred[(size_t)m * deferred_stride + slot] = MsmDeferred{i * W + w, (uint32_t)d};
            }
        }
        if (++w == W) {
            w = 0;
            i += step;
            fresh = true;
        }
    }
    __syncthreads();  // the scalar rows are dead: the same LDS now carries the reduction
    red[tid] = g1l_to_piece(run);
    red[tid] = g1_piece_load(&red[tid]);
    __syncthreads();
    // Tree reduction through LDS.  (A wave-level butterfly for the last six levels — wave.h, as in the bucket reduction
    // below — was measured here and is 1.7 % slower end to end: inlined it costs the 128-VGPR loop 51 spilled registers,
    // out of line the accumulator travels through scratch; profiles/r02_g_msm_reduce_ab.txt.)
    for (unsigned s = MSM_BLOCK / 2; s > 0; s >>= 1) {
        if (tid < s) {
            G1Xyzz x = red[tid];
            g1_add(x, red[tid + s]);
            red[tid] = x;
        }
        __syncthreads();
    }
    if (tid == 0) partial[(size_t)m * G + g] = red[0];
}

// out_xy[m] = canonical affine of sum_g partial[m][g] + the deferred additions; flags[m] = 1 for the identity
__global__ void __launch_bounds__(64) msm_lookup_finalize_kernel(const G1Xyzz* partial, size_t M, unsigned G, const G1Affine* lookup,
                                                                 size_t table_n, unsigned c, unsigned W, const MsmDeferred* deferred,
                                                                 size_t deferred_stride, const uint32_t* n_deferred, Fq* out_xy,
                                                                 uint8_t* flags) {
    for (size_t m = (size_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (size_t)gridDim.x * blockDim.x) {
        G1Xyzz acc = partial[m * G];
        for (unsigned g = 1; g < G; g++) g1_add(acc, partial[m * G + g]);
        const uint32_t nd = n_deferred[m] < MSM_DEFER_CAP ? n_deferred[m] : MSM_DEFER_CAP;  // past the cap: msm_lookup_slow_kernel
        for (uint32_t k = 0; k < nd; k++) {
            const MsmDeferred e = deferred[m * deferred_stride + k];
            const uint32_t i = e.bucket / W, w = e.bucket - i * W;  // `bucket` carries the item index here
            const int d = (int)e.entry;
            const uint32_t ad = d < 0 ? (uint32_t)-d : (uint32_t)d;
            const G1Affine* src = lookup + ((((size_t)w * table_n + i) << (c - 1)) + (ad - 1));
            G1Affine pt;
            pt.x = fp_load(&src->x);
            pt.y = fp_load(&src->y);
            if (d < 0) pt.y = fp_neg(pt.y);
            g1_madd(acc, pt);
        }
        G1Affine a = g1_to_affine(acc);
        flags[m] = g1_affine_is_identity(a) ? 1 : 0;
        fp_store(out_xy + 2 * m, fp_from_mont(a.x));
        fp_store(out_xy + 2 * m + 1, fp_from_mont(a.y));
    }
}

// The same for few MSMs cut into many workgroups (a lone commitment: G = 64): one WAVE per MSM, lane g takes partial g and
// the 64 of them are summed by the cross-lane butterfly of wave.h (six general additions instead of 63 in a row — the
// serial form made a lone 2^11 commitment 0.75 ms, most of the reference-shaped Prover's latency); lane 0 finishes.
__global__ void __launch_bounds__(64) msm_lookup_finalize_wave_kernel(const G1Xyzz* partial, size_t M, unsigned G, const G1Affine* lookup,
                                                                      size_t table_n, unsigned c, unsigned W, const MsmDeferred* deferred,
                                                                      size_t deferred_stride, const uint32_t* n_deferred, Fq* out_xy,
                                                                      uint8_t* flags) {
    const size_t m = blockIdx.x;
    const unsigned lane = threadIdx.x;
    G1Xyzz acc = g1_xyzz_identity();
    for (unsigned g = lane; g < G; g += 64) {  // G <= 64 in practice: at most one partial per lane
        if (g == lane) acc = partial[m * G + g];
        else g1_add(acc, partial[m * G + g]);
    }
    g1_wave_reduce(acc, lane);
    if (lane) return;
    const uint32_t nd = n_deferred[m] < MSM_DEFER_CAP ? n_deferred[m] : MSM_DEFER_CAP;
    for (uint32_t k = 0; k < nd; k++) {
        const MsmDeferred e = deferred[m * deferred_stride + k];
        const uint32_t i = e.bucket / W, w = e.bucket - i * W;
        const int d = (int)e.entry;
        const uint32_t ad = d < 0 ? (uint32_t)-d : (uint32_t)d;
        const G1Affine* src = lookup + ((((size_t)w * table_n + i) << (c - 1)) + (ad - 1));
        G1Affine pt;
        pt.x = fp_load(&src->x);
        pt.y = fp_load(&src->y);
        if (d < 0) pt.y = fp_neg(pt.y);
        g1_madd(acc, pt);
    }
    G1Affine a = g1_to_affine(acc);
    flags[m] = g1_affine_is_identity(a) ? 1 : 0;
    fp_store(out_xy + 2 * m, fp_from_mont(a.x));
    fp_store(out_xy + 2 * m + 1, fp_from_mont(a.y));
}

// Recovery path (see MSM_DEFER_CAP): MSM m is recomputed with the general addition formulas, which handle every
// exceptional case (identity, P == Q, P == -Q), and its output overwritten.  One workgroup per MSM; it exits at
// once unless the MSM overflowed its deferred list, so the launch costs a few microseconds on the normal path.
// kind 0: lookup table (entry |d| of item (i, w));  kind 1: window table T[w][i] (|d| * T by double-and-add).
__global__ void __launch_bounds__(256) msm_slow_kernel(int kind, const G1Affine* tab, size_t table_n, unsigned c, unsigned W,
                                                       const Fr* scalars, size_t n, size_t stride, size_t inner,
                                                       size_t outer_stride, MsmRecode rc, const uint32_t* n_deferred,
                                                       Fq* out_xy, uint8_t* flags) {
    __shared__ G1Xyzz red[256];
    const unsigned m = blockIdx.x, tid = threadIdx.x;
    if (n_deferred[m] <= MSM_DEFER_CAP) return;
    const Fr* sc = scalars + (m % inner) * stride + (m / inner) * outer_stride;
    G1Xyzz acc = g1_xyzz_identity();
    for (size_t i = tid; i < n; i += 256) {
        uint32_t limb[10];
        msm_recode(sc, i, rc, limb);
        msm_for_each_digit(limb, c, W, [&](unsigned w, int d) {
            if (!d) return;
            const uint32_t ad = d < 0 ? (uint32_t)-d : (uint32_t)d;
            const G1Affine* src = kind == 0 ? tab + ((((size_t)w * table_n + i) << (c - 1)) + (ad - 1)) : tab + (size_t)w * table_n + i;
            G1Affine pt;
            pt.x = fp_load(&src->x);
            pt.y = fp_load(&src->y);
            if (d < 0) pt.y = fp_neg(pt.y);
            if (kind == 0) {
                g1_madd<true>(acc, pt);
            } else {
                G1Xyzz t = g1_xyzz_identity();
                for (int bit = (int)c - 1; bit >= 0; bit--) {
                    g1_dbl(t);
                    if ((ad >> bit) & 1) g1_madd<true>(t, pt);
                }
                g1_add(acc, t);
            }
        });
    }
    red[tid] = acc;
    __syncthreads();
    for (unsigned s = 128; s >= 64; s >>= 1) {
        if (tid < s) {
            G1Xyzz x = red[tid];
            g1_add(x, red[tid + s]);
            red[tid] = x;
        }
        __syncthreads();
    }
    if (tid >= 64) return;
    G1Xyzz total = red[tid];
    g1_wave_reduce(total, tid);
    if (tid == 0) {
        G1Affine a = g1_to_affine(total);
        flags[m] = g1_affine_is_identity(a) ? 1 : 0;
        fp_store(out_xy + 2 * m, fp_from_mont(a.x));
        fp_store(out_xy + 2 * m + 1, fp_from_mont(a.y));
    }
}

#define LUT_VERIFY_SAMPLES 8
#include "msm_comb.h"

// ------------------------------------------------------------------------------------------------
// Registry of lookup tables: one per (process, device, base set, layout, bits), shared by every plonk_srs that
// was loaded from the same bytes — several contexts / streams / BatchProvers of one GPU use ONE table.
#include <algorithm>
#include <mutex>
static std::mutex g_lut_mu;
static std::vector<MsmLookupTable*> g_luts;

static void lut_attach(plonk_srs* srs, MsmLookupTable* t) {  // g_lut_mu held
    if (srs->shared == t) return;
    if (srs->shared && --srs->shared->refs == 0) {
        for (size_t k = 0; k < g_luts.size(); k++)
            if (g_luts[k] == srs->shared) g_luts.erase(g_luts.begin() + k);
        hipFree(srs->shared->data);
        delete srs->shared;
    }
    srs->shared = t;
    srs->lookup = t ? t->data : nullptr;
    srs->lookup_bits = t ? t->bits : 0;
    srs->lookup_windows = t ? t->windows : 0;
    srs->lookup_kind = t ? t->kind : 0;
    srs->lookup_top_bits = t ? t->top_bits : 0;
    srs->lookup_top_g = t ? t->top_g : 0;
    if (t) t->refs++;
}

// The registry key is a 64-bit FNV-1a of the loaded bytes — not collision resistant — so a candidate (same device, key,
// number of bases and window bits: lut_find_verified) is only attached after it was compared with THIS SRS on the device:
//   1. its d = 1 entries of window 0 — the bases themselves — ALL equal this SRS's bases (lut_verify_kernel);
//   2. for LUT_VERIFY_SAMPLES bases spread over the set and EVERY window w, its d = 1 entry equals 2^(c w) P_i and its last
//      entry (d = 2^(c-1)) equals 2^(c w + c - 1) P_i, both recomputed here by doublings from this SRS's own base
//      (lut_verify_windows_kernel) — a table of another window size or window count filed under the same key, or one whose
//      higher windows belong to other bases, fails here;
//   3. the registered n_points / bits / windows match what this call would build.
// Every other entry is a function of (bases, number of bases, window bits) alone, computed by this library when the table
// was registered.
__global__ void lut_verify_kernel(const G1Affine* bases, const G1Affine* lookup, size_t n, unsigned c, unsigned* mismatches) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const G1Affine* e = lookup + (i << (c - 1));
    if (!fp_eq(fp_load(&bases[i].x), fp_load(&e->x)) || !fp_eq(fp_load(&bases[i].y), fp_load(&e->y))) atomicAdd(mismatches, 1u);
}
// lane = (sample s, window w): P = 2^(c w) bases[i_s] by doublings; compare with entries d = 1 and d = 2^(c-1) of (w, i_s)
__global__ void __launch_bounds__(64) lut_verify_windows_kernel(const G1Affine* bases, const G1Affine* lookup, size_t n, unsigned c, unsigned W,
                                                                unsigned* mismatches) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= LUT_VERIFY_SAMPLES * W) return;
    const unsigned s = t / W, w = t - s * W;
    const size_t i = n <= LUT_VERIFY_SAMPLES ? (s < n ? s : n - 1) : (size_t)s * (n - 1) / (LUT_VERIFY_SAMPLES - 1);
    G1Affine b;
    b.x = fp_load(&bases[i].x);
    b.y = fp_load(&bases[i].y);
    if (g1_affine_is_identity(b)) return;  // (0, 0) stays (0, 0) in every window: covered by check 1
    G1Xyzz p = g1_xyzz_from_affine(b);
#pragma unroll 1
    for (unsigned k = 0; k < c * w; k++) g1_dbl(p);
    const G1Affine* e = lookup + ((((size_t)w * n + i) << (c - 1)));
    G1Affine a = g1_to_affine(p);
    bool ok = fp_eq(a.x, fp_load(&e[0].x)) && fp_eq(a.y, fp_load(&e[0].y));
#pragma unroll 1
    for (unsigned k = 0; k + 1 < c; k++) g1_dbl(p);
    a = g1_to_affine(p);
    const size_t last = ((size_t)1 << (c - 1)) - 1;
    ok = ok && fp_eq(a.x, fp_load(&e[last].x)) && fp_eq(a.y, fp_load(&e[last].y));
    if (!ok) atomicAdd(mismatches, 1u);
}
static unsigned windows_for(unsigned c);
static size_t msm_comb_stage_entries(size_t n, unsigned h);
static MsmCombScale msm_comb_scale_constant();
static MsmLookupTable* lut_verified(plonk_ctx* ctx, const plonk_srs* srs, MsmLookupTable* t) {  // g_lut_mu held
    if (!t) return nullptr;
    if (t->n_points != srs->n_points) return nullptr;
    if (t->kind == MSM_TABLE_COMB) {
        if (t->bits < 2 || t->bits > MSM_COMB_MAX_TEETH) return nullptr;
        const bool top = t->top_g != 0;
        if (top && !msm_comb_top_ok(t->bits)) return nullptr;
        const MsmCombShape sh = msm_comb_shape(t->bits, top);
        if (t->windows != sh.a || t->top_bits != sh.top_bits || t->top_g != sh.top_g ||
            t->bytes != (msm_comb_blocks(t->n_points, sh) << (t->bits - 1)) * sizeof(G1Affine))
            return nullptr;
    } else if (t->windows != windows_for(t->bits) || t->bytes != t->n_points * t->windows * ((size_t)1 << (t->bits - 1)) * sizeof(G1Affine)) {
        return nullptr;
    }
    void* flag;
    if (ctx_scratch(ctx, 3, 64, &flag) != PLONK_OK) return nullptr;
    unsigned bad = 1;
    if (hipMemsetAsync(flag, 0, 4, ctx->stream) != hipSuccess) return nullptr;
    if (t->kind == MSM_TABLE_COMB) {
        const size_t lanes = srs->n_points + 2 * LUT_VERIFY_SAMPLES;
        PLONK_LAUNCH(msm_comb_verify_kernel, dim3((unsigned)((lanes + 63) / 64)), dim3(64), 0, ctx->stream, (const G1Affine*)srs->bases,
                     (const G1Affine*)t->data, srs->n_points, t->windows, t->bits, (unsigned)LUT_VERIFY_SAMPLES, msm_comb_scale_constant(), (unsigned*)flag);
        if (t->top_g) {
            const size_t groups = msm_comb_top_groups(srs->n_points, t->top_g);
            PLONK_LAUNCH(msm_comb_verify_top_kernel, dim3((unsigned)((groups + 63) / 64)), dim3(64), 0, ctx->stream, (const G1Affine*)srs->bases,
                         (const G1Affine*)t->data, srs->n_points, t->windows, t->bits, t->top_g, (2u << t->top_bits) - 1u, msm_comb_scale_constant(),
                         (unsigned*)flag);
        }
    } else {
        PLONK_LAUNCH(lut_verify_kernel, dim3((unsigned)((srs->n_points + 255) / 256)), dim3(256), 0, ctx->stream, (const G1Affine*)srs->bases,
                     (const G1Affine*)t->data, srs->n_points, t->bits, (unsigned*)flag);
        PLONK_LAUNCH(lut_verify_windows_kernel, dim3((LUT_VERIFY_SAMPLES * t->windows + 63) / 64), dim3(64), 0, ctx->stream,
                     (const G1Affine*)srs->bases, (const G1Affine*)t->data, srs->n_points, t->bits, t->windows, (unsigned*)flag);
    }
    if (hipMemcpyAsync(&bad, flag, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return nullptr;
    return bad ? nullptr : t;
}
// the registered table of this base set with `bits` window bits (0: the one with the most) that passes the comparison above;
// several tables may sit under one key (a collision, or several window sizes): every candidate is tried, widest first
// (top: 0 = without top tables, 1 = with, -1 = either)
static MsmLookupTable* lut_find_verified(plonk_ctx* ctx, const plonk_srs* srs, unsigned kind, unsigned bits, int top = -1) {  // g_lut_mu held
    std::vector<MsmLookupTable*> cand;
    for (MsmLookupTable* t : g_luts)
        if (t->device == srs->device && t->key == srs->content_key && t->n_points == srs->n_points && t->kind == kind && (!bits || t->bits == bits) &&
            (top < 0 || (t->top_g != 0) == (top != 0)))
            cand.push_back(t);
    std::sort(cand.begin(), cand.end(), [](const MsmLookupTable* a, const MsmLookupTable* b) { return a->bits > b->bits; });
    for (MsmLookupTable* t : cand)
        if (lut_verified(ctx, srs, t)) return t;
    return nullptr;
}
// A Lagrange-basis view is a base set of its own with a table of its own: the automatic choice charges the tables of its
// parent SRS and of the parent's other views against the same budget, so that what the caller granted is not spent twice.
static size_t lut_bytes_of_family(const plonk_srs* srs) {  // g_lut_mu held
    const plonk_srs* root = srs->parent ? srs->parent : srs;
    size_t total = 0;
    if (root != srs && root->shared) total += root->shared->bytes;
    for (const auto& kv : root->lagrange)
        if (kv.second != srs && kv.second->shared) total += kv.second->shared->bytes;
    return total;
}

void msm_srs_release(plonk_srs* srs) {
    std::lock_guard<std::mutex> lk(g_lut_mu);
    lut_attach(srs, nullptr);
}

int msm_lookup_info(const plonk_srs* srs, unsigned* bits, size_t* bytes, double* build_s, int* sharers) {
    std::lock_guard<std::mutex> lk(g_lut_mu);
    const MsmLookupTable* t = srs->shared;
    *bits = t ? t->bits : 0;
    *bytes = t ? t->bytes : 0;
    *build_s = t ? t->build_s : 0;
    *sharers = t ? t->refs : 0;
    return PLONK_OK;
}

int msm_lookup_layout(const plonk_srs* srs, unsigned* kind, unsigned* additions_per_base) {
    std::lock_guard<std::mutex> lk(g_lut_mu);
    const MsmLookupTable* t = srs->shared;
    *kind = t ? t->kind : 0;
    *additions_per_base = t ? t->windows : 0;
    return PLONK_OK;
}

bool msm_comb_takes_top(unsigned teeth) { return teeth >= 2 && teeth <= MSM_COMB_MAX_TEETH && msm_comb_top_ok(teeth); }

int msm_lookup_top(const plonk_srs* srs, unsigned* top_bits, unsigned* bases_per_group) {
    std::lock_guard<std::mutex> lk(g_lut_mu);
    const MsmLookupTable* t = srs->shared;
    *top_bits = t ? t->top_bits : 0;
    *bases_per_group = t ? t->top_g : 0;
    return PLONK_OK;
}

static unsigned windows_for(unsigned c) {
    // Smallest W with  s + sum_w 2^(c w + c - 1) < 2^(c W)  for every canonical scalar s < r: the recoding
    // constant is < 2^(cW-1) / (1 - 2^-c), and r < 0.76 * 2^254, so c W >= 255 is enough once c >= 3
    // (17-bit windows need 15 of them, not 16).
    return c >= 3 ? (255 + c - 1) / c : (256 + c - 1) / c;
}

int msm_build_table(plonk_ctx* ctx, plonk_srs* srs, unsigned c) {
    if (srs->table && srs->window_bits == c) return PLONK_OK;
    if (srs->table) {
        hipFree(srs->table);
        srs->table = nullptr;
    }
    const unsigned W = windows_for(c);
    const size_t n = srs->n_points, total = n * W;
    void *tmp = nullptr, *tab = nullptr;
    if (!plonk_dev_malloc(&tmp, total * sizeof(G1Xyzz)) || !plonk_dev_malloc(&tab, total * sizeof(G1Affine))) {
        if (tmp) hipFree(tmp);
        plonk_set_error("hipMalloc of the %zu-point window table failed", total);
        return PLONK_ERR_NOMEM;
    }
    unsigned grid = (unsigned)((n + 63) / 64);
    if (grid > 2048) grid = 2048;
    PLONK_LAUNCH(msm_table_kernel, dim3(grid), dim3(64), 0, ctx->stream, srs->bases, n, c, W, (G1Xyzz*)tmp);
    size_t chunks = (total + AFF_CHUNK - 1) / AFF_CHUNK;
    unsigned g2 = (unsigned)((chunks + 63) / 64);
    if (g2 > 4096) g2 = 4096;
    PLONK_LAUNCH(g1_batch_to_affine_kernel, dim3(g2), dim3(64), 0, ctx->stream, (const G1Xyzz*)tmp, (G1Affine*)tab, total);
    PLONK_CHECK_HIP(hipGetLastError());
    PLONK_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    hipFree(tmp);
    srs->table = (G1Affine*)tab;
    srs->window_bits = c;
    srs->n_windows = W;
    return PLONK_OK;
}

static void msm_recode_constant(unsigned c, unsigned W, MsmRecode* rc) {
    memset(rc, 0, sizeof *rc);
    for (unsigned w = 0; w < W; w++) {
        unsigned bit = c * w + c - 1;
        rc->k[bit >> 5] |= 1u << (bit & 31);
    }
}

static size_t msm_lookup_bytes(size_t n, unsigned c) {  // table + the XYZZ staging buffer of one window
    const size_t half = (size_t)1 << (c - 1);
    return n * windows_for(c) * half * sizeof(G1Affine) + n * half * sizeof(G1Xyzz);
}

// Builds srs->lookup for window size c.  PLONK_ERR_NOMEM (nothing allocated, nothing changed) if it does not fit.
static int msm_lookup_build(plonk_ctx* ctx, plonk_srs* srs, unsigned c) {  // g_lut_mu held
    const auto t0 = std::chrono::steady_clock::now();
    const unsigned W = windows_for(c);
    const size_t n = srs->n_points, half = (size_t)1 << (c - 1);
    void *wx = nullptr, *wb = nullptr, *tmp = nullptr, *tab = nullptr;
    auto fail = [&]() {
        if (wx) hipFree(wx);
        if (wb) hipFree(wb);
        if (tmp) hipFree(tmp);
        if (tab) hipFree(tab);
        (void)hipGetLastError();
        plonk_set_error("the %u-bit lookup table (%zu MiB) does not fit in device memory", c, msm_lookup_bytes(n, c) >> 20);
        return PLONK_ERR_NOMEM;
    };
    if (!plonk_dev_malloc(&tab, n * W * half * sizeof(G1Affine))) return fail();
    if (!plonk_dev_malloc(&tmp, n * half * sizeof(G1Xyzz))) return fail();
    if (!plonk_dev_malloc(&wx, n * W * sizeof(G1Xyzz))) return fail();
    if (!plonk_dev_malloc(&wb, n * W * sizeof(G1Affine))) return fail();
    // window bases 2^(c w) P_i, affine
    unsigned grid = (unsigned)((n + 63) / 64);
    if (grid > 2048) grid = 2048;
    PLONK_LAUNCH(msm_table_kernel, dim3(grid), dim3(64), 0, ctx->stream, srs->bases, n, c, W, (G1Xyzz*)wx);
    size_t chunks = (n * W + AFF_CHUNK - 1) / AFF_CHUNK;
    unsigned g2 = (unsigned)((chunks + 63) / 64);
    PLONK_LAUNCH(g1_batch_to_affine_kernel, dim3(g2 > 4096 ? 4096 : g2), dim3(64), 0, ctx->stream, (const G1Xyzz*)wx, (G1Affine*)wb,
                 n * W);
    const size_t seg_len = half < 256 ? half : 256, fill_lanes = n * (half / seg_len);
    unsigned gf = (unsigned)((fill_lanes + 63) / 64);
    if (gf > 65536) gf = 65536;
    chunks = (n * half + AFF_CHUNK - 1) / AFF_CHUNK;
    unsigned ga = (unsigned)((chunks + 63) / 64 > 65536 ? 65536 : (chunks + 63) / 64);
    for (unsigned w = 0; w < W; w++) {
        PLONK_LAUNCH(msm_lookup_fill_kernel, dim3(gf), dim3(64), 0, ctx->stream, (const G1Affine*)wb, n, c, w, (G1Xyzz*)tmp);
        PLONK_LAUNCH(g1_batch_to_affine_kernel, dim3(ga), dim3(64), 0, ctx->stream, (const G1Xyzz*)tmp,
                     (G1Affine*)tab + (size_t)w * n * half, n * half);
    }
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return fail();
    hipFree(wx);
    hipFree(wb);
    hipFree(tmp);
    MsmLookupTable* t = new MsmLookupTable();
    t->device = srs->device;
    t->key = srs->content_key;
    t->n_points = n;
    t->kind = MSM_TABLE_WINDOWS;
    t->bits = c;
    t->windows = W;
    t->data = (G1Affine*)tab;
    t->bytes = n * W * half * sizeof(G1Affine);
    t->build_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    g_luts.push_back(t);
    lut_attach(srs, t);
    return PLONK_OK;
}

// ---- comb tables (msm_comb.h) -------------------------------------------------------------------
// XYZZ staging of the build: an eighth of the table's entries at a time (whole bases), at most 2^27 of them (17 GB)
static size_t msm_comb_stage_entries(size_t n, unsigned h) {
    const size_t half = (size_t)1 << (h - 1);
    size_t bases = n / 8 ? n / 8 : 1;
    while (bases > 1 && bases * half > ((size_t)1 << 27)) bases /= 2;
    return bases * half;
}
static size_t msm_comb_bytes(size_t n, unsigned h, bool top = false) {  // table + staging
    return (msm_comb_blocks(n, msm_comb_shape(h, top)) << (h - 1)) * sizeof(G1Affine) + msm_comb_stage_entries(n, h) * sizeof(G1Xyzz);
}

// R^-1 mod r as a plain integer (R = 2^261, Fr's Montgomery radix): the comb tables hold multiples of R^-1 P_i (msm_comb.h)
static MsmCombScale msm_comb_scale_constant() {
    Fr one_plain = fp_zero<FrParams>();
    one_plain.v[0] = 1;
    const Fr c = fp_from_mont(one_plain);  // fp_from_mont multiplies the integer it is given by R^-1 mod r: here the integer 1
    MsmCombScale k;
    for (int i = 0; i < 8; i++) k.c[i] = c.v[i];
    return k;
}

// Builds the comb table of h teeth.  PLONK_ERR_NOMEM (nothing allocated, nothing changed) if it does not fit.
// top: with top tables (msm_comb.h: floor(254 / h) columns, a joint table per group of bases for the bits left over)
static int msm_comb_build(plonk_ctx* ctx, plonk_srs* srs, unsigned h, bool top = false) {  // g_lut_mu held
    const auto t0 = std::chrono::steady_clock::now();
    if (top && !msm_comb_top_ok(h)) {
        plonk_set_error("a comb of %u teeth takes no top tables (254 mod teeth must be 1 or 2)", h);
        return PLONK_ERR_ARG;
    }
    const MsmCombShape sh = msm_comb_shape(h, top);
    if (!msm_comb_top_reach_ok(srs->n_points, sh)) {
        plonk_set_error("%zu bases are too many for the top tables of a %u-tooth comb (a virtual scalar's block offset must fit 31 bits)", srs->n_points, h);
        return PLONK_ERR_ARG;
    }
    const unsigned a = sh.a, sb = h - 1 < MSM_COMB_SEG_BITS ? h - 1 : MSM_COMB_SEG_BITS;
    const size_t n = srs->n_points, half = (size_t)1 << (h - 1), stage = msm_comb_stage_entries(n, h), chunk_bases = stage / half;
    const size_t blocks = msm_comb_blocks(n, sh);
    void *gx = nullptr, *gb = nullptr, *dx = nullptr, *db = nullptr, *tmp = nullptr, *tab = nullptr;
    auto fail = [&]() {
        for (void* q : {gx, gb, dx, db, tmp, tab})
            if (q) hipFree(q);
        (void)hipGetLastError();
        plonk_set_error("the %u-tooth comb table (%zu MiB) does not fit in device memory", h, msm_comb_bytes(n, h, top) >> 20);
        return PLONK_ERR_NOMEM;
    };
    if (!plonk_dev_malloc(&tab, blocks * half * sizeof(G1Affine))) return fail();
    if (!plonk_dev_malloc(&tmp, stage * sizeof(G1Xyzz))) return fail();
    if (!plonk_dev_malloc(&gx, n * h * sizeof(G1Xyzz))) return fail();
    if (!plonk_dev_malloc(&gb, n * h * sizeof(G1Affine))) return fail();
    if (!plonk_dev_malloc(&dx, n * (sb ? sb : 1) * sizeof(G1Xyzz))) return fail();
    if (!plonk_dev_malloc(&db, n * (sb ? sb : 1) * sizeof(G1Affine))) return fail();
    // P'_i = R^-1 P_i (the scalars arrive as Montgomery residues: msm_comb.h), through the staging buffers of the next step
    unsigned grid = (unsigned)((n + 63) / 64);
    if (grid > 2048) grid = 2048;
    void* pb = nullptr;
    if (!plonk_dev_malloc(&pb, n * sizeof(G1Affine))) return fail();
    PLONK_LAUNCH(msm_comb_scale_kernel, dim3(grid), dim3(64), 0, ctx->stream, (const G1Affine*)srs->bases, n, msm_comb_scale_constant(), (G1Xyzz*)gx);
    g1_batch_to_affine(ctx, (const G1Xyzz*)gx, (G1Affine*)pb, n);
    // tooth points G_k = 2^(a k) P'_i (k < h) and the Gray-code steps 2 G_k (k < sb), affine
    PLONK_LAUNCH(msm_table_kernel, dim3(grid), dim3(64), 0, ctx->stream, (const G1Affine*)pb, n, a, h, (G1Xyzz*)gx);
    g1_batch_to_affine(ctx, (const G1Xyzz*)gx, (G1Affine*)gb, n * h);
    if (sb) {
        unsigned gd = (unsigned)((n * sb + 255) / 256);
        PLONK_LAUNCH(msm_comb_delta_kernel, dim3(gd > 4096 ? 4096 : gd), dim3(256), 0, ctx->stream, (const G1Affine*)gb, n * sb, (G1Xyzz*)dx);
        g1_batch_to_affine(ctx, (const G1Xyzz*)dx, (G1Affine*)db, n * sb);
    }
    for (size_t i0 = 0; i0 < n; i0 += chunk_bases) {
        const size_t nb = n - i0 < chunk_bases ? n - i0 : chunk_bases, lanes = nb << (h - 1 - sb);
        unsigned gf = (unsigned)((lanes + 63) / 64 > 65536 ? 65536 : (lanes + 63) / 64);
        PLONK_LAUNCH(msm_comb_fill_kernel, dim3(gf), dim3(64), 0, ctx->stream, (const G1Affine*)gb, (const G1Affine*)db, n, i0, nb, h, sb,
                     (G1Xyzz*)tmp);
        g1_batch_to_affine(ctx, (const G1Xyzz*)tmp, (G1Affine*)tab + i0 * half, nb * half);
    }
    bool build_failed = false;
    if (top) {
        // tooth points of the top tables 2^(L - j) P'_i (through gx / gb, free by now), then the joint tables a chunk of groups at a
        // time; a block past the last group (the columns are padded to equal lengths) stays all identity
        PLONK_LAUNCH(msm_comb_top_base_kernel, dim3(grid), dim3(64), 0, ctx->stream, (const G1Affine*)pb, n, a * h, a, sh.top_g, (G1Xyzz*)gx);
        g1_batch_to_affine(ctx, (const G1Xyzz*)gx, (G1Affine*)gb, n);
        const size_t groups = msm_comb_top_groups(n, sh.top_g), vblocks = blocks - n, run = sh.top_g >= 2 ? (size_t)sh.top_b * sh.top_b : sh.top_b;
        for (size_t g0 = 0; g0 < vblocks; g0 += chunk_bases) {
            const size_t nb = vblocks - g0 < chunk_bases ? vblocks - g0 : chunk_bases;
            const size_t ng = g0 >= groups ? 0 : (groups - g0 < nb ? groups - g0 : nb);
            if (hipMemsetAsync(tmp, 0, nb * half * sizeof(G1Xyzz), ctx->stream) != hipSuccess) {
                build_failed = true;
                break;
            }
            if (ng) {
                const size_t lanes = ng * (sh.top_entries / run);
                unsigned gf = (unsigned)((lanes + 63) / 64 > 65536 ? 65536 : (lanes + 63) / 64);
                PLONK_LAUNCH(msm_comb_top_fill_kernel, dim3(gf), dim3(64), 0, ctx->stream, (const G1Affine*)gb, n, g0, ng, h - 1, sh.top_g, sh.top_b,
                             sh.top_entries, (G1Xyzz*)tmp);
            }
            g1_batch_to_affine(ctx, (const G1Xyzz*)tmp, (G1Affine*)tab + (n + g0) * half, nb * half);
        }
    }
    if (build_failed || hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
        hipFree(pb);
        return fail();
    }
    for (void* q : {gx, gb, dx, db, tmp, pb}) hipFree(q);
    MsmLookupTable* t = new MsmLookupTable();
    t->device = srs->device;
    t->key = srs->content_key;
    t->n_points = n;
    t->kind = MSM_TABLE_COMB;
    t->bits = h;
    t->windows = a;
    t->top_bits = sh.top_bits;
    t->top_g = sh.top_g;
    t->data = (G1Affine*)tab;
    t->bytes = blocks * half * sizeof(G1Affine);
    t->build_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    g_luts.push_back(t);
    lut_attach(srs, t);
    return PLONK_OK;
}

static size_t msm_table_bytes(size_t n, unsigned kind, unsigned bits, bool top = false) {
    return kind == MSM_TABLE_COMB ? msm_comb_bytes(n, bits, top) : msm_lookup_bytes(n, bits);
}
static int msm_table_build(plonk_ctx* ctx, plonk_srs* srs, unsigned kind, unsigned bits, bool top = false) {
    return kind == MSM_TABLE_COMB ? msm_comb_build(ctx, srs, bits, top) : msm_lookup_build(ctx, srs, bits);
}

static size_t msm_default_lookup_budget() {
    // The table is a memory-for-time trade the CALLER opts into beyond a modest default: 1/16 of the device's memory (18 GB of an
    // MI355X's 288: the comb of 17 teeth for 2^11 bases, 8.6 GB + 1.1 GB while it is built, 15 additions per base) and never more
    // than a quarter of what is FREE at the moment — the default is per process and per SRS family, so several processes or
    // several SRS on one device each take theirs (eight ranks sharing a GPU: 8 x 9.7 GB), and a device that is already
    // nearly full must not be pushed over by a table nobody asked for.  More only through plonk_msm_lookup_configure(budget)
    // or PLONK_MSM_TABLE_GB (bench.py asks for 180 GB: the 157.6 GB comb of 21 teeth with top tables, 12.15 additions; 100 GB buys
    // the 68.7 GB comb of 20 teeth, 13 additions).
    // Window tables, measured (profiles/r05_d_msm_sweep.jsonl, 1152 MSMs of 2^11 per call): c = 11 4.50 ms, 12 4.11, 13 3.86, 14 3.65.
    const char* e = getenv("PLONK_MSM_TABLE_GB");
    if (e && atof(e) > 0) return (size_t)(atof(e) * 1e9);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || !total_b) {
        (void)hipGetLastError();
        return (size_t)4 << 30;
    }
    return total_b / 16 < free_b / 4 ? total_b / 16 : free_b / 4;
}

// Decides whether this call runs on a lookup table: attaches the table another context of this device already
// built for the same bases, or builds one on first use.
static bool msm_lookup_prepare(plonk_ctx* ctx, plonk_srs* srs) {
    if (ctx->msm_lookup_mode == 1) return false;
    const unsigned want = ctx->msm_lookup_bits, kind = ctx->msm_lookup_kind;
    const bool wtop = want && kind == MSM_TABLE_COMB && ctx->msm_lookup_top;  // an explicit size names its variant
    std::lock_guard<std::mutex> lk(g_lut_mu);
    const auto attached_is = [&](unsigned bits) { return srs->shared && srs->lookup_kind == kind && srs->lookup_bits == bits && (srs->lookup_top_g != 0) == wtop; };
    if (ctx->msm_lookup_mode == 2) {  // forced size, any base set
        if (attached_is(want)) return true;
        if (MsmLookupTable* t = lut_find_verified(ctx, srs, kind, want, wtop)) {
            lut_attach(srs, t);
            return true;
        }
        return msm_table_build(ctx, srs, kind, want, wtop) == PLONK_OK;
    }
    if (!srs->fixed) return false;
    if (srs->shared && srs->lookup_kind == kind && (!want || attached_is(want))) return true;
    if (want) {
        if (MsmLookupTable* t = lut_find_verified(ctx, srs, kind, want, wtop)) {
            lut_attach(srs, t);
            return true;
        }
    }
    if (srs->lookup_failed) return false;
    const size_t budget = ctx->msm_lookup_budget ? ctx->msm_lookup_budget : msm_default_lookup_budget();
    // A table another context of this device already built for these bases is taken as it is — unless this context's
    // budget affords a better one (fewer additions per base), which is then built and shared in its turn.
    MsmLookupTable* have = want ? nullptr : lut_find_verified(ctx, srs, kind, 0);
    // The automatic choice charges the tables of the same SRS family (an SRS and its Lagrange-basis views) against one
    // budget.  An explicit size (`want`) is an explicit request and only has to fit the budget by itself.
    const size_t used = want ? 0 : lut_bytes_of_family(srs);
    // below 8 bits the table no longer beats the bucket method — which, however, cannot index more than 2^15 bases, so
    // larger base sets accept any table that fits
    const unsigned c_min = want ? want : (srs->n_points > 32768 ? 4 : 8);
    const unsigned c_max = want ? want : (kind == MSM_TABLE_COMB ? 22 : 17);
    // Candidates in the order of their additions per base, the smaller table first among equals (a comb one tooth shorter with
    // as many columns costs the same additions for half the memory).  Combs come without and — where 254 mod teeth allows — with
    // top tables (msm_comb.h): 21 teeth + top tables = 12.15 additions per base of 2^11 from 157.5 GB, between the 13 of 20 teeth
    // (68.7 GB) and the 12 of 22 (275 GB).
    struct Cand { unsigned c; bool top; double adds; size_t bytes; };
    std::vector<Cand> cands;
    const auto adds_of = [&](unsigned c, bool top) {
        if (kind != MSM_TABLE_COMB) return (double)windows_for(c);
        const MsmCombShape sh = msm_comb_shape(c, top);
        return (double)sh.a + (top ? 1.0 / sh.top_g : 0.0);
    };
    for (unsigned c = c_max; c >= c_min; c--)
        for (int top = 0; top < 2; top++) {
            if (top && (kind != MSM_TABLE_COMB || !msm_comb_top_ok(c) || !msm_comb_top_reach_ok(srs->n_points, msm_comb_shape(c, true)))) continue;
            if (want && (top != 0) != wtop) continue;
            cands.push_back(Cand{c, top != 0, adds_of(c, top != 0), msm_table_bytes(srs->n_points, kind, c, top != 0)});
        }
    std::sort(cands.begin(), cands.end(), [](const Cand& x, const Cand& y) { return x.adds != y.adds ? x.adds < y.adds : x.bytes < y.bytes; });
    const double have_adds = have ? adds_of(have->bits, have->top_g != 0) : 1e9;
    for (const Cand& k : cands) {
        if (k.adds >= have_adds) break;
        if (k.bytes + used > budget) continue;
        if (msm_table_build(ctx, srs, kind, k.c, k.top) == PLONK_OK) return true;
    }
    if (have) {
        lut_attach(srs, have);
        return true;
    }
    srs->lookup_failed = true;
    return false;
}

// Workgroups per MSM.  `g0` is what fills the chip; a launch, however, runs in ROUNDS of (CUs x 4) resident 256-thread
// workgroups, and a last round that is half empty leaves half the SIMD slots without a wave for the time of a whole round
// (M = 1536 MSMs at one workgroup each: 1.5 rounds on 1024 slots — two of the four MSM launches of a lock-step batch of 512 proofs).
// Cutting every MSM into twice the workgroups halves the length of a round for `overhead` more work per workgroup (its tree
// reduction / its extra pieces): taken when the model  rounds x (1 / G + overhead)  says it pays by more than 3 %.
// PLONK_MSM_ROUNDS=0 keeps g0 (A/B runs).  Measured (profiles/r05_f_msm_rounds_stagger_ab.json, r05_d_msm_sweep.jsonl): 1152 MSMs on the
// bucket method 5.59 -> 5.14 ms per call; the prover's own launch shapes gain under 1 % on either method.
static unsigned msm_round_aware_groups(int device, size_t M, unsigned g0, unsigned g_max, double overhead) {
    static const bool off = [] { const char* e = getenv("PLONK_MSM_ROUNDS"); return e && !strcmp(e, "0"); }();
    if (off || g0 >= g_max) return g0;
    static int cus[16] = {0};
    int& n_cu = cus[device & 15];
    if (!n_cu) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) n_cu = prop.multiProcessorCount;
        else {
            (void)hipGetLastError();
            n_cu = 256;
        }
    }
    const double slots = 4.0 * (double)n_cu;
    auto cost = [&](unsigned G) {
        const double wgs = (double)M * G;
        double rounds = wgs / slots;
        rounds = rounds <= 1.0 ? 1.0 : (double)(size_t)(rounds + 0.999999);
        return rounds * (1.0 / G + overhead);
    };
    return cost(2 * g0) < 0.97 * cost(g0) ? 2 * g0 : g0;
}

// digits kernel of the comb with h teeth (one instantiation per tooth count: the bit gather is unrolled at compile time); TOP: the
// comb with top tables, whose workgroups take whole groups of scalars and add the virtual scalars' digits (nd = n + their number)
template <unsigned H, bool TOP> static void msm_comb_launch_digits(plonk_ctx* ctx, const Fr* d_scalars, size_t n, size_t stride, size_t inner,
                                                                   size_t outer_stride, size_t M, uint32_t* digits, size_t nd) {
    if constexpr (!TOP || msm_comb_top_ok(H)) {
        constexpr MsmCombShape SH = msm_comb_shape(H, TOP);
        constexpr unsigned G = TOP ? SH.top_g : 1, PER = (256 / G) * G;
        size_t gx = (n + PER - 1) / PER;
        if (TOP) {  // every virtual scalar's digit is written, also those of the groups past the last scalar
            const size_t gv = ((nd - n) * SH.a + PER / G - 1) / (PER / G);
            gx = gx > gv ? gx : gv;
        }
        void (*kern)(const Fr*, size_t, size_t, size_t, size_t, size_t, uint32_t*, size_t) = msm_comb_digits_kernel<H, TOP>;  // (a template-id's comma would split the macro's arguments)
        for (size_t m0 = 0; m0 < M; m0 += 32768) {  // (a grid's second dimension ends at 65 535)
            const size_t rows = M - m0 < 32768 ? M - m0 : 32768;
            PLONK_LAUNCH(kern, dim3((unsigned)gx, (unsigned)rows), dim3(256), 0, ctx->stream, d_scalars, n, stride, inner,
                         outer_stride, m0, digits, nd);
        }
    }
}
typedef void (*msm_comb_digits_fn)(plonk_ctx*, const Fr*, size_t, size_t, size_t, size_t, size_t, uint32_t*, size_t);
template <bool TOP, unsigned... H> static msm_comb_digits_fn msm_comb_digits_for(unsigned h, std::integer_sequence<unsigned, H...>) {
    msm_comb_digits_fn fn = nullptr;
    ((h == H + 2 && (!TOP || msm_comb_top_ok(H + 2)) ? (void)(fn = &msm_comb_launch_digits<H + 2, TOP>) : (void)0), ...);
    return fn;
}

static int msm_run_comb(plonk_ctx* ctx, plonk_srs* srs, const Fr* d_scalars, size_t n_real, size_t M, size_t stride, Fq* d_out_xy, uint8_t* d_flags,
                        size_t inner, size_t outer_stride) {
    const unsigned h = srs->lookup_bits, a = srs->lookup_windows, hb = h - 1;
    // top tables (msm_comb.h): the kernels see n = n_real + nv scalars, the last nv of each column being its share of the groups
    const bool top = srs->lookup_top_g != 0;
    const MsmCombShape sh = msm_comb_shape(h, top);
    const size_t nv = msm_comb_virtual(n_real, sh), n = n_real + nv;
    const unsigned top_delta = (unsigned)(srs->n_points - n_real);  // block of virtual scalar n_real + v = n_points + v (+ the digit's offset)
    PLONK_REQUIRE((uint64_t)n * a < ((uint64_t)1 << 32), PLONK_ERR_ARG, "MSM size %zu too large for the lookup path", n_real);
    PLONK_REQUIRE(msm_comb_top_reach_ok(n_real, sh), PLONK_ERR_ARG, "MSM size %zu too large for the top tables of %u teeth", n_real, h);
    unsigned G = ctx->msm_groups;
    if (!G) {  // enough waves to occupy 1024 SIMDs three to four deep, in as few workgroups per MSM as that takes
        G = 1;
        while (G < 64 && M * G * (MSM_BLOCK / 64) < 3072) G *= 2;
    }
    if (!ctx->msm_groups) G = msm_round_aware_groups(ctx->device, M, G, 64, 0.035);
    while (G > 1 && (size_t)G * MSM_BLOCK * 2 > n * a) G /= 2;  // at least two additions per lane
    while (G > 1 && (size_t)G > n) G /= 2;
    const size_t part_bytes = (M * G * a * sizeof(G1Xyzz) + 255) & ~(size_t)255;
    const size_t col_bytes = G >= 4 ? (M * a * sizeof(G1Xyzz) + 255) & ~(size_t)255 : 0;
    const size_t cnt_bytes = (M * 4 + 255) & ~(size_t)255;
    const size_t dfr_bytes = M * MSM_DEFER_CAP * sizeof(MsmDeferred);
    const size_t dig_bytes = (M * a * n * 4 + 255) & ~(size_t)255;
    void* s;
    PLONK_TRY(ctx_scratch(ctx, 1, part_bytes + col_bytes + cnt_bytes + dfr_bytes + dig_bytes, &s));
    G1Xyzz* partial = (G1Xyzz*)s;
    G1Xyzz* colsum = (G1Xyzz*)((uint8_t*)s + part_bytes);
    uint32_t* n_deferred = (uint32_t*)((uint8_t*)s + part_bytes + col_bytes);
    MsmDeferred* deferred = (MsmDeferred*)((uint8_t*)s + part_bytes + col_bytes + cnt_bytes);
    uint32_t* digits = (uint32_t*)((uint8_t*)s + part_bytes + col_bytes + cnt_bytes + dfr_bytes);
    const auto teeth = std::make_integer_sequence<unsigned, MSM_COMB_MAX_TEETH - 1>();
    const msm_comb_digits_fn digits_fn = top ? msm_comb_digits_for<true>(h, teeth) : msm_comb_digits_for<false>(h, teeth);
    PLONK_REQUIRE(digits_fn, PLONK_ERR_ARG, "no comb of %u teeth", h);
    PLONK_CHECK_HIP(hipMemsetAsync(n_deferred, 0, M * 4, ctx->stream));
    PLONK_TRY(prof_begin(ctx, "msm_digits", (double)M * (double)n * (32.0 + 4.0 * a)));
    digits_fn(ctx, d_scalars, n_real, stride, inner, outer_stride, M, digits, n);
    PLONK_TRY(prof_end(ctx));
    const size_t lds = (size_t)(MSM_BLOCK + a) * sizeof(G1Xyzz);
    PLONK_TRY(prof_begin(ctx, "msm_comb", (double)M * (96.0 * (double)n_real + 64.0)));
    PLONK_LAUNCH(msm_comb_kernel, dim3((unsigned)(M * G)), dim3(MSM_BLOCK), lds, ctx->stream, (const G1Affine*)srs->lookup, hb, a,
                 (const uint32_t*)digits, n, G, partial, deferred, (size_t)MSM_DEFER_CAP, n_deferred, (unsigned)n_real, top_delta);
    PLONK_TRY(prof_end(ctx));
    const G1Xyzz* sums = partial;
    unsigned Gf = G;
    if (G >= 4) {  // few MSMs in many pieces: a wave per (MSM, column) sums the pieces in parallel
        PLONK_LAUNCH(msm_comb_colsum_kernel, dim3((unsigned)(M * a)), dim3(64), 0, ctx->stream, (const G1Xyzz*)partial, G, a, colsum, n_deferred);
        sums = colsum;
        Gf = 1;
    }
    // lanes per MSM in the Horner step (chain of a - 1 doublings and the additions between them, on lazy limbs): four for a batch
    // (12 doublings + 6 additions per lane, 16 MSMs per wave), sixteen when the MSMs are few (12 + 4: the shortest chain);
    // one lane per MSM would be the least work and the longest chain (12 + 12).  PLONK_MSM_COMB_LPM = 1 / 4 / 16 forces one (A/B).
    static const unsigned forced_lpm = [] { const char* e = getenv("PLONK_MSM_COMB_LPM"); return e ? (unsigned)atoi(e) : 0u; }();
    const unsigned lpm = forced_lpm ? forced_lpm : (M >= 64 ? 4u : 16u);
#define PLONK_COMB_FINALIZE(L)                                                                                                                  \
    PLONK_LAUNCH(msm_comb_finalize_kernel<L>, dim3((unsigned)((M * L + 63) / 64)), dim3(64), 0, ctx->stream, sums, M, Gf, a,                      \
                 (const G1Affine*)srs->lookup, hb, n, (const MsmDeferred*)deferred, (size_t)MSM_DEFER_CAP, n_deferred, d_out_xy, d_flags,         \
                 (unsigned)n_real, top_delta)
    if (lpm >= 16) PLONK_COMB_FINALIZE(16);
    else if (lpm >= 4) PLONK_COMB_FINALIZE(4);
    else PLONK_COMB_FINALIZE(1);
#undef PLONK_COMB_FINALIZE
    PLONK_LAUNCH(msm_comb_slow_kernel, dim3((unsigned)M), dim3(256), 0, ctx->stream, (const G1Affine*)srs->lookup, hb, a, (const uint32_t*)digits, n,
                 (const uint32_t*)n_deferred, d_out_xy, d_flags, (unsigned)n_real, top_delta);
    PLONK_CHECK_HIP(hipGetLastError());
    return PLONK_OK;
}

static int msm_run_lookup(plonk_ctx* ctx, plonk_srs* srs, const Fr* d_scalars, size_t n, size_t M, size_t stride, Fq* d_out_xy,
                          uint8_t* d_flags, size_t inner, size_t outer_stride) {
    if (srs->lookup_kind == MSM_TABLE_COMB) return msm_run_comb(ctx, srs, d_scalars, n, M, stride, d_out_xy, d_flags, inner, outer_stride);
    const unsigned c = srs->lookup_bits, W = srs->lookup_windows;
    const size_t items = n * W;
    PLONK_REQUIRE(items < ((size_t)1 << 32), PLONK_ERR_ARG, "MSM size %zu too large for the lookup path", n);
    unsigned G = ctx->msm_groups;
    if (!G) {  // enough waves to occupy 1024 SIMDs three to four deep, in as few workgroups per MSM as that takes
        G = 1;
        while (G < 64 && M * G * (MSM_BLOCK / 64) < 3072) G *= 2;
    }
    if (!ctx->msm_groups) G = msm_round_aware_groups(ctx->device, M, G, 64, 0.027);
    while (G > 1 && (size_t)G * MSM_BLOCK * 2 > items) G /= 2;  // at least two additions per lane
    const size_t part_bytes = (M * G * sizeof(G1Xyzz) + 255) & ~(size_t)255;
    const size_t cnt_bytes = (M * 4 + 255) & ~(size_t)255;
    const size_t dfr_bytes = M * MSM_DEFER_CAP * sizeof(MsmDeferred);
    void* s;
    PLONK_TRY(ctx_scratch(ctx, 1, part_bytes + cnt_bytes + dfr_bytes, &s));
    G1Xyzz* partial = (G1Xyzz*)s;
    uint32_t* n_deferred = (uint32_t*)((uint8_t*)s + part_bytes);
    MsmDeferred* deferred = (MsmDeferred*)((uint8_t*)s + part_bytes + cnt_bytes);
    MsmRecode rc;
    msm_recode_constant(c, W, &rc);
    // lanes walk scalars t, t + lanes, .. (window after window) whenever every lane gets a scalar: the chip then reads one
    // contiguous 1 - 2 GB of the table at a time, which the translation caches hold (see msm_lookup_kernel).  PLONK_MSM_ORDER=flat
    // restores round 4's order (each lane 8 consecutive scalars) for A/B runs.
    static const bool force_flat = [] { const char* e = getenv("PLONK_MSM_ORDER"); return e && !strcmp(e, "flat"); }();
    const unsigned strided = (!force_flat && n >= (size_t)G * MSM_BLOCK) ? 1u : 0u;
    PLONK_CHECK_HIP(hipMemsetAsync(n_deferred, 0, M * 4, ctx->stream));
    PLONK_TRY(prof_begin(ctx, "msm_lookup", (double)M * (96.0 * (double)n + 64.0)));
    PLONK_LAUNCH(msm_lookup_kernel, dim3((unsigned)(M * G)), dim3(MSM_BLOCK), (size_t)MSM_BLOCK * sizeof(G1Xyzz), ctx->stream,
                 (const G1Affine*)srs->lookup, srs->n_points, c, W, d_scalars, n, stride, inner, outer_stride, rc, G, partial,
                 deferred, (size_t)MSM_DEFER_CAP, n_deferred, strided);
    PLONK_TRY(prof_end(ctx));
    if (G >= 8)  // few MSMs in many pieces: a wave per MSM sums the pieces in parallel
        PLONK_LAUNCH(msm_lookup_finalize_wave_kernel, dim3((unsigned)M), dim3(64), 0, ctx->stream, (const G1Xyzz*)partial, M, G,
                     (const G1Affine*)srs->lookup, srs->n_points, c, W, (const MsmDeferred*)deferred, (size_t)MSM_DEFER_CAP,
                     (const uint32_t*)n_deferred, d_out_xy, d_flags);
    else
        PLONK_LAUNCH(msm_lookup_finalize_kernel, dim3((unsigned)((M + 63) / 64)), dim3(64), 0, ctx->stream, (const G1Xyzz*)partial, M, G,
                     (const G1Affine*)srs->lookup, srs->n_points, c, W, (const MsmDeferred*)deferred, (size_t)MSM_DEFER_CAP,
                     (const uint32_t*)n_deferred, d_out_xy, d_flags);
    PLONK_LAUNCH(msm_slow_kernel, dim3((unsigned)M), dim3(256), 0, ctx->stream, 0, (const G1Affine*)srs->lookup, srs->n_points, c, W,
                 d_scalars, n, stride, inner, outer_stride, rc, (const uint32_t*)n_deferred, d_out_xy, d_flags);
    PLONK_CHECK_HIP(hipGetLastError());
    return PLONK_OK;
}

// ------------------------------------------------------------------------------------------------
// Lagrange-basis SRS (setup.py:66-72 says commit = ifft + lincomb with powers_of_x): the points
//     [L_i(tau)]_1 = sum_j (w^(-ij) / n) [tau^j]_1,      i < n = 2^log_n,
// i.e. the inverse DFT of the SRS over the group, computed once per (SRS, n) — as n batched MSMs over the
// monomial bases whose scalar rows are the inverse NTT of the identity matrix (row i = coefficients of L_i), so the
// "EC-iNTT" reuses the NTT and MSM kernels as they are (2^11 rows: a few milliseconds on the lookup table).
// commit(values) = sum_i values_i [L_i(tau)]_1 is then one MSM with no inverse NTT in front of it.
__global__ void fr_identity_rows_kernel(Fr* out, size_t n, size_t row0, size_t rows) {
    const size_t total = rows * n;
    const Fr one = fp_one<FrParams>(), zero = fp_zero<FrParams>();
    for (size_t gI = (size_t)blockIdx.x * blockDim.x + threadIdx.x; gI < total; gI += (size_t)gridDim.x * blockDim.x) {
        const size_t r = gI / n, j = gI - r * n;
        fp_store(out + gI, j == row0 + r ? one : zero);
    }
}
__global__ void fq_to_mont_kernel(const Fq* in, Fq* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        fp_store(out + i, fp_to_mont(fp_load(in + i)));
}

int msm_lagrange_srs(plonk_ctx* ctx, plonk_srs* srs, unsigned log_n, plonk_srs** out) {
    auto it = srs->lagrange.find(log_n);
    if (it != srs->lagrange.end()) {
        *out = it->second;
        return PLONK_OK;
    }
    const size_t n = (size_t)1 << log_n;
    PLONK_REQUIRE(n <= srs->n_points, PLONK_ERR_ARG, "Lagrange basis of size %zu needs %zu powers, the SRS has %zu", n, n, srs->n_points);
    PLONK_REQUIRE(log_n <= PLONK_FR_TWO_ADICITY, PLONK_ERR_ARG, "size 2^%u exceeds the 2-adicity of Fr", log_n);
    // Two routes to the same points (bit-identical: both end in the unique affine representative).  Up to 2^12: n MSMs over the
    // monomial bases (below) — a few milliseconds once the lookup table exists, but quadratic.  Above: the inverse DFT over the
    // group (g1_ntt.hip), n log n group operations.  PLONK_LAGRANGE_SRS = msm / ntt forces one (tests, A/B).
    {
        const char* e = getenv("PLONK_LAGRANGE_SRS");
        const bool by_ntt = e ? !strcmp(e, "ntt") : log_n > 12;
        if (by_ntt) {
            G1Affine* nb = nullptr;
            if (!plonk_dev_malloc(&nb, n * sizeof(G1Affine))) {
                plonk_set_error("hipMalloc failed while building the Lagrange-basis SRS of size %zu", n);
                return PLONK_ERR_NOMEM;
            }
            const int rcn = g1_lagrange_by_ntt(ctx, srs, log_n, nb);
            if (rcn != PLONK_OK) {
                hipFree(nb);
                return rcn;
            }
            plonk_srs* child = new plonk_srs();
            child->device = srs->device;
            child->n_points = n;
            child->bases = nb;
            child->fixed = srs->fixed;
            child->parent = srs;
            const uint64_t tag[2] = {srs->content_key, 0x4c61677200000000ull | log_n};  // "Lagr" | log_n
            child->content_key = plonk_fnv1a64(tag, sizeof tag);
            srs->lagrange[log_n] = child;
            *out = child;
            return PLONK_OK;
        }
    }
    const size_t rows = n < 1024 ? n : 1024;  // rows per round: bounds the scalar matrix at 1024 * n elements
    void *mat = nullptr, *res = nullptr, *bases = nullptr;
    auto cleanup = [&]() {
        if (mat) hipFree(mat);
        if (res) hipFree(res);
    };
    if (!plonk_dev_malloc(&mat, rows * n * sizeof(Fr)) || !plonk_dev_malloc(&res, rows * (2 * sizeof(Fq) + 1) + 64) ||
        !plonk_dev_malloc(&bases, n * sizeof(G1Affine))) {
        cleanup();
        if (bases) hipFree(bases);
        plonk_set_error("hipMalloc failed while building the Lagrange-basis SRS of size %zu", n);
        return PLONK_ERR_NOMEM;
    }
    Fq* d_xy = (Fq*)res;
    uint8_t* d_fl = (uint8_t*)res + rows * 2 * sizeof(Fq);
    int rc = PLONK_OK;
    for (size_t row0 = 0; row0 < n && rc == PLONK_OK; row0 += rows) {
        unsigned g = (unsigned)((rows * n + 255) / 256);
        PLONK_LAUNCH(fr_identity_rows_kernel, dim3(g > 4096 ? 4096 : g), dim3(256), 0, ctx->stream, (Fr*)mat, n, row0, rows);
        rc = ntt_run(ctx, (const Fr*)mat, (Fr*)mat, log_n, true, rows, n, n, n, nullptr, nullptr, true);
        if (rc == PLONK_OK) rc = msm_run_device(ctx, srs, (const Fr*)mat, n, rows, n, d_xy, d_fl);
        if (rc == PLONK_OK) {  // canonical x||y -> Montgomery bases; the identity stays (0, 0)
            unsigned g2 = (unsigned)((2 * rows + 255) / 256);
            PLONK_LAUNCH(fq_to_mont_kernel, dim3(g2), dim3(256), 0, ctx->stream, (const Fq*)d_xy, (Fq*)((G1Affine*)bases + row0), 2 * rows);
        }
    }
    if (rc == PLONK_OK && (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)) {
        plonk_set_error("building the Lagrange-basis SRS failed on the device");
        rc = PLONK_ERR_HIP;
    }
    cleanup();
    if (rc != PLONK_OK) {
        hipFree(bases);
        return rc;
    }
    plonk_srs* child = new plonk_srs();
    child->device = srs->device;
    child->n_points = n;
    child->bases = (G1Affine*)bases;
    child->fixed = srs->fixed;
    child->parent = srs;
    const uint64_t tag[2] = {srs->content_key, 0x4c61677200000000ull | log_n};  // "Lagr" | log_n
    child->content_key = plonk_fnv1a64(tag, sizeof tag);
    srs->lagrange[log_n] = child;
    *out = child;
    return PLONK_OK;
}

// Enqueue a batch of M MSMs; results land in device buffers (d_out_xy: 2*M Fq canonical, d_flags: M bytes).
int msm_run_device(plonk_ctx* ctx, plonk_srs* srs, const Fr* d_scalars, size_t n, size_t M, size_t stride,
                   Fq* d_out_xy, uint8_t* d_flags, size_t inner, size_t outer_stride) {
    if (!inner) inner = M ? M : 1;
    PLONK_REQUIRE(n >= 1 && n <= srs->n_points, PLONK_ERR_ARG, "MSM size %zu exceeds the %zu loaded bases", n, srs->n_points);
    if (!M) return PLONK_OK;
    if (msm_lookup_prepare(ctx, srs)) return msm_run_lookup(ctx, srs, d_scalars, n, M, stride, d_out_xy, d_flags, inner, outer_stride);
    PLONK_REQUIRE(n <= 32768, PLONK_ERR_ARG, "MSM size %zu > 32768 is not supported by the bucket method's entry encoding", n);
    unsigned c = ctx->msm_window_bits ? ctx->msm_window_bits : MSM_DEFAULT_WINDOW_BITS;
    PLONK_TRY(msm_build_table(ctx, srs, c));
    const unsigned W = srs->n_windows, K = 1u << (c - 1);
    unsigned G = ctx->msm_groups;
    if (!G) {
        // enough workgroups to fill 256 CUs a few times over, but no more pieces than needed
        G = 1;
        while (G < 16 && M * G < 1024) G *= 2;
    }
    const size_t max_entries = (size_t)W * n;
    if (!ctx->msm_groups) G = msm_round_aware_groups(ctx->device, M, G, 16, 0.03);
    while (G > 1 && (size_t)G * MSM_BLOCK * 4 > max_entries) G /= 2;  // tiny MSMs: one segment is plenty
    // lanes per MSM in the bucket reduction (shorter local walks vs more lanes paying the scan and the reduction): batches (G < 8)
    // take 128 — measured best with the suffix-scan weighting (profiles/r04_c_bucket_reduce_lanes_ab.jsonl: 23.3 k proofs/s against
    // 22.5 k at 64; round 3's double-and-add weighting: 22.8 k at 128) — a lone MSM cut into many workgroups (G >= 8) is latency
    // bound and takes 256.  PLONK_MSM_RED_LANES = 64 / 128 / 256 overrides (A/B runs).
    unsigned red_lanes = G >= 8 ? 256 : 128;
    {
        static const unsigned forced = [] {
            const char* e = getenv("PLONK_MSM_RED_LANES");
            const unsigned v = e ? (unsigned)atoi(e) : 0u;
            return (v == 64 || v == 128 || v == 256) ? v : 0u;
        }();
        if (forced) red_lanes = forced;
    }
    // msm_bucket_reduce_kernel weighs a lane's run by pb = K / red_lanes through log2(pb) doublings: both must be powers of two
    PLONK_REQUIRE((red_lanes & (red_lanes - 1)) == 0 && red_lanes >= 64 && (K % red_lanes == 0 || K < red_lanes), PLONK_ERR_ARG,
                  "bucket reduction needs a power-of-two lane count dividing the %u buckets (got %u)", K, red_lanes);
    const size_t entry_stride = ((max_entries + 3) & ~(size_t)3) + 4;
    const size_t piece_stride = (size_t)G * MSM_BLOCK + K;
    const size_t ent_bytes = (M * entry_stride * 4 + 255) & ~(size_t)255;
    const size_t st_bytes = (M * (size_t)(K + 2) * 4 + 255) & ~(size_t)255;
    const size_t piece_bytes = (M * piece_stride * sizeof(G1Xyzz) + 255) & ~(size_t)255;
    const size_t cnt_bytes = (M * 4 + 255) & ~(size_t)255;
    const size_t dfr_bytes = M * MSM_DEFER_CAP * sizeof(MsmDeferred);  // bounded: an MSM that overflows is redone by msm_slow_kernel
    void* s;
    PLONK_TRY(ctx_scratch(ctx, 1, ent_bytes + st_bytes + piece_bytes + cnt_bytes + dfr_bytes, &s));
    uint32_t* entries = (uint32_t*)s;
    uint32_t* starts = (uint32_t*)((uint8_t*)s + ent_bytes);
    G1Xyzz* pieces = (G1Xyzz*)((uint8_t*)s + ent_bytes + st_bytes);
    uint32_t* n_deferred = (uint32_t*)((uint8_t*)s + ent_bytes + st_bytes + piece_bytes);
    MsmDeferred* deferred = (MsmDeferred*)((uint8_t*)s + ent_bytes + st_bytes + piece_bytes + cnt_bytes);

    MsmRecode rc;
    msm_recode_constant(c, W, &rc);
    const size_t sort_lds = (size_t)(K + 2) * 4;
    if (!ctx->msm_attr_set) {  // a per-device attribute: tracked per context, not per process
        PLONK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(msm_sort_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(128 * 1024)));
        ctx->msm_attr_set = true;
    }
    PLONK_TRY(prof_begin(ctx, "msm_sort", (double)M * 32.0 * (double)n));
    PLONK_LAUNCH(msm_sort_kernel, dim3((unsigned)M), dim3(MSM_BLOCK), sort_lds, ctx->stream, d_scalars, n, stride, inner, outer_stride, c, W, rc,
                 entries, entry_stride, starts, n_deferred);
    PLONK_TRY(prof_end(ctx));
    // algorithmic bytes of an MSM of size n: (64 + 32) * n + 64   
    PLONK_TRY(prof_begin(ctx, "msm_accumulate", (double)M * (96.0 * (double)n + 64.0)));
    PLONK_LAUNCH(msm_accumulate_kernel, dim3((unsigned)(M * G)), dim3(MSM_BLOCK), sort_lds, ctx->stream,
                 (const G1Affine*)srs->table, srs->n_points, (const uint32_t*)entries, entry_stride,
                 (const uint32_t*)starts, c, G, pieces, piece_stride, deferred, n_deferred);
    PLONK_TRY(prof_end(ctx));
    PLONK_TRY(prof_begin(ctx, "msm_bucket_reduce", (double)M * (double)piece_stride * sizeof(G1Xyzz)));
    PLONK_LAUNCH(msm_bucket_reduce_kernel, dim3((unsigned)M), dim3(red_lanes), (size_t)red_lanes * sizeof(G1Xyzz), ctx->stream,
                 (const uint32_t*)starts, c, G * MSM_BLOCK, (const G1Xyzz*)pieces, piece_stride,
                 (const G1Affine*)srs->table, srs->n_points, (const MsmDeferred*)deferred, (size_t)MSM_DEFER_CAP, (const uint32_t*)n_deferred,
                 d_out_xy, d_flags);
    PLONK_TRY(prof_end(ctx));
    PLONK_LAUNCH(msm_slow_kernel, dim3((unsigned)M), dim3(256), 0, ctx->stream, 1, (const G1Affine*)srs->table, srs->n_points, c, W,
                 d_scalars, n, stride, inner, outer_stride, rc, (const uint32_t*)n_deferred, d_out_xy, d_flags);
    PLONK_CHECK_HIP(hipGetLastError());
    return PLONK_OK;
}
