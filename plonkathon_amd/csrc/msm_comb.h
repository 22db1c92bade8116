// msm_comb.h — fixed-base MSM on COMB tables (round 5).  Included by msm.hip after its common definitions.
//
// Same job as the window tables of msm.hip (ec_lincomb over a reusable SRS, /root/reference/curve.py:38-111 and
// setup.py:66-72), fewer additions per base for the same memory.  With h teeth spaced a = ceil(254 / h) bits apart,
// a scalar s is cut into a columns of h bits:  s = sum_{j < a} 2^j  sum_{k < h} b_(j + a k) 2^(a k),  and
//     s P = sum_j 2^j  E_P[ bits j, j + a, j + 2a, .. ]      with ONE table per base,  E_P[idx] = sum_k (+-) 2^(a k) P.
// The a - 1 doublings of the outer sum are shared by all the bases of an MSM (they are done once per MSM on the a column sums),
// so an MSM of N bases costs N * a additions — against N * ceil(255 / c) for the window tables, whose table of the same size
// holds ceil(255 / c) windows of c bits where this one holds a single "window" of h bits:
//     2^11 bases     68.7 GB: h = 20, 13 additions per base   (windows: c = 16, 16 additions; c = 17 needs 128.8 GB for 15)
//                     8.6 GB: h = 17, 15 additions            (windows: c = 12/13, 22/20 additions)
//                     67 MB:  h = 10, 26 additions            (what the bucket method does with a sort and a bucket reduction)
// Signed digits without a zero digit: the scalar is made odd (s even: r - s, and the base's sign flips; r is odd), then written
// with all digits +-1:  s = sum_j b_j 2^j, b_j = 2 t_j - 1, t = (s + 2^L - 1) / 2, L = a h.  A column whose top tooth is -1 is
// the negated entry of the complemented lower teeth, so the table has 2^(h-1) entries per base,
//     E_P[idx] = 2^(a (h-1)) P + sum_{k < h-1} (idx_k ? + : -) 2^(a k) P,
// and EVERY (scalar, column) item is exactly one mixed addition: no zero digits, no branches in the loop.
// The table is built over P'_i = R^-1 P_i (R = 2^261, the Montgomery radix of Fr): the scalars reach the MSM as Montgomery residues
// s R mod r, and (s R) (R^-1 P) = s P — the digit kernel takes the residue as it is, sparing a conversion (a quarter of its work).
//
// With TOP TABLES (round 6, below: "combs with top tables") the comb has floor(254 / h) columns and the one or two bits they leave over
// go through joint tables of g bases each: 21 teeth, 12 columns, g = 7 — 12.15 additions per base of 2^11 from 157.6 GB.
//
// Kernels (one launch each per batch of MSMs):
//   msm_comb_digits_kernel<H, TOP>  one lane per scalar: the residue as stored, parity fold, t, and the a column indices (sign in bit 31)
//                               written column-major (digits[m][j][i], 4 B per item) — read back coalesced by the next kernel;
//                               TOP: a workgroup takes whole groups of g scalars and adds the digits of the virtual scalars
//   msm_comb_kernel             one workgroup per (MSM, scalar sub-range).  Lanes are bound to COLUMNS (a lane's accumulator can
//                               only hold one column's sum): q = 256 / a lanes per column walk scalars r, r + q, .. — the whole
//                               chip reads the tables of q consecutive bases at a time (address translation: msm.hip,
//                               msm_lookup_kernel) — for p = ceil(a n / 256) steps, and the 256 - a q lanes left over take the
//                               scalars the columns' lanes did not reach, column after column, leaving one partial sum
//                               ("piece") in LDS per column they touch.  Every lane performs p additions (+-0): 104 for 2^11
//                               scalars in 13 columns.  The pieces are tree-reduced per column through LDS; the workgroup
//                               stores a column sums.
//   msm_comb_colsum_kernel      (only when an MSM is cut into >= 4 workgroups) a wave per (MSM, column) sums the workgroups' sums
//   msm_comb_finalize_kernel    sum_j 2^j S_j by Horner over groups of LPM lanes per MSM, deferred additions, unique affine form; an
//                               addition of OPPOSITE operands gives the identity here (an MSM whose result is the identity — all
//                               scalars zero — cancels in these last additions, not before)
//   msm_comb_slow_kernel        the recovery path of MSM_DEFER_CAP (general formulas throughout)
// Table build: msm_comb_scale_kernel gives P'_i; msm_table_kernel G_k = 2^(a k) P'_i; msm_comb_fill_kernel walks each run of 2^8 consecutive indices in
// Gray-code order (one mixed addition of +-2 G_k per entry); g1_batch_to_affine_kernel converts a chunk of bases at a time;
// msm_comb_top_base_kernel / msm_comb_top_fill_kernel: the joint tables, one block of 2^(h-1) entries per group behind the bases' blocks.
#pragma once

#define MSM_COMB_MAX_TEETH 24
#define MSM_COMB_SEG_BITS 8
#define MSM_COMB_SCALAR_BITS 254   // r < 2^254
// n_deferred[m]: the number of deferred additions of MSM m in the low bits (the list holds the first MSM_DEFER_CAP of them), and this
// flag when a tree / Horner addition met equal or opposite operands.  Either way past the cap: msm_comb_slow_kernel redoes the MSM.
#define MSM_COMB_REDO 0x80000000u
PLONK_HD bool msm_comb_needs_redo(uint32_t v) { return (v & MSM_COMB_REDO) != 0 || v > MSM_DEFER_CAP; }

PLONK_HD unsigned msm_comb_columns(unsigned h) { return (MSM_COMB_SCALAR_BITS + h - 1) / h; }

// ---- combs with TOP TABLES (round 6) -----------------------------------------------------------
// ceil(254 / h) rounds up: 20 teeth need 13 columns (260 positions for 254 bits), and the next count of columns, 12, needs 22
// teeth — 275 GB for 2^11 bases.  21 teeth x 12 columns cover 252 bits; the two bits left over are worth one addition per base
// if they get a column of their own, but only 1 / g of one when g bases share a joint table for them:
//     s' = v + 2^L u,  L = a h,  v = s' mod 2^L (odd: all digits +-1 as before),  u = s' >> L in [0, 2^R),  R = 254 - L  (1 or 2)
//     sum_i s'_i P_i = (the comb over the v_i)  +  sum_groups  T_grp[ codes of its g bases ],
//     T_grp[idx] = sum_pos code_pos 2^(L - j) P'_(g grp + pos),   code = +-u in [-(2^R - 1), 2^R - 1]  (B = 2^(R+1) - 1 values),
//     idx = sum_pos (code_pos + (B - 1) / 2) B^pos  <  B^g <= 2^(h-1).
// Group grp is dealt to column j = grp mod a (hence the 2^(L - j): the Horner step multiplies column j's sum by 2^j), as the
// "virtual scalar" n + grp / a of that column: msm_comb_kernel sees an MSM of n + ceil(ceil(n / g) / a) scalars and runs
// unchanged — its table simply has one more 2^(h-1)-entry block per group behind the blocks of the N bases (block N + grp), and
// a virtual scalar's digit carries the block offset (v (a - 1) + j) beside idx: scalar n + v starts from block N + v (the kernels add
// top_delta = N - n to the index of a virtual scalar, msm_comb_block_of), and N + v + v (a - 1) + j = N + grp.  2^11 bases, h = 21: 12 columns of 2 073
// (virtual) scalars = 12.15 additions per base from 137.4 + 20.1 GB, against 13 from 68.7 GB.
struct MsmCombShape {
    unsigned h, a;                // teeth, columns
    unsigned top_bits, top_g;     // R and g (0, 0: no top tables — a = ceil(254 / h))
    unsigned top_b;               // B = 2^(R+1) - 1
    uint32_t top_entries;         // B^g
};
PLONK_HD constexpr MsmCombShape msm_comb_shape(unsigned h, bool top) {
    MsmCombShape s{h, (MSM_COMB_SCALAR_BITS + h - 1) / h, 0, 0, 0, 0};
    if (!top) return s;
    s.a = MSM_COMB_SCALAR_BITS / h;
    s.top_bits = MSM_COMB_SCALAR_BITS - s.a * h;
    s.top_b = (2u << s.top_bits) - 1;
    s.top_entries = 1;
    while (s.top_bits && (uint64_t)s.top_entries * s.top_b <= ((uint64_t)1 << (h - 1))) {
        s.top_entries *= s.top_b;
        s.top_g++;
    }
    return s;
}
// a comb of h teeth can take top tables: one or two bits left over, and a block holds the joint table of at least one base
PLONK_HD constexpr bool msm_comb_top_ok(unsigned h) {
    const MsmCombShape s = msm_comb_shape(h, true);
    return h >= 2 && s.a >= 1 && s.top_bits >= 1 && s.top_bits <= 2 && s.top_g >= 1;
}
PLONK_HD size_t msm_comb_top_groups(size_t n, unsigned g) { return g ? (n + g - 1) / g : 0; }
// virtual scalars per column of an MSM of n scalars (= virtual blocks per column of a table of n bases)
PLONK_HD size_t msm_comb_virtual(size_t n, const MsmCombShape& s) { return s.top_g ? (msm_comb_top_groups(n, s.top_g) + s.a - 1) / s.a : 0; }
// blocks of 2^(h-1) entries in the table of n bases
PLONK_HD size_t msm_comb_blocks(size_t n, const MsmCombShape& s) { return n + msm_comb_virtual(n, s) * s.a; }
// a virtual scalar's digit carries its block offset (v (a - 1) + j) beside the index: 31 bits in all
PLONK_HD bool msm_comb_top_reach_ok(size_t n, const MsmCombShape& s) {
    return !s.top_g || ((uint64_t)(msm_comb_virtual(n, s) * (s.a - 1) + s.a) << (s.h - 1)) < ((uint64_t)1 << 31);
}
// block of the table an item of scalar i (real: i < n_real; virtual: the others) starts from
PLONK_HD size_t msm_comb_block_of(size_t i, size_t n_real, size_t top_delta) { return i + (i >= n_real ? top_delta : 0); }

// c P by double-and-add over the bits of a wave-uniform constant (c = R^-1 mod r, 254 bits): one-off, per base
struct MsmCombScale { uint32_t c[8]; };
PLONK_DEV G1Xyzz msm_comb_scaled_base(const G1Affine& P, const MsmCombScale& k) {
    G1Xyzz acc = g1_xyzz_identity();
#pragma unroll 1
    for (int bit = 255; bit >= 0; bit--) {
        g1_dbl(acc);
        if ((k.c[bit >> 5] >> (bit & 31)) & 1u) g1_madd<true>(acc, P);
    }
    return acc;
}
__global__ void __launch_bounds__(64) msm_comb_scale_kernel(const G1Affine* bases, size_t n, MsmCombScale k, G1Xyzz* out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        G1Affine b;
        b.x = fp_load(&bases[i].x);
        b.y = fp_load(&bases[i].y);
        out[i] = msm_comb_scaled_base(b, k);
    }
}

// E_P[idx] by the definition (signed Horner from the top tooth): used by the verification of a shared table only
PLONK_DEV G1Xyzz msm_comb_entry_slow(const G1Affine& P, unsigned a, unsigned h, uint32_t idx) {
    G1Xyzz acc = g1_xyzz_from_affine(P);
    const G1Affine N = g1_affine_is_identity(P) ? P : g1_affine_neg(P);
#pragma unroll 1
    for (int k = (int)h - 2; k >= 0; k--) {
#pragma unroll 1
        for (unsigned d = 0; d < a; d++) g1_dbl(acc);
        g1_madd<true>(acc, ((idx >> k) & 1) ? P : N);
    }
    return acc;
}

// out[k * n + i] = 2 * cb[k * n + i]  (the Gray-code steps of the fill), XYZZ
__global__ void msm_comb_delta_kernel(const G1Affine* cb, size_t count, G1Xyzz* out) {
    for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < count; t += (size_t)gridDim.x * blockDim.x) {
        G1Affine b;
        b.x = fp_load(&cb[t].x);
        b.y = fp_load(&cb[t].y);
        out[t] = g1_affine_is_identity(b) ? g1_xyzz_identity() : g1_dbl_affine(b);
    }
}

// tmp[(i - i0) << (h-1) | idx] = E_{P_i}[idx] for the bases i0 .. i0 + nb - 1, XYZZ.  cb[k n + i] = 2^(a k) P_i (k < h),
// cd[k n + i] = 2^(a k + 1) P_i (k < sb), both affine.  A lane fills a run of 2^sb consecutive indices: the first entry from the
// h tooth points, the others in Gray-code order, each one mixed addition of +-2 G_k away from the previous one.
__global__ void __launch_bounds__(64) msm_comb_fill_kernel(const G1Affine* cb, const G1Affine* cd, size_t n, size_t i0, size_t nb, unsigned h,
                                                           unsigned sb, G1Xyzz* tmp) {
    const size_t nseg = (size_t)1 << (h - 1 - sb), seg_len = (size_t)1 << sb;
    for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < nb * nseg; id += (size_t)gridDim.x * blockDim.x) {
        const size_t bi = id / nseg, i = i0 + bi, idx0 = (id % nseg) << sb;
        G1Affine t;
        t.x = fp_load(&cb[(size_t)(h - 1) * n + i].x);
        t.y = fp_load(&cb[(size_t)(h - 1) * n + i].y);
        G1Xyzz acc = g1_xyzz_from_affine(t);
#pragma unroll 1
        for (unsigned k = 0; k + 1 < h; k++) {
            t.x = fp_load(&cb[(size_t)k * n + i].x);
            t.y = fp_load(&cb[(size_t)k * n + i].y);
            if (!((idx0 >> k) & 1)) t.y = fp_neg(t.y);
            g1_madd<true>(acc, t);
        }
        G1Xyzz* out = tmp + (bi << (h - 1)) + idx0;
        out[0] = acc;
#pragma unroll 1
        for (uint32_t s = 1; s < seg_len; s++) {
            const unsigned b = (unsigned)__builtin_ctz(s);
            const uint32_t gray = s ^ (s >> 1);
            t.x = fp_load(&cd[(size_t)b * n + i].x);
            t.y = fp_load(&cd[(size_t)b * n + i].y);
            if (!((gray >> b) & 1)) t.y = fp_neg(t.y);
            g1_madd<true>(acc, t);
            out[gray] = acc;
        }
    }
}

// ---- top tables: build ----
// out[i] = 2^(L - j) P'_i, j = (i / g) mod a: the tooth point of base i in the joint table of its group (XYZZ)
__global__ void __launch_bounds__(64) msm_comb_top_base_kernel(const G1Affine* pb, size_t n, unsigned L, unsigned a, unsigned g, G1Xyzz* out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        G1Affine b;
        b.x = fp_load(&pb[i].x);
        b.y = fp_load(&pb[i].y);
        G1Xyzz p = g1_xyzz_from_affine(b);
        const unsigned j = (unsigned)((i / g) % a);
#pragma unroll 1
        for (unsigned k = 0; k < L - j; k++) g1_dbl(p);
        out[i] = p;
    }
}
PLONK_DEV G1Affine msm_comb_top_point(const G1Affine* tq, size_t n, size_t i, bool neg) {
    G1Affine q = g1_affine_identity();
    if (i < n) {
        q.x = fp_load(&tq[i].x);
        q.y = fp_load(&tq[i].y);
        if (neg && !g1_affine_is_identity(q)) q.y = fp_neg(q.y);
    }
    return q;
}
// tmp[(grp - g0) << hb | idx] = T_grp[idx], idx < B^g, for the groups g0 .. g0 + ng - 1 (XYZZ; the caller zeroes tmp: the entries
// from B^g up stay the identity).  tq: the n tooth points, affine.  A lane fills a run of B^2 consecutive indices (B when a
// group is one base): the higher digits' share first, then the two low digits from (-c, -c) upwards in boustrophedon order, so that
// every entry is one mixed addition of +-Q away from the one before.
__global__ void __launch_bounds__(64) msm_comb_top_fill_kernel(const G1Affine* tq, size_t n, size_t g0, size_t ng, unsigned hb, unsigned g, unsigned B,
                                                               uint32_t entries, G1Xyzz* tmp) {
    const unsigned low = g >= 2 ? 2u : 1u, run = low == 2 ? B * B : B, c = (B - 1) / 2;
    const size_t runs = entries / run;
    for (size_t id = (size_t)blockIdx.x * blockDim.x + threadIdx.x; id < ng * runs; id += (size_t)gridDim.x * blockDim.x) {
        const size_t bi = id / runs, grp = g0 + bi, i0 = grp * g;
        uint32_t t = (uint32_t)(id - bi * runs);
        G1Xyzz* out = tmp + (bi << hb) + (size_t)t * run;
        G1Xyzz acc = g1_xyzz_identity();
#pragma unroll 1
        for (unsigned pos = low; pos < g; pos++) {
            const int code = (int)(t % B) - (int)c;
            t /= B;
            const G1Affine q = msm_comb_top_point(tq, n, i0 + pos, code < 0);
#pragma unroll 1
            for (int k = 0; k < (code < 0 ? -code : code); k++) g1_madd<true>(acc, q);
        }
        const G1Affine q0 = msm_comb_top_point(tq, n, i0, false), m0 = msm_comb_top_point(tq, n, i0, true);
        const G1Affine q1 = msm_comb_top_point(tq, n, low == 2 ? i0 + 1 : n, false), m1 = msm_comb_top_point(tq, n, low == 2 ? i0 + 1 : n, true);
#pragma unroll 1
        for (unsigned k = 0; k < c; k++) {
            g1_madd<true>(acc, m0);
            g1_madd<true>(acc, m1);
        }
#pragma unroll 1
        for (unsigned d1 = 0; d1 < (low == 2 ? B : 1u); d1++) {
            const bool down = (d1 & 1u) != 0;
#pragma unroll 1
            for (unsigned st = 0; st < B; st++) {
                out[d1 * B + (down ? B - 1 - st : st)] = acc;
                if (st + 1 < B) g1_madd<true>(acc, down ? m0 : q0);
            }
            if (d1 + 1 < B) g1_madd<true>(acc, q1);
        }
    }
}

// digits[(m * A + j) * nd + i] = column j of scalar i of MSM m: the index of its table entry in bits 0 .. H-2, bit 31 set when the
// entry is to be subtracted.  Scalar vector of MSM m as in msm_sort_kernel (stride / inner / outer_stride).  nd = n without top
// tables.  TOP (see msm_comb_shape): a workgroup takes PER = g floor(256 / g) scalars — whole groups — and, after the columns,
// lane t < PER / g folds the top codes of its group into the digit of virtual scalar n + grp / A of column grp % A
// (nd = n + ceil(ceil(n / g) / A); a group past the last scalar gets the all-zero code: the identity entry of its block).
template <unsigned H, bool TOP> __global__ void __launch_bounds__(256) msm_comb_digits_kernel(const Fr* scalars, size_t n, size_t stride, size_t inner,
                                                                                              size_t outer_stride, size_t m0, uint32_t* digits, size_t nd) {
    constexpr MsmCombShape SH = msm_comb_shape(H, TOP);
    constexpr unsigned A = SH.a, L = A * H;
    static_assert(L <= 9 * 32 && H <= MSM_COMB_MAX_TEETH, "the recoded scalar is kept in nine words");
    constexpr unsigned PER = TOP ? (256 / (SH.top_g ? SH.top_g : 1)) * (SH.top_g ? SH.top_g : 1) : 256;
    const size_t m = m0 + blockIdx.y, i = (size_t)blockIdx.x * PER + threadIdx.x;  // (a grid row per MSM: no division per lane)
    const bool live = threadIdx.x < PER && i < n;
    int code = 0;
    if (live) {
        const Fr* sc = scalars + (m % inner) * stride + (m / inner) * outer_stride;
        const Fr s = fp_load(sc + i);  // the Montgomery residue s R mod r, taken as the integer it is: the table holds multiples of R^-1 P
        // odd representative: s, or r - s with the sign of the base flipped (r is odd; s = 0 becomes r, and r P = O comes out of the sums)
        const bool even = !(s.v[0] & 1u);
        uint32_t sp[8], br = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t d = fp_sbb(FrParams::mod(k), s.v[k], br);
            sp[k] = even ? d : s.v[k];
        }
        // t = (v + 2^L - 1) / 2 = (v >> 1) + 2^(L-1), v = s' (mod 2^L when the top tables take the rest): bit p of t is 1 where the
        // digit of 2^p is +1
        uint32_t w[9];
#pragma unroll
        for (int k = 0; k < 8; k++) w[k] = (sp[k] >> 1) | (k < 7 ? sp[k + 1] << 31 : 0u);
        w[8] = 0;
        if constexpr (TOP) {
            constexpr unsigned hi = L >> 5, sh = L & 31;  // u = bits L .. 253 of s'
            uint32_t u = sp[hi] >> sh;
            if constexpr (sh + SH.top_bits > 32 && hi + 1 < 8) u |= sp[hi + 1] << (32 - sh);
            code = (int)(u & ((1u << SH.top_bits) - 1u));
            if (even) code = -code;
#pragma unroll
            for (unsigned k = 0; k < 9; k++) {  // v >> 1 keeps bits 0 .. L-2
                if (k > ((L - 1) >> 5)) w[k] = 0;
                else if (k == ((L - 1) >> 5)) w[k] &= (1u << ((L - 1) & 31)) - 1u;
            }
        }
        w[(L - 1) >> 5] |= 1u << ((L - 1) & 31);
        uint32_t* out = digits + (m * A) * nd + i;
#pragma unroll
        for (unsigned j = 0; j < A; j++) {
            uint32_t idx = 0;
#pragma unroll
            for (unsigned k = 0; k < H; k++) {
                const unsigned p = j + A * k;
                idx |= ((w[p >> 5] >> (p & 31)) & 1u) << k;
            }
            const uint32_t top = idx >> (H - 1), mask = (1u << (H - 1)) - 1u;
            const uint32_t low = top ? (idx & mask) : (~idx & mask);  // top tooth -1: minus the entry of the complemented teeth
            const uint32_t neg = (top ? 0u : 1u) ^ (even ? 1u : 0u);
            out[(size_t)j * nd] = low | (neg << 31);
        }
    }
    if constexpr (TOP) {
        constexpr unsigned G = SH.top_g, B = SH.top_b;
        __shared__ int codes[256];
        codes[threadIdx.x] = code;
        __syncthreads();
        const size_t grp = (size_t)blockIdx.x * (PER / G) + threadIdx.x, nv = msm_comb_virtual(n, SH);
        if (threadIdx.x < PER / G && grp < nv * A) {
            uint32_t idx = 0, pw = 1;
#pragma unroll
            for (unsigned pos = 0; pos < G; pos++) {
                idx += (uint32_t)(codes[threadIdx.x * G + pos] + (int)((B - 1) / 2)) * pw;
                pw *= B;
            }
            const size_t v = grp / A, j = grp - v * A;
            digits[(m * A + j) * nd + n + v] = (uint32_t)((v * (A - 1) + j) << (H - 1)) + idx;
        }
    }
}

// Partition of a workgroup's nsub scalars x a columns over its MSM_BLOCK lanes (see the header): q lanes per column take p scalars
// each out of the first n_main; the other lanes share the a * n_left items left, in column-major order.
struct MsmCombPlan {
    uint32_t q, p, n_main, n_left;
};
PLONK_HD MsmCombPlan msm_comb_plan(uint32_t nsub, uint32_t a) {
    MsmCombPlan pl;
    pl.q = MSM_BLOCK / a;
    pl.p = (a * nsub + MSM_BLOCK - 1) / MSM_BLOCK;
    const uint32_t reach = pl.q * pl.p;
    pl.n_main = reach < nsub ? reach : nsub;
    pl.n_left = nsub - pl.n_main;
    return pl;
}
// pieces of column j left by the left-over lanes: lanes ulo .. ulo + x - 1
PLONK_HD void msm_comb_left_pieces(const MsmCombPlan& pl, uint32_t j, uint32_t& ulo, uint32_t& x) {
    if (!pl.n_left) {
        ulo = 0;
        x = 0;
        return;
    }
    ulo = (j * pl.n_left) / pl.p;
    x = ((j + 1) * pl.n_left - 1) / pl.p - ulo + 1;
}
// LDS slot of piece r of column j: the q column lanes first, then the left-over lanes' pieces (lane u, column j: slot a q + u + j —
// lanes and the columns they touch are both monotone, so the slot is unique)
PLONK_HD uint32_t msm_comb_slot(const MsmCombPlan& pl, uint32_t a, uint32_t j, uint32_t r, uint32_t ulo) {
    return r < pl.q ? j * pl.q + r : a * pl.q + ulo + j + (r - pl.q);
}

__global__ void __launch_bounds__(MSM_BLOCK, MSM_ACC_WAVES) msm_comb_kernel(const G1Affine* lookup, unsigned hb, unsigned a,
                                                                             const uint32_t* digits, size_t n, unsigned G,
                                                                             G1Xyzz* partial, MsmDeferred* deferred, size_t deferred_stride,
                                                                             uint32_t* n_deferred, unsigned n_real, unsigned top_delta) {
    // (n counts the virtual scalars of a comb with top tables, n_real the real ones: block of scalar i = msm_comb_block_of)
    PLONK_DYN_SMEM(smem);  // (MSM_BLOCK + a) pieces of 128 B
    G1Xyzz* red = reinterpret_cast<G1Xyzz*>(smem);
    const unsigned m = blockIdx.x / G, g = blockIdx.x % G, tid = threadIdx.x;
    const uint32_t per_g = (uint32_t)((n + G - 1) / G);
    const uint32_t s0 = g * per_g < n ? g * per_g : (uint32_t)n, s1 = s0 + per_g < n ? s0 + per_g : (uint32_t)n, nsub = s1 - s0;
    const MsmCombPlan pl = msm_comb_plan(nsub, a);
    const uint32_t* dg = digits + (size_t)m * a * n + s0;  // dg[j * n + i], i relative to s0
    const G1Affine* tab = lookup + ((size_t)s0 << hb);
    const uint32_t nr_rel = n_real > s0 ? n_real - s0 : 0u;  // first virtual scalar, relative to s0
    uint32_t j, i, step, count, slot;
    const bool column_lane = tid < a * pl.q;
    if (column_lane) {
        j = tid / pl.q;
        i = tid - j * pl.q;
        step = pl.q;
        count = i < pl.n_main ? (pl.n_main - 1 - i) / pl.q + 1 : 0;
        slot = tid;
    } else {
        const uint32_t u = tid - a * pl.q, items = a * pl.n_left;
        const uint32_t lo = u * pl.p < items ? u * pl.p : items, hi = lo + pl.p < items ? lo + pl.p : items;
        count = hi - lo;
        j = pl.n_left ? lo / pl.n_left : 0;
        i = pl.n_main + (lo - j * pl.n_left);
        step = 1;
        slot = a * pl.q + u + j;
    }
    G1XyzzL run = g1l_identity();
    auto flush = [&]() { red[slot] = g1l_to_piece(run); };  // [0, 4m) words: what g1l_from_piece takes
    for (uint32_t k = 0; k < count; k++) {
        const uint32_t d = dg[(size_t)j * n + i];
        const G1Affine* src = tab + (((size_t)(i + (i >= nr_rel ? top_delta : 0u)) << hb) + (d & 0x7fffffffu));
        const Fq x = fp_load(&src->x), y = fp_load(&src->y);
        if (!g1l_madd_fast(run, x, y, (d >> 31) != 0) && !(fp_is_zero(x) && fp_is_zero(y))) {  // see msm_accumulate_kernel
            const uint32_t sl = atomicAdd(n_deferred + m, 1u);
            if (sl < MSM_DEFER_CAP) deferred[(size_t)m * deferred_stride + sl] = MsmDeferred{(uint32_t)(j * n + s0 + i), d};
        }
        i += step;
        if (i >= nsub && k + 1 < count) {  // a left-over lane moves on to the next column
            flush();
            run = g1l_identity();
            slot++;
            j++;
            i = pl.n_main;
        }
    }
    if (column_lane || count) flush();
    __syncthreads();
    // Per-column tree over the pieces: level by level the upper half of every column's list is added onto its lower half.
    // Lists differ in length by at most the left-over pieces (<= 2 + 1); `cur` walks the longest one.
    uint32_t cur = pl.q + (pl.n_left ? (pl.n_left + pl.p - 1) / pl.p + 1 : 0);
    while (cur > 1) {
        const uint32_t half = (cur + 1) / 2, cnt = cur - half;
        for (uint32_t w = tid; w < a * cnt; w += MSM_BLOCK) {
            const uint32_t jj = w / cnt, r = w - jj * cnt;
            uint32_t ulo, x;
            msm_comb_left_pieces(pl, jj, ulo, x);
            if (r + half < pl.q + x) {
                const uint32_t sa = msm_comb_slot(pl, a, jj, r, ulo), sb = msm_comb_slot(pl, a, jj, r + half, ulo);
                const G1XyzzL o = g1l_from_piece(&red[sb]);
                if (!o.inf) {
                    G1XyzzL v = g1l_from_piece(&red[sa]);
                    if (!g1l_add_fast(v, o)) atomicOr(n_deferred + m, MSM_COMB_REDO);  // equal or opposite sums: msm_comb_slow_kernel
                    red[sa] = g1l_to_piece_wide(v);
                }
            }
        }
        __syncthreads();
        cur = half;
    }
    if (tid < a) partial[((size_t)m * G + g) * a + tid] = red[tid * pl.q];
}

// colsum[m * a + j] = sum_g partial[(m * G + g) * a + j]: a wave per (MSM, column), lane g (G <= 64)
__global__ void __launch_bounds__(64) msm_comb_colsum_kernel(const G1Xyzz* partial, unsigned G, unsigned a, G1Xyzz* colsum, uint32_t* n_deferred) {
    const size_t m = blockIdx.x / a, j = blockIdx.x % a;
    const unsigned lane = threadIdx.x;
    G1XyzzL acc = g1l_identity();
    bool ok = true;
    for (unsigned g = lane; g < G; g += 64) ok &= g1l_add_fast(acc, g1l_from_piece(&partial[(m * G + g) * a + j]));
    ok &= g1l_wave_reduce_step<32>(acc, lane);
    ok &= g1l_wave_reduce_step<16>(acc, lane);
    ok &= g1l_wave_reduce_step<8>(acc, lane);
    ok &= g1l_wave_reduce_step<4>(acc, lane);
    ok &= g1l_wave_reduce_step<2>(acc, lane);
    ok &= g1l_wave_reduce_step<1>(acc, lane);
    if (!ok) atomicOr(n_deferred + m, MSM_COMB_REDO);  // equal or opposite operands somewhere: msm_comb_slow_kernel
    if (lane == 0) colsum[m * a + j] = g1l_to_piece_wide(acc);
}

// sum_j 2^j S_j for MSM m, S_j = sum_g sums[(m G + g) a + j] + the deferred additions of column j.  LPM lanes per MSM: lane l
// takes the columns j = l (mod LPM) by Horner from the top (LPM doublings between two of them), doubles its share l more times,
// and the LPM shares are summed by the cross-lane butterfly.  LPM = 1 is the plain Horner chain, one MSM per lane (least work:
// a batch); larger groups shorten the chain of a - 1 serial doublings at the price of idle lanes (few MSMs: latency).
template <unsigned LPM> __global__ void __launch_bounds__(64) msm_comb_finalize_kernel(const G1Xyzz* sums, size_t M, unsigned G, unsigned a,
                                                                                       const G1Affine* lookup, unsigned hb, size_t n,
                                                                                       const MsmDeferred* deferred, size_t deferred_stride,
                                                                                       uint32_t* n_deferred, Fq* out_xy, uint8_t* flags, unsigned n_real,
                                                                                       unsigned top_delta) {
    const size_t gid = (size_t)blockIdx.x * 64 + threadIdx.x, m = gid / LPM;
    const unsigned l = (unsigned)(gid % LPM), lane = threadIdx.x;
    G1XyzzL acc = g1l_identity();
    bool ok = true;  // false: an addition met equal or opposite operands — the MSM goes to msm_comb_slow_kernel
    if (m < M && l < a) {
        const uint32_t nd = msm_comb_needs_redo(n_deferred[m]) ? 0u : n_deferred[m];  // (an MSM to be redone: its list may have holes, its sum is overwritten)
        const unsigned jtop = l + ((a - 1 - l) / LPM) * LPM;
#pragma unroll 1
        for (int j = (int)jtop; j >= 0; j -= (int)LPM) {
            if (j != (int)jtop)
#pragma unroll 1
                for (unsigned d = 0; d < LPM; d++) g1l_dbl(acc);
#pragma unroll 1
            for (unsigned g = 0; g < G; g++) ok &= g1l_add_fast<true>(acc, g1l_from_piece(&sums[(m * G + g) * a + j]));  // (<true>: opposite operands give the identity)
#pragma unroll 1
            for (uint32_t k = 0; k < nd; k++) {
                const MsmDeferred e = deferred[m * deferred_stride + k];  // `bucket` carries the item j * n + i here
                if (e.bucket / n != (uint32_t)j) continue;
                const size_t i = e.bucket - (size_t)j * n;
                const G1Affine* src = lookup + ((msm_comb_block_of(i, n_real, top_delta) << hb) + (e.entry & 0x7fffffffu));
                // deferred for being exceptional against its LANE's sum at the time; against the column's sum it normally is not
                const Fq x = fp_load(&src->x), y = fp_load(&src->y);
                if (!(fp_is_zero(x) && fp_is_zero(y))) ok &= g1l_madd_fast(acc, x, y, (e.entry >> 31) != 0);
            }
        }
#pragma unroll 1
        for (unsigned d = 0; d < l; d++) g1l_dbl(acc);
    }
    if constexpr (LPM >= 2) ok &= g1l_wave_reduce_step<1, true>(acc, lane);
    if constexpr (LPM >= 4) ok &= g1l_wave_reduce_step<2, true>(acc, lane);
    if constexpr (LPM >= 8) ok &= g1l_wave_reduce_step<4, true>(acc, lane);
    if constexpr (LPM >= 16) ok &= g1l_wave_reduce_step<8, true>(acc, lane);
    if constexpr (LPM >= 32) ok &= g1l_wave_reduce_step<16, true>(acc, lane);
    if constexpr (LPM >= 64) ok &= g1l_wave_reduce_step<32, true>(acc, lane);
    if (m < M && !ok) atomicOr(n_deferred + m, MSM_COMB_REDO);
    if (m < M && l == 0) {
        G1Affine r = g1_to_affine(g1l_to_xyzz(acc));
        flags[m] = g1_affine_is_identity(r) ? 1 : 0;
        fp_store(out_xy + 2 * m, fp_from_mont(r.x));
        fp_store(out_xy + 2 * m + 1, fp_from_mont(r.y));
    }
}

// Recovery path (MSM_DEFER_CAP): MSM m recomputed from its digits with the general formulas; one workgroup per MSM, which exits
// at once unless the MSM overflowed its deferred list.  Lane t: Horner over the columns of the scalars t, t + 256, ..
__global__ void __launch_bounds__(256) msm_comb_slow_kernel(const G1Affine* lookup, unsigned hb, unsigned a, const uint32_t* digits, size_t n,
                                                            const uint32_t* n_deferred, Fq* out_xy, uint8_t* flags, unsigned n_real, unsigned top_delta) {
    __shared__ G1Xyzz red[256];
    const unsigned m = blockIdx.x, tid = threadIdx.x;
    if (!msm_comb_needs_redo(n_deferred[m])) return;
    const uint32_t* dg = digits + (size_t)m * a * n;
    G1Xyzz acc = g1_xyzz_identity();
#pragma unroll 1
    for (int j = (int)a - 1; j >= 0; j--) {
        g1_dbl(acc);
#pragma unroll 1
        for (size_t i = tid; i < n; i += 256) {
            const uint32_t d = dg[(size_t)j * n + i];
            const G1Affine* src = lookup + ((msm_comb_block_of(i, n_real, top_delta) << hb) + (d & 0x7fffffffu));
            G1Affine pt;
            pt.x = fp_load(&src->x);
            pt.y = fp_load(&src->y);
            if (d >> 31) pt.y = fp_neg(pt.y);
            g1_madd<true>(acc, pt);
        }
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
        G1Affine r = g1_to_affine(total);
        flags[m] = g1_affine_is_identity(r) ? 1 : 0;
        fp_store(out_xy + 2 * m, fp_from_mont(r.x));
        fp_store(out_xy + 2 * m + 1, fp_from_mont(r.y));
    }
}

// Verification of a comb table found in the registry by its 64-bit key (msm.hip, lut_verified), against THIS SRS's bases:
//   every base: the all-ones entry = (sum_k 2^(a k)) R^-1 P_i, recomputed by doublings and additions from the base;
//   LUT_VERIFY_SAMPLES bases: entry 0 (every lower tooth -1) and the entry of the alternating index 0101.. as well.
// A table of another tooth count / spacing, or of other bases, filed under the same key fails here.
PLONK_DEV bool msm_comb_entry_matches(const G1Affine& P0, const MsmCombScale& k, unsigned a, unsigned h, uint32_t idx, const G1Affine* e) {
    const G1Affine P = g1_to_affine(msm_comb_scaled_base(P0, k));
    const G1Xyzz v = msm_comb_entry_slow(P, a, h, idx);
    const Fq ex = fp_load(&e->x), ey = fp_load(&e->y);
    if (g1_is_identity(v)) return fp_is_zero(ex) && fp_is_zero(ey);
    return fp_eq(fp_mul(ex, v.zz), v.x) && fp_eq(fp_mul(ey, v.zzz), v.y);
}
__global__ void __launch_bounds__(64) msm_comb_verify_kernel(const G1Affine* bases, const G1Affine* lookup, size_t n, unsigned a, unsigned h,
                                                             unsigned samples, MsmCombScale k, unsigned* mismatches) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n + 2 * (size_t)samples) return;
    const uint32_t mask = (1u << (h - 1)) - 1u;
    size_t i;
    uint32_t idx;
    if (t < n) {
        i = t;
        idx = mask;
    } else {
        const unsigned s = (unsigned)(t - n) >> 1;
        i = n <= samples ? (s < n ? s : n - 1) : (size_t)s * (n - 1) / (samples - 1);
        idx = ((t - n) & 1) ? (0x55555555u & mask) : 0u;
    }
    G1Affine b;
    b.x = fp_load(&bases[i].x);
    b.y = fp_load(&bases[i].y);
    if (!msm_comb_entry_matches(b, k, a, h, idx, lookup + ((i << (h - 1)) + idx))) atomicAdd(mismatches, 1u);
}

// ... and of its top tables: for every group the entry whose codes are all +1 = sum_pos 2^(L - j) R^-1 P_(g grp + pos), recomputed
// from this SRS's bases (a table of another tooth count, group size or base set fails here)
__global__ void __launch_bounds__(64) msm_comb_verify_top_kernel(const G1Affine* bases, const G1Affine* lookup, size_t n, unsigned a, unsigned h, unsigned g,
                                                                 unsigned B, MsmCombScale k, unsigned* mismatches) {
    const size_t grp = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (grp >= msm_comb_top_groups(n, g)) return;
    const unsigned L = a * h, j = (unsigned)(grp % a);
    G1Xyzz sum = g1_xyzz_identity();
    uint32_t idx = 0, pw = 1;
#pragma unroll 1
    for (unsigned pos = 0; pos < g; pos++, pw *= B) {
        const size_t i = grp * g + pos;
        idx += ((B - 1) / 2 + (i < n ? 1u : 0u)) * pw;
        if (i >= n) continue;
        G1Affine b;
        b.x = fp_load(&bases[i].x);
        b.y = fp_load(&bases[i].y);
        G1Xyzz p = msm_comb_scaled_base(b, k);
#pragma unroll 1
        for (unsigned d = 0; d < L - j; d++) g1_dbl(p);
        g1_add(sum, p);
    }
    const G1Affine* e = lookup + (((n + grp) << (h - 1)) + idx);
    const Fq ex = fp_load(&e->x), ey = fp_load(&e->y);
    const bool ok = g1_is_identity(sum) ? (fp_is_zero(ex) && fp_is_zero(ey)) : (fp_eq(fp_mul(ex, sum.zz), sum.x) && fp_eq(fp_mul(ey, sum.zzz), sum.y));
    if (!ok) atomicAdd(mismatches, 1u);
}
