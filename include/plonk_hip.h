/* plonk_hip.h — C-ABI of libplonk_hip.so, the MI355X (gfx950) backend for plonkathon's prover hot path.
 *
 * The reference (0xPARC/plonkathon) is single-process pure Python and has no FFI layer; the drop-in
 * boundary is therefore the set of Python methods on its hot path.  Each entry
 * point below names the reference method it replaces (file:line under /root/reference); the
 * ctypes stubs a maintainer would add to the reference are shown in INTEGRATION.md, and
 * plonkathon_amd/ is a Python host layer with the reference's own class/method names on top.
 *
 * Conventions
 *   - plain C, no C++ types, no exceptions cross the boundary; every call returns PLONK_OK (0) or
 *     a negative PLONK_ERR_* and sets a thread-local message readable with plonk_last_error().
 *   - host-side field elements are CANONICAL integers (< modulus), 32 bytes little-endian.
 *     Device-side Fr vectors are 32 bytes/element Montgomery residues (R = 2^261, 8 x u32 LE
 *     limbs, canonical) and are opaque to the caller.
 *   - G1 points on the host are affine canonical x||y, 64 bytes little-endian each coordinate,
 *     plus an out-of-band identity flag (py_ecc represents the identity as None).
 *   - all work is enqueued on the context's HIP stream; calls that return host data synchronise.
 *   - the library never retains caller host pointers past return.
 */
#ifndef PLONK_HIP_H
#define PLONK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLONK_OK 0
#define PLONK_ERR_ARG (-1)   /* bad argument (maps to AssertionError / ValueError in Python) */
#define PLONK_ERR_HIP (-2)   /* a HIP runtime call failed */
#define PLONK_ERR_NOMEM (-3) /* device allocation failed */
#define PLONK_ERR_STATE (-4) /* object used in the wrong state */
#define PLONK_ERR_TIMEOUT (-5) /* a collective (or ncclCommInitRank) did not complete within the communicator's deadline */

#define PLONK_ABI_VERSION 2

typedef struct plonk_ctx plonk_ctx; /* one per (process, device): stream, twiddle caches, scratch */
typedef struct plonk_srs plonk_srs; /* device-resident G1 bases + fixed-base window table */

/* ---- library / context ---------------------------------------------------------------------- */
const char* plonk_last_error(void);
int plonk_abi_version(void);
int plonk_device_count(int* out_count);
int plonk_ctx_create(int device, plonk_ctx** out_ctx);
int plonk_ctx_destroy(plonk_ctx* ctx);
int plonk_ctx_sync(plonk_ctx* ctx);
int plonk_ctx_device_name(plonk_ctx* ctx, char* buf, size_t buf_len);

/* ---- raw device memory (Polynomial.values storage; poly.py:10-21) --------------------------- */
/* page-locked host memory for plonk_prover_upload_variables_async (hipHostMalloc / hipHostFree) */
int plonk_host_alloc(plonk_ctx* ctx, size_t bytes, void** out_hptr);
int plonk_host_free(plonk_ctx* ctx, void* hptr);
int plonk_mem_alloc(plonk_ctx* ctx, size_t bytes, void** out_dptr);
int plonk_mem_free(plonk_ctx* ctx, void* dptr);
int plonk_mem_info(plonk_ctx* ctx, size_t* out_free, size_t* out_total); /* hipMemGetInfo of the context's device */
int plonk_mem_h2d(plonk_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int plonk_mem_d2h(plonk_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);
int plonk_mem_d2d(plonk_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);
int plonk_mem_zero(plonk_ctx* ctx, void* d_dst, size_t bytes);

/* ---- Fr vectors: host canonical LE  <->  device Montgomery ----------------------------------
 * Replaces building `list[Scalar]` (curve.py:10-11; poly.py:14-18). Conversion runs on the GPU, the range check
 * `value < r` with it: when plonk_fr_upload returns PLONK_ERR_ARG (an element >= r; plonk_last_error names the first) the
 * destination has ALREADY been overwritten with the converted input and must not be used; plonk_prover_upload_variables
 * then leaves the prover with no resident batch (run / download return PLONK_ERR_STATE until the next good upload). */
int plonk_fr_upload(plonk_ctx* ctx, void* d_dst, const uint8_t* h_src_le32, size_t count);
int plonk_fr_download(plonk_ctx* ctx, uint8_t* h_dst_le32, const void* d_src, size_t count);

/* ---- NTT family ------------------------------------------------------------------------------
 * plonk_fr_ntt          Polynomial.fft / ifft            poly.py:113-148 (natural order in and out;
 *                       inverse includes the 1/N scale; roots w = 5^((r-1)/N), curve.py:14-24)
 * plonk_fr_coset_extend Polynomial.to_coset_extended_lagrange(offset)   poly.py:156-163
 *                       in: N Lagrange values, out: 4N values P(offset * mu^k), mu = root_of_unity(4N)
 * plonk_fr_coset_to_coeffs  Polynomial.coset_extended_lagrange_to_coeffs(offset)  poly.py:169-177
 *                       in: M values on the coset, out: M coefficients (M = 1 << log_m)
 * `batch` independent vectors are laid out back to back ([batch][N]). in may equal out.        */
int plonk_fr_ntt(plonk_ctx* ctx, const void* d_in, void* d_out, unsigned log_n, int inverse, size_t batch);
int plonk_fr_coset_extend(plonk_ctx* ctx, const void* d_in, void* d_out, unsigned log_n,
                          const uint8_t offset_le32[32], size_t batch);
int plonk_fr_coset_to_coeffs(plonk_ctx* ctx, const void* d_in, void* d_out, unsigned log_m,
                             const uint8_t offset_le32[32], size_t batch);
/* Two-pass transforms multiply every output of the column pass by w_N^(column * frequency).  Within this budget (bytes per
 * context, default 4 GiB; 0 = never) the library keeps those N factors per (size, direction) as one table in the order
 * the kernel reads them — 80 bytes per point: 84 MB at 2^20, 1.3 GB at 2^24 — and spends one multiplication per element
 * instead of two (factors from two small tables).  Same results either way. */
int plonk_ntt_set_table_budget(plonk_ctx* ctx, size_t bytes);
/* ---- the same transform over the BLS12-381 scalar field (ntt_bls.hip) -------------------------
 * r = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001, w = 7^((r-1)/N): the field BASELINE.json's
 * standalone-NTT metric is quoted on.  The reference has no such field (curve.py:2: BN254 throughout), so this replaces no
 * reference call: it is poly.py:113-148's transform (natural order in and out, the inverse includes 1/N) with the modulus
 * and generator swapped, on the same wave kernels.  Elements are 32 bytes on the device like Fr (Montgomery form, R = 2^261);
 * plonk_mem_* allocate and move them.  Sizes: 2^8 .. 2^26 (PLONK_ERR_ARG otherwise); in may equal out. */
int plonk_bls_fr_upload(plonk_ctx* ctx, void* d_dst, const uint8_t* h_src_le32, size_t count);
int plonk_bls_fr_download(plonk_ctx* ctx, uint8_t* h_dst_le32, const void* d_src, size_t count);
int plonk_bls_fr_ntt(plonk_ctx* ctx, const void* d_in, void* d_out, unsigned log_n, int inverse, size_t batch);
/* poly.py:156-163 and 169-177 over that field (what plonk_fr_coset_extend / plonk_fr_coset_to_coeffs are for BN254):
 * n Lagrange values -> their 4n values on the coset offset * <w_4n>; M coset values -> M coefficients. */
int plonk_bls_fr_coset_extend(plonk_ctx* ctx, const void* d_in, void* d_out, unsigned log_n, const uint8_t offset_le32[32], size_t batch);
int plonk_bls_fr_coset_to_coeffs(plonk_ctx* ctx, const void* d_in, void* d_out, unsigned log_m, const uint8_t offset_le32[32], size_t batch);
/* Lower-level pieces used by the batched prover: coefficient form in, fixed offset table. */
int plonk_fr_coset_ntt_from_coeffs(plonk_ctx* ctx, const void* d_coeffs, void* d_out, unsigned log_n,
                                   unsigned log_expand, const uint8_t offset_le32[32], size_t batch);

/* ---- pointwise arithmetic ---------------------------------------------------------------------
 * plonk_fr_pointwise    Polynomial.__add__/__sub__/__mul__/__truediv__ with a Polynomial
 *                       poly.py:23-36, 45-58, 68-77, 85-94   (x / 0 == 0, as py_ecc)
 * plonk_fr_scalar_op    the same operators with a Scalar operand, poly.py:37-43, 59-65, 78-83, 95-100;
 *                       constant_term_only=1 gives the MONOMIAL-basis +/- rule (poly.py:39-43, 61-65)
 * plonk_fr_rotate       Polynomial.shift(k)                  poly.py:102-109
 * plonk_fr_batch_inverse   the per-element inversions inside `/` and barycentric_eval
 * plonk_fr_barycentric  Polynomial.barycentric_eval(x)       poly.py:181-195                      */
#define PLONK_OP_ADD 0
#define PLONK_OP_SUB 1
#define PLONK_OP_MUL 2
#define PLONK_OP_DIV 3
int plonk_fr_pointwise(plonk_ctx* ctx, int op, const void* d_a, const void* d_b, void* d_out, size_t count);
int plonk_fr_scalar_op(plonk_ctx* ctx, int op, const void* d_a, const uint8_t scalar_le32[32], void* d_out,
                       size_t count, int constant_term_only);
int plonk_fr_rotate(plonk_ctx* ctx, const void* d_in, void* d_out, size_t count, size_t shift);
int plonk_fr_batch_inverse(plonk_ctx* ctx, const void* d_in, void* d_out, size_t count);
int plonk_fr_barycentric(plonk_ctx* ctx, const void* d_vals, unsigned log_n, const uint8_t x_le32[32],
                         uint8_t out_le32[32]);
/* batches of the operators above, for callers that issue them in runs (each call of the single forms is a launch, and each
 * barycentric_eval a host synchronisation):
 * plonk_fr_barycentric_many  n_polys <= 16 polynomials (separate buffers, 2^log_n Lagrange values each), polynomial k
 *                       evaluated at xs[k]: the six evaluations of prover.py:228-239 in one kernel and one synchronisation
 * plonk_fr_lincomb      d_out[i] = constant + sum_k scalars[k] * d_terms[k][i]: a run of `Polynomial * Scalar`, `+`, `-`
 *                       (poly.py:23-83) such as the linearisation polynomial and the opening numerators of
 *                       prover.py:245-288, in one pass; n_terms <= 20; d_out may alias a term                          */
int plonk_fr_barycentric_many(plonk_ctx* ctx, size_t n_polys, const void* const* d_vals, unsigned log_n, const uint8_t* xs_le32,
                              uint8_t* out_le32);
int plonk_fr_lincomb(plonk_ctx* ctx, size_t n_terms, const void* const* d_terms, const uint8_t* scalars_le32,
                     const uint8_t constant_le32[32], void* d_out, size_t count);

/* plonk_fr_powers       out[k] = first * base^k, k < count: Scalar.roots_of_unity (curve.py:19-24), and the coset points
 *                       X_big[k] = fft_cofactor * mu^k the prover divides by (prover.py:160-161, 271-283)
 * plonk_fr_equal        the reference's `==` on two Polynomials (poly.py:20-21) and its `values[k:] == [0] * m` sanity
 *                       checks (prover.py:205-208, 288, 299) on the device: *out_equal = (a == b elementwise), d_b NULL
 *                       compares with zero; only the verdict crosses PCIe                                             */
int plonk_fr_powers(plonk_ctx* ctx, const uint8_t first_le32[32], const uint8_t base_le32[32], size_t count, void* d_out);
int plonk_fr_equal(plonk_ctx* ctx, const void* d_a, const void* d_b, size_t count, int* out_equal);

/* ---- the fused round kernels on their own (what Prover.round_2 / round_3 of the reference-shaped API call) -----------
 * plonk_fr_grand_product  prover.py:121-146: the permutation accumulator from the wire values A, B, C and the
 *                       permutation polynomials S1, S2, S3 (Lagrange, 2^log_n each): Z_0 = 1, Z_{i+1} = Z_i num_i / den_i
 *                       with num_i = (A_i + beta w^i + gamma)(B_i + 2 beta w^i + gamma)(C_i + 3 beta w^i + gamma),
 *                       den_i = (A_i + beta S1_i + gamma)(..S2..)(..S3..); x / 0 == 0 as py_ecc.  Two block scans and
 *                       one inversion instead of n.  *out_closes = 1 iff the product closes to 1 (prover.py:132).
 * plonk_fr_quotient     prover.py:188-203: QUOT_big = (gate + alpha * permutation + alpha^2 (Z - 1) L0) / Z_H on the
 *                       4n-point coset offset * mu^k, one fused pass.  d_evals = the coset extensions (fft_expand,
 *                       4 * 2^log_n values each) of A, B, C, PI, Z, QL, QR, QM, QO, QC, S1, S2, S3, L0, in that order;
 *                       Z(w x) is Z_big read 4 places ahead (prover.py:173), X_big and 1 / Z_H come from `offset`.      */
int plonk_fr_grand_product(plonk_ctx* ctx, const void* d_a, const void* d_b, const void* d_c, const void* d_s1, const void* d_s2,
                           const void* d_s3, unsigned log_n, const uint8_t beta_le32[32], const uint8_t gamma_le32[32],
                           void* d_z_out, int* out_closes);
int plonk_fr_quotient(plonk_ctx* ctx, unsigned log_n, const void* const d_evals[14], const uint8_t offset_le32[32],
                      const uint8_t alpha_le32[32], const uint8_t beta_le32[32], const uint8_t gamma_le32[32], void* d_out);

/* ---- G1 multi-scalar multiplication ------------------------------------------------------------
 * plonk_srs_load_ptau   Setup.from_file's G1 section, setup.py:29-41: `n_points` affine points, each
 *                       64 B = x||y little-endian in Montgomery form with R = 2^256 — bytes 80.. of a snarkjs
 *                       .ptau are passed through as they are; the device rescales them (x 2^5) to the
 *                       library's radix instead of dividing the factor out as setup.py:39-40 does.
 * plonk_srs_load_affine arbitrary bases for ec_lincomb (curve.py:38-44): canonical x||y LE,
 *                       (0,0) encodes the identity (py_ecc None).
 * plonk_g1_msm          ec_lincomb / lincomb / multisubset, curve.py:38-111, i.e. the body of
 *                       Setup.commit after its ifft (setup.py:66-72).  Pippenger over a fixed-base
 *                       window table; `batch` scalar vectors share the bases.  Scalars are device Fr
 *                       (Montgomery), vector b starts at d_scalars + b*scalar_stride elements and
 *                       uses the first n bases.  Output: batch x 64 B canonical affine x||y LE and
 *                       batch identity flags (1 = identity, coordinates then zero).               */
int plonk_srs_load_ptau(plonk_ctx* ctx, const uint8_t* g1_mont_le, size_t n_points, plonk_srs** out_srs);
int plonk_srs_load_affine(plonk_ctx* ctx, const uint8_t* xy_le, size_t n_points, plonk_srs** out_srs);
int plonk_srs_free(plonk_ctx* ctx, plonk_srs* srs);
/* Lagrange-basis view of an SRS: the 2^log_n points [L_i(tau)]_1 = sum_j (w^-ij / n) [tau^j]_1 (an inverse DFT of
 * the SRS over the group, run once per size on the device and cached), so that Setup.commit(values)
 * (setup.py:66-72: ifft, then lincomb with powers_of_x) is ONE MSM of the Lagrange values with no ifft in front, and
 * Setup.verification_key (setup.py:75-77) eight of them.  The view is owned by `srs` and freed with it; it is a
 * plonk_srs like any other (plonk_g1_msm, lookup tables).                                                        */
int plonk_srs_lagrange(plonk_ctx* ctx, plonk_srs* srs, unsigned log_n, plonk_srs** out_view);
int plonk_srs_size(const plonk_srs* srs, size_t* out_n);
int plonk_g1_msm(plonk_ctx* ctx, plonk_srs* srs, const void* d_scalars, size_t n, size_t batch,
                 size_t scalar_stride, uint8_t* h_out_xy_le, uint8_t* h_out_is_identity);
/* Table MSM: for a reusable SRS (plonk_srs_load_ptau) a table per base is precomputed once into HBM, after which an MSM is
 * N * a mixed additions of looked-up points: no sorting, no buckets.  Two layouts, same results as the bucket method
 * (curve.py:38-111), bit for bit:
 *   comb tables (default; csrc/msm_comb.h)   h teeth spaced a = ceil(254 / h) bits apart: 2^(h-1) entries per base, a additions
 *       per base and a - 1 doublings per MSM (shared by its bases).  2^11 points: 68.7 GB at h = 20 (13 additions per base),
 *       8.6 GB at h = 17 (15), 67 MB at h = 10 (26).
 *   ... with TOP TABLES (round 6; where 254 mod h is 1 or 2)   floor(254 / h) columns cover all but the top one or two bits of a
 *       scalar; those select an entry of a joint table shared by g consecutive bases ((2^(R+1) - 1)^g <= 2^(h-1) entries, R the
 *       bits left over): 1 / g of an addition per base instead of a whole column.  2^11 points: h = 21, 12 columns, g = 7 —
 *       12.15 additions per base from 157.6 GB (137.4 + 20.1), between the 13 of h = 20 and the 12 of h = 22 (275 GB).
 *   window tables (mode | 16; rounds 2 - 5)  every multiple d * 2^(c w) * P_i, d <= 2^(c-1): ceil(255 / c) windows of 2^(c-1)
 *       entries per base and as many additions, no doublings.  128.8 GB at c = 17 (15 additions), 10.7 GB at c = 13 (20).
 * mode 0 (default): automatic — the table with the fewest additions per base that, with its build staging, fits `budget_bytes`;
 * budget 0 = the library default of 1/16 of the device's memory (18 GB on an MI355X: h = 17 for 2^11 points), or
 * PLONK_MSM_TABLE_GB gigabytes if that variable is set: the big tables are a memory-for-time trade the caller opts into
 * explicitly.  Bucket method when nothing fits or for plonk_srs_load_affine bases; mode 1: never; mode 2: use `bits`
 * (h, or c with | 16) for every base set (tests).  The automatic choice weighs combs with and without top tables alike; an
 * explicit `bits` means the plain comb of h teeth, and mode | 32 the one with top tables (PLONK_ERR_ARG if h takes none). */
int plonk_msm_lookup_configure(plonk_ctx* ctx, int mode, unsigned bits, size_t budget_bytes);
/* bits (teeth h / window bits c) of the table currently attached to `srs` (0 = none: its MSMs use the bucket method) */
int plonk_srs_lookup_bits(const plonk_srs* srs, unsigned* out_bits);
/* the same plus the table's size in bytes, the wall time its build took, and how many plonk_srs objects of this
 * process share it: there is ONE table per (device, base set, layout, bits) whatever the number of contexts /
 * streams / provers using that SRS on the device; it is freed with the last plonk_srs that references it.      */
int plonk_srs_lookup_info(const plonk_srs* srs, unsigned* out_bits, size_t* out_bytes, double* out_build_s,
                          int* out_sharers);
/* its layout: kind 1 = comb, 2 = windows (0 = no table), and the mixed additions an MSM performs per base on it (a comb: its
 * columns) */
int plonk_srs_lookup_layout(const plonk_srs* srs, unsigned* out_kind, unsigned* out_additions_per_base);
/* the top tables of the attached comb: the bits R they take and the bases g sharing one (0, 0 = none).  An MSM of n scalars
 * then performs  columns * (n + ceil(ceil(n / g) / columns))  mixed additions.                                           */
int plonk_srs_lookup_top(const plonk_srs* srs, unsigned* out_top_bits, unsigned* out_bases_per_group);

/* ---- batched GPU-resident prover ---------------------------------------------------------------------
 * Replaces Prover.__init__ / Prover.prove / round_1..round_5 (prover.py:45-306) for `batch`
 * independent proofs of ONE circuit proved in lock-step, Fiat-Shamir transcript included
 * (transcript.py:77-123 runs on the device, 32 lanes per proof).
 *   plonk_prover_create   Prover(setup, program): `selectors_le32` = the eight CommonPreprocessedInput
 *                         vectors QM, QL, QR, QO, QC, S1, S2, S3 (compiler/program.py:10-30), each
 *                         2^log_n canonical Fr values; n_public = len(program.get_public_assignments()).
 *                         Extends the circuit polynomials to the quotient coset once.
 *   plonk_prover_upload_witness   the wire columns A, B, C of round 1 (prover.py:94-103) laid out
 *                         [3][batch][n] and the public inputs [batch][n_public] (PI = -public,
 *                         prover.py:57-62), canonical LE; they stay resident in HBM.
 *   plonk_prover_run      enqueue all five rounds for the resident witnesses (asynchronous); `batch` must be the
 *                         batch size of the last upload (PLONK_ERR_STATE otherwise), also for download.
 *   plonk_prover_download wait, then per proof 768 bytes: a_1, b_1, c_1, z_1, t_lo_1, t_mid_1, t_hi_1,
 *                         W_z_1, W_zw_1 as canonical x||y LE (Proof.flatten order, prover.py:18-35) then
 *                         a_eval, b_eval, c_eval, s1_eval, s2_eval, z_shifted_eval canonical LE;
 *                         status[b]: bit 0 = some commitment is the identity (the reference's
 *                         append_point(None) raises), bit 1 = Z does not close to 1 (prover.py:132),
 *                         bit 2 = a gate constraint fails on some row (prover.py:108-116; what the quotient-degree assert of
 *                         prover.py:205-208 detects),
 *                         bit 3 = a value uploaded by plonk_prover_upload_variables_async was not a canonical Fr value.
 *   plonk_prover_challenges  beta, gamma, alpha, fft_cofactor, zeta, v of proof b (tests).          */
typedef struct plonk_prover plonk_prover;
int plonk_prover_create(plonk_ctx* ctx, plonk_srs* srs, unsigned log_n, const uint8_t* selectors_le32,
                        size_t n_public, plonk_prover** out);
int plonk_prover_destroy(plonk_prover* p);
/* options (0 = default): PLONK_PROVER_LAGRANGE_COMMITS commits a_1, b_1, c_1 and z_1 (rounds 1-2) from their
 * Lagrange values over plonk_srs_lagrange instead of from coefficient forms — same group elements, same proof. */
#define PLONK_PROVER_LAGRANGE_COMMITS 1u
int plonk_prover_set_options(plonk_prover* p, unsigned flags);
int plonk_prover_upload_witness(plonk_prover* p, const uint8_t* abc_le32, const uint8_t* public_le32, size_t batch);
/* The same inputs at n_vars * 32 bytes per proof instead of 3 * n * 32: the wiring is given once per circuit —
 * cell_index[3][n] = index of the variable each wire cell (column L / R / O, row) carries, n_vars for an empty cell
 * or a padding row (witness[None] = 0, prover.py:94-95); public_index[n_public] = the public variables in row
 * order (prover.py:57-62) — and a batch is the variables' values, [batch][n_vars] canonical LE; the A, B, C columns
 * (prover.py:97-103) and the public inputs are gathered from them on the device.                                */
int plonk_prover_set_wiring(plonk_prover* p, const uint32_t* cell_index, const uint32_t* public_index, size_t n_vars);
int plonk_prover_upload_variables(plonk_prover* p, const uint8_t* vars_le32, size_t batch);
/* The same upload without a host wait: the copy runs on a copy stream of the context and overlaps the compute stream's
 * kernels (other provers' rounds on the same GPU, or this prover's previous batch), conversion and gather follow behind
 * an event.  vars_le32 must stay valid until this batch's plonk_prover_download returns, and should be page-locked
 * (plonk_host_alloc) — a pageable buffer makes the copy synchronous again.  A value that is not below r cannot be
 * reported here: it sets status bit 3 of the proof it belongs to at plonk_prover_download.                       */
int plonk_prover_upload_variables_async(plonk_prover* p, const uint8_t* vars_le32, size_t batch);
int plonk_prover_run(plonk_prover* p, size_t batch);
int plonk_prover_download(plonk_prover* p, size_t batch, uint8_t* out_proofs, uint8_t* out_status);
int plonk_prover_challenges(plonk_prover* p, size_t b, uint8_t out_le32[6 * 32]);
/* the same download as 480-byte records: the nine commitments compressed (plonk_g1_compress's encoding), then the six
 * evaluations as 32-byte BIG-endian scalars — north_star's "compressed G1 bytes" form of a proof */
int plonk_prover_download_compressed(plonk_prover* p, size_t batch, uint8_t* out_proofs480, uint8_t* out_status);

/* ---- compressed G1 encoding ---------------------------------------------------------------------------------
 * The reference's one G1 byte encoding is x then y as 32-byte big-endian integers (append_point, transcript.py:62-67); it
 * has no compressed form, so one is defined here, derived from those bytes: the 32 big-endian bytes of x with the two
 * spare top bits of byte 0 = 10 (y is the smaller root of x^3 + 3, y <= (p-1)/2), 11 (the larger root) or 01 (the point
 * at infinity, x = 0; py_ecc's None) — gnark-crypto's flag layout for BN254.
 *   plonk_g1_compress    count affine points x||y (canonical little-endian, (0,0) = infinity, the layout plonk_g1_msm
 *                        returns) -> count x 32 bytes.  PLONK_ERR_ARG if a coordinate is not below p.
 *   plonk_g1_decompress  count x 32 bytes -> x||y canonical little-endian + status[i]: 0 ok, 1 malformed (flag bits 00,
 *                        x >= p, infinity with x != 0), 2 x^3 + 3 is not a square (not a curve point); y by one
 *                        exponentiation per point, (p + 1) / 4, on the device.                                        */
int plonk_g1_compress(plonk_ctx* ctx, const uint8_t* h_xy_le, size_t count, uint8_t* h_out32);
int plonk_g1_decompress(plonk_ctx* ctx, const uint8_t* h_in32, size_t count, uint8_t* h_out_xy_le, uint8_t* h_status);

/* ---- multi-GPU: gather of finished proofs, RCCL over xGMI ---------------------------------------------
 * Proofs are independent (prover.py:51-84 has no cross-proof state), so N GPUs prove disjoint index sets with no
 * data-path exchange; the one collective is an all-gather of the results, 768 bytes per proof.
 * One process per GPU.  Rank 0 calls plonk_comm_unique_id and passes the 128 bytes to the other ranks out of band;
 * every rank then calls plonk_comm_create (ncclCommInitRank on the context's device).  librccl is loaded on the
 * first of these calls, never for single-GPU use.
 *   plonk_gather_results  h_recv[r * bytes_per_rank ...] = rank r's h_send, for all ranks (ncclAllGather, uint8)
 *   plonk_comm_max_f64    *inout = max over ranks (the benchmark clock: slowest rank); plonk_comm_barrier likewise */
#define PLONK_COMM_ID_BYTES 128
typedef struct plonk_comm plonk_comm;
int plonk_comm_unique_id(uint8_t out_id[PLONK_COMM_ID_BYTES]);
int plonk_comm_create(plonk_ctx* ctx, const uint8_t id[PLONK_COMM_ID_BYTES], int rank, int world, plonk_comm** out);
int plonk_comm_destroy(plonk_comm* comm);
int plonk_comm_size(const plonk_comm* comm, int* out_rank, int* out_world);
int plonk_gather_results(plonk_comm* comm, const uint8_t* h_send, size_t bytes_per_rank, uint8_t* h_recv);
/* the same gather for the proofs of `n_provers` lock-step provers of this rank without a host round trip: each packs its
 * resident batch (768-byte records, or 480-byte compressed ones: plonk_prover_download_compressed's form) and its status
 * bytes into the send buffer on its own stream, one ncclAllGather, one copy to the host.  h_recv = world blocks of
 * [n_provers * batch records | n_provers * batch status bytes, padded to a multiple of 16].                        */
int plonk_gather_proofs_device(plonk_comm* comm, plonk_prover* const* provers, size_t n_provers, size_t batch, int compressed,
                               uint8_t* h_recv);
int plonk_comm_max_f64(plonk_comm* comm, double* inout);
int plonk_comm_barrier(plonk_comm* comm);
/* Deadlines.  A collective waits for every rank, so one dead or stuck rank would block the others for ever.  Every wait behind
 * a collective (and ncclCommInitRank inside plonk_comm_create) therefore has a deadline, after which the communicator is aborted
 * (ncclCommAbort) and the call returns PLONK_ERR_TIMEOUT, plonk_last_error naming the operation, the rank and the device; every
 * later call on that communicator returns PLONK_ERR_STATE.  plonk_comm_set_default_timeout: the process default, taken by
 * plonk_comm_create (initially $PLONK_COMM_TIMEOUT_S, else 600 s; 0 = wait for ever); plonk_comm_set_timeout: one communicator. */
int plonk_comm_set_default_timeout(double seconds);
int plonk_comm_set_timeout(plonk_comm* comm, double seconds);
/* out_row[p] = 1 iff `device` can map the memory of device p (hipDeviceCanAccessPeer; 1 on the diagonal), p < min(cap, devices):
 * what RCCL's peer-to-peer transport over xGMI needs between two ranks of a node (bench.py --preflight prints it per rank). */
int plonk_device_peer_access(int device, int* out_row, size_t cap);
/* device time of the last plonk_gather_proofs_device on this communicator: the ncclAllGather itself and the copy of every
 * rank's records to the host (HIP events on the communicator's stream).                                              */
int plonk_comm_last_gather_ms(plonk_comm* comm, float* out_allgather_ms, float* out_to_host_ms);
/* which RCCL this process talks to: the file the loaded ncclGetUniqueId lives in (load order: $PLONK_RCCL_LIB, then
 * $ROCM_PATH/lib/librccl.so.1, /opt/rocm/lib/librccl.so.1, then the sonames), ncclGetVersion's code, and how many RCCL
 * collectives / point-to-point groups `comm` has issued (comm may be NULL; any out pointer may be NULL).              */
int plonk_comm_info(const plonk_comm* comm, char* out_path, size_t path_cap, int* out_version, uint64_t* out_collectives);
/* ---- one transform across the GPUs of a communicator (four-step NTT) ----------------------
 * Polynomial.fft / ifft (poly.py:113-148) for N = 2^log_n = R1 R2 points that need not fit one GPU (log_n = 18, 20, 22,
 * 24, 26: R1 x R2 = 2^9 x 2^9, 2^11 x 2^9, 2^11 x 2^11, 2^13 x 2^11, 2^13 x 2^13), over W = 2^k ranks, W <= min(R1, R2) / 32.
 *   input   rank g holds the columns c = g R2/W .. (g+1) R2/W - 1 of x[i1 R2 + c], stored [R1][R2/W]   (N / W elements)
 *   output  rank g holds the frequencies k = k1 + R1 k2 with k1 = g R1/W .. (g+1) R1/W - 1, stored [R2][R1/W]
 * (the inverse transform includes 1/N).  plonk_fr_ntt_distributed = plonk_fr_ntt_dist_columns (local R1-point transforms
 * and the w_N^(c k1) twiddles) -> plonk_comm_all_to_all (d_recv block r = block `rank` of rank r's d_send: grouped
 * ncclSend / ncclRecv over xGMI, N / W^2 elements per pair) -> plonk_fr_ntt_dist_rows (local R2-point transforms).
 * The two local steps are exported so the exchange can run over another transport (tests: sockets on CPU).        */
/* (the column pass's output is an intermediate: packed residues in [0, 2r), not canonical values — it is what
 * plonk_fr_ntt_dist_rows, directly or after the all-to-all, takes as input) */
int plonk_fr_ntt_dist_columns(plonk_ctx* ctx, const void* d_in, void* d_out, unsigned log_n, unsigned log_world, unsigned rank, int inverse);
int plonk_fr_ntt_dist_rows(plonk_ctx* ctx, const void* d_in, void* d_out, unsigned log_n, unsigned log_world, unsigned rank, int inverse);
int plonk_comm_all_to_all(plonk_comm* comm, const void* d_send, void* d_recv, size_t bytes_per_peer);
int plonk_fr_ntt_distributed(plonk_comm* comm, const void* d_in, void* d_out, unsigned log_n, int inverse);

/* ---- verifier support: pairing-product check (host CPU) ------------------------------------------------
 * Replaces the `b.pairing(...)` comparisons of the verifier, TESTING_verifier_DO_NOT_OPEN.py:148-160, 237-262 /
 * verifier.py:40-92 (py_ecc.bn128.pairing): out_ok = 1 iff prod_i e(P_i, Q_i) == 1 in GT.  P_i in G1: affine
 * canonical x||y LE (64 B) + identity flag; Q_i in G2: affine canonical x.c0||x.c1||y.c0||y.c1 LE (128 B, py_ecc's FQ2
 * coefficient order; all zero = identity).  A check e(A, X) == e(B, Y) is asked as e(A, X) e(-B, Y) == 1.  Runs on the
 * host (the verifier is off the prover hot path); no context needed.                          */
int plonk_pairing_check(const uint8_t* g1_xy_le, const uint8_t* g1_is_identity, const uint8_t* g2_le, size_t count, int* out_ok);

/* ---- Fiat-Shamir transcript (host) ----------------------------------------------------------------
 * Replaces `merlin.MerlinTranscript` (third-party) as subclassed by transcript.py:58-60:
 *   plonk_transcript_new              MerlinTranscript(label)             (prover.py:53 uses b"plonk")
 *   plonk_transcript_append_message   Transcript.append / append_message  transcript.py:59-67
 *   plonk_transcript_challenge_bytes  MerlinTranscript.challenge_bytes    transcript.py:71
 *   plonk_transcript_challenge_scalar Transcript.get_and_append_challenge transcript.py:69-75
 *                                     (255 PRF bytes -> big-endian int mod r, retried while zero, then
 *                                     the bytes are appended under the same label); canonical LE out. */
typedef struct plonk_transcript plonk_transcript;
int plonk_transcript_new(const uint8_t* label, size_t label_len, plonk_transcript** out);
int plonk_transcript_clone(const plonk_transcript* t, plonk_transcript** out);
int plonk_transcript_free(plonk_transcript* t);
int plonk_transcript_append_message(plonk_transcript* t, const uint8_t* label, size_t label_len,
                                    const uint8_t* msg, size_t msg_len);
int plonk_transcript_challenge_bytes(plonk_transcript* t, const uint8_t* label, size_t label_len, uint8_t* out,
                                     size_t n);
int plonk_transcript_challenge_scalar(plonk_transcript* t, const uint8_t* label, size_t label_len,
                                      uint8_t out_le32[32]);

/* ---- DIAGNOSTICS: tuning knobs (tests, A/B runs, bench.py) --------------------------------------------------------
 * Not part of the drop-in surface: nothing in the reference corresponds to them and no caller needs them — the defaults are
 * what measured fastest on MI355X.  They are PER-CONTEXT MUTABLE STATE (a setting stays until it is set back to 0), so a
 * caller that uses one should scope it: plonkathon_amd.Context.tuning() is a `with` block that restores the defaults, and it
 * is how bench.py sets them.  (Resource POLICIES — how much HBM the library may take — are with their subsystems:
 * plonk_ntt_set_table_budget, plonk_msm_lookup_configure.) */
/* tuning / test knob (0 = default): LDS tile = 2^tile_log elements (<= 12), sizes <= 2^single_pass_log
 * (<= 11) run as one pass, larger sizes split into passes of radix <= 2^radix_log (<= 10). */
int plonk_ntt_configure(plonk_ctx* ctx, unsigned tile_log, unsigned single_pass_log, unsigned radix_log);
/* kernel family: 0 = auto (what measures fastest on MI355X: the in-register "wave" kernels — 2, 4 or 8 elements per
 * thread as signed 29-bit limbs, digits exchanged inside a wave by DPP / v_permlane16_swap / v_permlane32_swap, LDS only
 * across waves — for 2^7 .. 2^13 in one launch and 2^14 .. 2^26 as two passes of those; the LDS kernel, radix-2 stages,
 * below 2^7 and above 2^26), 1 = the LDS kernel at every size, 4 = the same (A/B runs; it used to choose among two LDS
 * kernels), 5 = the wave kernels wherever they apply, whatever plonk_ntt_configure says; 6 / 7 = as 5, but never / always
 * on the two-element "latency" forms (2^9, and the splits of 2^14 .. 2^18 built on 2^7 and 2^9) that 0 and 5 pick for
 * calls of at most 2^18 elements; 8 = as 5, with 2^12 on its 1024-thread, 4-element form instead of 512 threads x 8
 * elements (which 0 and 5 use wherever 2^12 is not a column pass on the one-table inter-pass twiddles).
 * (2, the Stockham LDS kernel, and 3 are retired.) */
int plonk_ntt_select_kernel(plonk_ctx* ctx, unsigned kind);
/* two-pass wave transforms N = R1 R2 (R1-point column transforms, then R2-point row transforms): log2 R1 for one
 * log2 N in [16, 26]; 0 = the default (as square as possible).  Both factors must lie in 2^8 .. 2^13.  A/B runs, tests. */
int plonk_ntt_set_split(plonk_ctx* ctx, unsigned log_n, unsigned log_r1);
/* log2 R1 of the split in force for 2^log_n (ctx NULL: the library default, which is what the distributed transform
 * uses on every rank — its column / frequency-strided layouts are [R1][R2 / W] and [R2][R1 / W]) */
int plonk_ntt_get_split(plonk_ctx* ctx, unsigned log_n, unsigned* out_log_r1);
/* tuning knobs (0 = library default): window bits c and window-groups per MSM (bucket method) */
int plonk_msm_configure(plonk_ctx* ctx, unsigned window_bits, unsigned groups);

/* ---- timing support for bench.py (HIP events on the context's stream) ------------------------ */
int plonk_timer_start(plonk_ctx* ctx);
/* Per-kernel profiling: while enabled, each launch of an instrumented kernel ("msm_accumulate",
 * "ntt_pass", ...) is bracketed by HIP events on the context's stream.  plonk_profile_read sums the
 * durations, launch count and algorithmic bytes (SURVEY.md 8(d) figures) recorded for one kernel. */
int plonk_profile_enable(plonk_ctx* ctx, int on);
int plonk_profile_read(plonk_ctx* ctx, const char* kernel, double* total_ms, uint64_t* launches, double* algo_bytes);
int plonk_profile_reset(plonk_ctx* ctx);
int plonk_timer_stop_ms(plonk_ctx* ctx, float* out_ms);

#ifdef __cplusplus
}
#endif
#endif /* PLONK_HIP_H */
