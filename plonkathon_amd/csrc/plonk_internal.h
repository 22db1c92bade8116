// plonk_internal.h — context object and helpers shared by the C-ABI implementation files.
#pragma once
#include <stdarg.h>
#include <stdio.h>

#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/plonk_hip.h"
#include "fp.h"
#include "g1.h"

void plonk_set_error(const char* fmt, ...);

#define PLONK_CHECK_HIP(expr)                                                                     \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            plonk_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return PLONK_ERR_HIP;                                                                 \
        }                                                                                         \
    } while (0)

// Allocation that may fail and be recovered from (the caller frees something and retries, or falls back to a smaller table):
// HIP parks the failure in its thread-local last-error slot, where the next `PLONK_CHECK_HIP(hipGetLastError())` after a kernel
// launch would find it and report a stale out-of-memory as PLONK_ERR_HIP.  The slot is emptied here, at the failing call.
template <class T> static inline bool plonk_dev_malloc(T** p, size_t bytes) {
    if (hipMalloc((void**)p, bytes) == hipSuccess) return true;
    (void)hipGetLastError();
    *p = nullptr;
    return false;
}
template <class T> static inline bool plonk_host_malloc(T** p, size_t bytes) {
    if (hipHostMalloc((void**)p, bytes) == hipSuccess) return true;
    (void)hipGetLastError();
    *p = nullptr;
    return false;
}

#define PLONK_REQUIRE(cond, code, ...)    \
    do {                                  \
        if (!(cond)) {                    \
            plonk_set_error(__VA_ARGS__); \
            return (code);                \
        }                                 \
    } while (0)

// HIP's current device is per thread; every entry point that touches a context makes its device current first
// (a process may hold contexts on several GPUs).
int plonk_use_device(int device);
#define PLONK_ENTER(ctx)                                                  \
    do {                                                                  \
        int rc_enter_ = plonk_use_device((ctx)->device);                  \
        if (rc_enter_ != PLONK_OK) return rc_enter_;                      \
    } while (0)

#define PLONK_TRY(expr)          \
    do {                         \
        int rc_ = (expr);        \
        if (rc_ != PLONK_OK) return rc_; \
    } while (0)

// device tables of the wave NTT kernels for one field, cached per context (the memory is in plonk_ctx::owned)
struct WaveTables {
    std::map<unsigned, int32_t*> prog, lo, hi;  // keys: log2(size) | inverse << 8 (| 512: hi scaled by 1/N)
    std::map<unsigned, int32_t*> interpass;     // full inter-pass tables: log2 N | inverse << 8 | scaled << 9 | log2 R1 << 12
    const int32_t* jm = nullptr;                // fpl_reduce_small's multiples of the modulus
    bool attr_set[4] = {false, false, false, false};   // hipFuncSetAttribute is per device: tracked per context (E = 4, E = 8, E = 4 column kernel, the 512-thread 2^12 kernel)
    std::map<unsigned, void*> packed[3];        // packed source tables (full, lo, hi) of a field that has no cache of its own (BLS12-381 Fr)
    std::map<unsigned, std::shared_ptr<void>> plans;  // WavePlan<P> per (log2 N | inverse << 8 | 1/N << 9 | full table << 10): ntt_wave_host.h
    unsigned plan_epoch = 0;                    // plonk_ctx::ntt_cfg_epoch the plans were built under
};

struct NttTables {
    // keys are log2(size) | inverse << 8
    std::map<unsigned, Fr*> small;   // w_R^k, k < R/2              (per-pass LDS twiddles)
    std::map<unsigned, Fr*> lo, hi;  // w_N^e = lo[e & 1023] * hi[e >> 10]   (inter-pass twiddles)
    std::map<unsigned, Fr*> full;    // w_N^k, k < N                 (barycentric / permutation argument)
    WaveTables wave;                 // the wave NTT kernels' tables for BN254 Fr (ntt_wave_host.h)
};

#define MSM_TABLE_COMB 1
#define MSM_TABLE_WINDOWS 2
// One lookup table per (process, device, base set), shared by every plonk_srs / context / stream that
// loads the same bases (msm.hip keeps the registry; reference counted, freed with its last plonk_srs).
struct MsmLookupTable {
    int device = 0;
    uint64_t key = 0;        // FNV-1a of the host bytes the bases were loaded from
    size_t n_points = 0;
    unsigned kind = 0;       // MSM_TABLE_COMB (msm_comb.h: bits = teeth h, windows = columns a) or MSM_TABLE_WINDOWS (bits = c, windows = W)
    unsigned bits = 0, windows = 0;
    unsigned top_bits = 0, top_g = 0;  // comb with top tables (msm_comb.h: windows = floor(254 / bits)): R and the bases per group; 0, 0 = none
    G1Affine* data = nullptr;
    size_t bytes = 0;
    double build_s = 0;      // wall time of the build (reported by bench.py)
    int refs = 0;
};

struct plonk_srs {
    int device = 0;
    uint64_t content_key = 0;   // FNV-1a of the loaded bytes: identifies the base set in the lookup-table registry
    size_t n_points = 0;
    G1Affine* bases = nullptr;  // device, Montgomery coordinates (.ptau layout)
    unsigned window_bits = 0;   // c of the current window table (0 = not built)
    unsigned n_windows = 0;
    G1Affine* table = nullptr;  // device: table[w * n_points + i] = 2^(c*w) * bases[i]
    // lookup MSM (msm.hip): every multiple d * 2^(c*w) * bases[i], d = 1 .. 2^(c-1), resident in HBM
    bool fixed = false;          // a reusable SRS (plonk_srs_load_ptau): worth a big table
    unsigned lookup_bits = 0;    // teeth h of the comb table / window bits c of the window table (0 = none)
    unsigned lookup_windows = 0; // additions per base: columns a of the comb / windows W
    unsigned lookup_kind = 0;    // MSM_TABLE_COMB / MSM_TABLE_WINDOWS (0 = none)
    unsigned lookup_top_bits = 0, lookup_top_g = 0;  // the attached comb's top tables (0, 0 = none)
    bool lookup_failed = false;  // an automatic build did not fit: do not retry on every call
    G1Affine* lookup = nullptr;  // comb: lookup[(i << (h - 1)) + idx]; windows: lookup[((w * n_points + i) << (c - 1)) + d - 1]   (= shared->data)
    MsmLookupTable* shared = nullptr;
    // Lagrange-basis SRS (setup.py:66-72 without the ifft): lagrange[log_n] = [L_i(tau)]_1, i < 2^log_n, built on
    // demand by an EC inverse NTT of the first 2^log_n bases (msm.hip); each is a plonk_srs of its own.
    std::map<unsigned, plonk_srs*> lagrange;
    plonk_srs* parent = nullptr;  // a Lagrange view's SRS (its table is charged against the parent's budget)
};

#define PLONK_SCRATCH_SLOTS 4
struct plonk_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;  // host-to-device staging that overlaps the compute stream (created on first use: ctx_copy_stream)
    NttTables tw;
    std::map<std::string, Fr*> power_tables;  // cached coset-offset power tables, keyed by (offset, n, kind)
    std::vector<void*> owned;                 // device allocations released with the context
    void* scratch[PLONK_SCRATCH_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    size_t scratch_bytes[PLONK_SCRATCH_SLOTS] = {0, 0, 0, 0};
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    Fr host_tmp[16];                 // small host-side staging that must outlive an asynchronous copy (every user synchronises before returning)
    unsigned msm_window_bits = 0, msm_groups = 0;
    int msm_lookup_mode = 0;         // 0 auto (fixed SRS only), 1 off, 2 force msm_lookup_bits for every base set
    unsigned msm_lookup_bits = 0;    // 0 = the table with the fewest additions per base that fits the budget
    bool msm_lookup_top = false;     // with explicit msm_lookup_bits: the comb WITH top tables (plonk_msm_lookup_configure mode | 32)
    unsigned msm_lookup_kind = MSM_TABLE_COMB;  // layout of the tables this context builds (plonk_msm_lookup_configure: mode | 16 = window tables)
    size_t msm_lookup_budget = 0;    // bytes; 0 = default (PLONK_MSM_TABLE_GB if set, else min(device memory / 16, free memory / 4): msm.hip)
    bool ntt_attr_set = false, msm_attr_set = false;  // hipFuncSetAttribute is per device: tracked per context
    // per-kernel HIP-event profiling (bench.py roofline): one record per instrumented launch
    struct ProfRec { const char* name; hipEvent_t a, b; double algo_bytes; };
    bool profiling = false;
    std::vector<ProfRec> prof;
    std::vector<hipEvent_t> event_pool;
    unsigned ntt_tile_log = 12, ntt_single_log = 11, ntt_radix_log = 10;
    bool ntt_adaptive_tiles = true;
    WaveTables wave_bls;  // the same for the standalone BLS12-381 Fr transform (ntt_bls.hip)
    size_t ntt_table_budget = (size_t)4 << 30, ntt_tables_bytes = 0;  // full inter-pass twiddle tables (80 B per point and direction): plonk_ntt_set_table_budget
    unsigned char ntt_split[32] = {0};  // plonk_ntt_set_split: log2 R1 of the two-pass wave plan per log2 N (0 = default)
    unsigned ntt_cfg_epoch = 0;  // bumped by every plonk_ntt_* setter: cached launch plans are rebuilt
    unsigned ntt_kind = 0;  // 0 = auto (wave kernels where they apply, else the LDS kernel), 1 / 4 = the LDS kernel, 5 = force wave, 6 / 7 = force wave without / with the two-element latency forms, 8 = force wave with 2^12 on 1024 threads
};

// scratch slot use: 0 = NTT inter-pass buffer, 1 = MSM digits/partials, 2-3 = API-level temporaries
int ctx_scratch(plonk_ctx* ctx, int slot, size_t bytes, void** out);
int ctx_copy_stream(plonk_ctx* ctx);
// bracket an instrumented launch: prof_begin before, prof_end after (no-ops unless profiling)
int prof_begin(plonk_ctx* ctx, const char* name, double algo_bytes);
int prof_end(plonk_ctx* ctx);

// ---- enqueue helpers shared between translation units (all on ctx->stream) ---------------------
// fr_ops.hip
int k_fr_to_mont(plonk_ctx*, const Fr* in, Fr* out, size_t n);
int k_fr_from_mont(plonk_ctx*, const Fr* in, Fr* out, size_t n);
int k_fr_to_mont_checked(plonk_ctx*, Fr* data, size_t n, unsigned long long* d_first_bad);
int k_fr_pointwise(plonk_ctx*, int op, const Fr* a, const Fr* b, Fr* out, size_t n);
int k_fr_pointwise_scalar(plonk_ctx*, int op, const Fr* a, const Fr& s_mont, Fr* out, size_t n, size_t limit);
int k_fr_batch_inverse(plonk_ctx*, const Fr* in, Fr* out, size_t n);
int k_fr_powers(plonk_ctx*, const Fr& base_mont, const Fr& first_mont, Fr* out, size_t n);
int k_fr_rotate(plonk_ctx*, const Fr* in, Fr* out, size_t n, size_t shift, size_t batch);
int k_fr_barycentric(plonk_ctx*, const Fr* vals, const Fr* roots, unsigned log_n, const Fr* xs_dev, size_t x_stride,
                     const Fr& n_inv_mont, Fr* out_dev, size_t n_polys);
int k_fr_barycentric_ptrs(plonk_ctx*, const Fr* const* polys, const Fr* roots, unsigned log_n, const Fr* xs_dev, const Fr& n_inv_mont,
                          Fr* out_dev, size_t n_polys);  // <= 16 polynomials in separate buffers, one point each
int k_fr_lincomb(plonk_ctx*, const Fr* const* terms, const Fr* scalars_mont, unsigned n_terms, const Fr& constant_mont, Fr* out, size_t n);  // <= 20 terms
int k_fr_count_diff(plonk_ctx*, const Fr* a, const Fr* b_or_null, size_t n, unsigned long long* d_count);
// api.hip
Fr fr_from_le32(const uint8_t* b);                 // canonical little-endian bytes -> Montgomery form (host)
bool le32_below_modulus(const uint8_t* b, bool fq);
int get_power_table(plonk_ctx*, const Fr& base, const Fr& first, size_t n, const Fr** out);  // first * base^i, cached per context
// ntt.hip
Fr host_root_of_unity(unsigned log_n, bool inverse);
// fan (optional): every batch entry is transformed `count` times — copy f reads in + f in_stride, writes out + f out_stride
// and takes its scaling vector at in_scale / out_scale + f scale_stride (elements): one launch on the wave kernels' single-pass
// sizes (the prover: a coefficient vector evaluated on three cosets), a loop of calls elsewhere
struct NttFan { unsigned count, in_stride, out_stride, scale_stride; };
int ntt_run(plonk_ctx*, const Fr* in, Fr* out, unsigned log_n, bool inverse, size_t batch, size_t in_len,
            size_t in_bstride, size_t out_bstride, const Fr* in_scale, const Fr* out_scale, bool scale_by_n_inv, const NttFan* fan = nullptr);
int ntt_get_roots(plonk_ctx*, unsigned log_n, bool inverse, const Fr** table_full);
int ntt_dist_plan(unsigned log_n, unsigned log_w, unsigned* log_r1, unsigned* log_r2);
bool ntt_wave_plan(const plonk_ctx* ctx /* null: the default splits */, unsigned log_n, bool latency, unsigned* log_r1, unsigned* log_r2);
int ntt_dist_columns(plonk_ctx*, const Fr* in, Fr* out, unsigned log_n, unsigned log_w, unsigned rank, bool inverse);
int ntt_dist_rows(plonk_ctx*, const Fr* in, Fr* out, unsigned log_n, unsigned log_w, unsigned rank, bool inverse);
// msm.hip
int msm_build_table(plonk_ctx*, plonk_srs*, unsigned c);
int msm_lagrange_srs(plonk_ctx*, plonk_srs*, unsigned log_n, plonk_srs** out);  // owned by (and freed with) the parent
void g1_batch_to_affine(plonk_ctx*, const G1Xyzz* in, G1Affine* out, size_t n);  // enqueue: XYZZ -> affine (Montgomery), identity -> (0, 0)
// g1_ntt.hip: the same Lagrange-basis points by an inverse DFT over the group (n log n group operations)
int g1_lagrange_by_ntt(plonk_ctx*, const plonk_srs*, unsigned log_n, G1Affine* d_bases_out);
void msm_srs_release(plonk_srs*);  // drops the reference on the shared lookup table
int msm_lookup_info(const plonk_srs*, unsigned* bits, size_t* bytes, double* build_s, int* sharers);
int msm_lookup_layout(const plonk_srs*, unsigned* kind, unsigned* additions_per_base);
int msm_lookup_top(const plonk_srs*, unsigned* top_bits, unsigned* bases_per_group);
bool msm_comb_takes_top(unsigned teeth);  // msm.hip: a comb of this many teeth can take top tables
uint64_t plonk_fnv1a64(const void* data, size_t n);
// MSM m reads its scalars at d_scalars + (m % inner) * stride + (m / inner) * outer_stride (inner = 0: inner = M)
int msm_run_device(plonk_ctx*, plonk_srs*, const Fr* d_scalars, size_t n, size_t M, size_t stride, Fq* d_out_xy,
                   uint8_t* d_flags, size_t inner = 0, size_t outer_stride = 0);
// prover.hip
struct plonk_prover;
int prover_pack_device(plonk_prover* p, size_t B, int compressed, uint8_t* d_proofs, uint8_t* d_status, hipEvent_t done);
plonk_ctx* prover_ctx(plonk_prover* p);
