// ntt_bls.hip — the standalone NTT over the BLS12-381 scalar field
//   r = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001   (255 bits, 2-adicity 32, generator 7)
// on the same wave kernels as BN254 Fr (ntt_wave.h, instantiated for BlsFrParams) behind plonk_bls_fr_*.
//
// Why it exists: BASELINE.json's north_star quotes a standalone NTT metric on this field.  The reference has no such field
// (curve.py:2, 10-11: BN254 throughout; poly.py:113-148 is the transform), so nothing here takes part in proof parity; the
// convention is the reference's transform with the field swapped — X[k] = sum_j x[j] w^(jk), natural order in and out,
// w = 7^((r-1)/N) (the bls12_381 crate's ROOT_OF_UNITY squared down), the inverse with w^-1 and 1/N — and the checker is
// oracle/c's oracle_bls_fr_ntt, pinned by definition (tests/test_oracle_c.py).
//
// What differs from BN254 for the kernels: R / m = 2^261 / r = 70.7 instead of 169, so a Shoup product of a multiplicand a
// lands in (-(1 + |a|/R) m, (2 + |a|/R) m) = (-1.26 m, 2.26 m) for the NTT's |a| < 18.1 m — inside the interval the
// butterflies assume; the emulator build asserts it on every multiplication (fpl.h).  Sizes: the wave kernels' own,
// 2^8 .. 2^13 in one launch and 2^14 .. 2^26 in two.
#include <string>

#include "ntt_wave_host.h"
#include "bls12_381_constants.h"

typedef Fp<BlsFrParams> BlsFr;

static inline dim3 bls_grid_for(size_t n, unsigned block) {
    size_t g = (n + block - 1) / block;
    return dim3((unsigned)(g > 2048 ? 2048 : (g ? g : 1)));
}

// out[i] = first * base^i: each lane raises base to its chunk start, then walks 16 steps
__global__ void bls_fr_powers_kernel(BlsFr base, BlsFr first, BlsFr* out, size_t n) {
    const size_t nchunks = (n + 15) / 16;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < nchunks; c += (size_t)gridDim.x * blockDim.x) {
        const size_t start = c * 16;
        BlsFr cur = fp_mul(first, fp_pow_u64(base, (uint64_t)start));
        for (int k = 0; k < 16 && start + k < n; k++) {
            fp_store(out + start + k, cur);
            cur = fp_mul(cur, base);
        }
    }
}

// canonical <-> Montgomery form; to_mont also records the first element >= r in *bad (~0: none)
__global__ void bls_fr_convert_kernel(const BlsFr* in, BlsFr* out, size_t n, int to_mont, unsigned long long* bad) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const BlsFr a = fp_load(in + i);
        if (to_mont) {
            uint32_t br = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) (void)fp_sbb(a.v[k], BlsFrParams::mod(k), br);  // a - r borrows  <=>  a < r
            if (!br) atomicMin(bad, (unsigned long long)i);
        }
        fp_store(out + i, to_mont ? fp_to_mont(a) : fp_from_mont(a));
    }
}

static int bls_powers(plonk_ctx* ctx, const BlsFr& base, const BlsFr& first, BlsFr* out, size_t n) {
    if (!n) return PLONK_OK;
    PLONK_LAUNCH(bls_fr_powers_kernel, bls_grid_for((n + 15) / 16, 64), dim3(64), 0, ctx->stream, base, first, out, n);
    PLONK_CHECK_HIP(hipGetLastError());
    return PLONK_OK;
}

static BlsFr bls_root_of_unity(unsigned log_n, bool inverse) {
    BlsFr w;
    for (int i = 0; i < 8; i++) w.v[i] = inverse ? BlsFrRoots::w32_inv(i) : BlsFrRoots::w32(i);
    for (unsigned i = log_n; i < PLONK_BLS_FR_TWO_ADICITY; i++) w = fp_sqr(w);
    return w;
}

// a cached table of powers of a root (the packed source of a limb table; kept: the conversion kernel reads it asynchronously)
static int bls_power_table(plonk_ctx* ctx, std::map<unsigned, void*>& cache, unsigned key, const BlsFr& base, size_t n, const BlsFr** out) {
    auto it = cache.find(key);
    if (it == cache.end()) {
        void* d = nullptr;
        if (!plonk_dev_malloc(&d, n * sizeof(BlsFr))) {
            plonk_set_error("hipMalloc of a %zu-entry twiddle table failed", n);
            return PLONK_ERR_NOMEM;
        }
        ctx->owned.push_back(d);
        PLONK_TRY(bls_powers(ctx, base, fp_one<BlsFrParams>(), (BlsFr*)d, n));
        it = cache.emplace(key, d).first;
    }
    *out = (const BlsFr*)it->second;
    return PLONK_OK;
}

struct BlsFrField {
    typedef BlsFrParams P;
    static BlsFr root_of_unity(unsigned log_n, bool inverse) { return bls_root_of_unity(log_n, inverse); }
    static BlsFr from_u64(uint64_t x) {
        BlsFr a = fp_zero<BlsFrParams>();
        a.v[0] = (uint32_t)x;
        a.v[1] = (uint32_t)(x >> 32);
        return fp_to_mont(a);
    }
    static int packed_roots(plonk_ctx* ctx, unsigned log_n, bool inverse, const BlsFr** out) {
        return bls_power_table(ctx, ctx->wave_bls.packed[0], log_n | (inverse ? 256u : 0u), bls_root_of_unity(log_n, inverse), (size_t)1 << log_n, out);
    }
    static int packed_lo_hi(plonk_ctx* ctx, unsigned log_n, bool inverse, const BlsFr** lo, const BlsFr** hi) {
        const unsigned key = log_n | (inverse ? 256u : 0u), log_lo = log_n < NTT_TW_LO_LOG ? log_n : NTT_TW_LO_LOG;
        const BlsFr w = bls_root_of_unity(log_n, inverse);
        BlsFr whi = w;
        for (unsigned i = 0; i < NTT_TW_LO_LOG; i++) whi = fp_sqr(whi);
        PLONK_TRY(bls_power_table(ctx, ctx->wave_bls.packed[1], key, w, (size_t)1 << log_lo, lo));
        return bls_power_table(ctx, ctx->wave_bls.packed[2], key, whi, log_n > NTT_TW_LO_LOG ? (size_t)1 << (log_n - NTT_TW_LO_LOG) : 1, hi);
    }
    static int powers(plonk_ctx* ctx, const BlsFr& base, const BlsFr& first, BlsFr* out, size_t n) { return bls_powers(ctx, base, first, out, n); }
    static WaveTables& tables(plonk_ctx* ctx) { return ctx->wave_bls; }
};

// ---- C-ABI ---------------------------------------------------------------------------------------------------------------
extern "C" {

int plonk_bls_fr_upload(plonk_ctx* ctx, void* d_dst, const uint8_t* h_src_le32, size_t count) {
    PLONK_REQUIRE(ctx && (count == 0 || (d_dst && h_src_le32)), PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    if (!count) return PLONK_OK;
    void* flag;
    PLONK_TRY(ctx_scratch(ctx, 3, 64, &flag));
    PLONK_CHECK_HIP(hipMemcpyAsync(d_dst, h_src_le32, count * 32, hipMemcpyHostToDevice, ctx->stream));
    PLONK_CHECK_HIP(hipMemsetAsync(flag, 0xff, sizeof(unsigned long long), ctx->stream));
    PLONK_LAUNCH(bls_fr_convert_kernel, bls_grid_for(count, 256), dim3(256), 0, ctx->stream, (const BlsFr*)d_dst, (BlsFr*)d_dst, count, 1, (unsigned long long*)flag);
    PLONK_CHECK_HIP(hipGetLastError());
    unsigned long long first_bad = 0;
    PLONK_CHECK_HIP(hipMemcpyAsync(&first_bad, flag, sizeof first_bad, hipMemcpyDeviceToHost, ctx->stream));
    PLONK_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    PLONK_REQUIRE(first_bad == ~0ull, PLONK_ERR_ARG, "element %llu is not a canonical BLS12-381 Fr value (>= r)", first_bad);
    return PLONK_OK;
}

int plonk_bls_fr_download(plonk_ctx* ctx, uint8_t* h_dst_le32, const void* d_src, size_t count) {
    PLONK_REQUIRE(ctx && (count == 0 || (h_dst_le32 && d_src)), PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    if (!count) return PLONK_OK;
    void* tmp;
    PLONK_TRY(ctx_scratch(ctx, 3, count * 32, &tmp));
    PLONK_LAUNCH(bls_fr_convert_kernel, bls_grid_for(count, 256), dim3(256), 0, ctx->stream, (const BlsFr*)d_src, (BlsFr*)tmp, count, 0, (unsigned long long*)nullptr);
    PLONK_CHECK_HIP(hipGetLastError());
    PLONK_CHECK_HIP(hipMemcpyAsync(h_dst_le32, tmp, count * 32, hipMemcpyDeviceToHost, ctx->stream));
    PLONK_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return PLONK_OK;
}

// first * base^i, i < n, cached per context beside BN254's tables (key prefix "bls"; ctx_destroy frees the map's values)
static int bls_power_table(plonk_ctx* ctx, const BlsFr& base, const BlsFr& first, size_t n, const BlsFr** out) {
    std::string key("bls");
    key.append((const char*)base.v, 32);
    key.append((const char*)first.v, 32);
    key.append((const char*)&n, sizeof n);
    auto it = ctx->power_tables.find(key);
    if (it == ctx->power_tables.end()) {
        void* p = nullptr;
        if (!plonk_dev_malloc(&p, n * sizeof(BlsFr))) {
            plonk_set_error("hipMalloc of a %zu-entry power table failed", n);
            return PLONK_ERR_NOMEM;
        }
        int rc = bls_powers(ctx, base, first, (BlsFr*)p, n);
        if (rc != PLONK_OK) {
            hipFree(p);
            return rc;
        }
        it = ctx->power_tables.emplace(key, (Fr*)p).first;
    }
    *out = (const BlsFr*)it->second;
    return PLONK_OK;
}

// canonical little-endian bytes -> Montgomery form; false if the value is not below r
static bool bls_from_le32(const uint8_t* b, BlsFr* out) {
    BlsFr a;
    memcpy(a.v, b, 32);
    for (int i = 7; i >= 0; i--) {
        if (a.v[i] < BlsFrParams::mod(i)) {
            *out = fp_to_mont(a);
            return true;
        }
        if (a.v[i] > BlsFrParams::mod(i)) return false;
    }
    return false;
}

static bool bls_size_ok(const plonk_ctx* ctx, unsigned log_n) {
    unsigned r1, r2;
    return ntt_wave_plan(ctx, log_n, false, &r1, &r2);
}

// poly.py:156-163 over this field: n Lagrange values -> coefficients (ifft) -> c_i offset^i, zero-padded to 4n -> forward transform:
// the values on the coset offset * <w_4n>.  The scaling and the padding ride in the first load of the 4n-point transform.
int plonk_bls_fr_coset_extend(plonk_ctx* ctx, const void* d_in, void* d_out, unsigned log_n, const uint8_t offset_le32[32], size_t batch) {
    PLONK_REQUIRE(ctx && d_in && d_out && offset_le32, PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    if (!batch) return PLONK_OK;
    BlsFr off;
    PLONK_REQUIRE(bls_from_le32(offset_le32, &off), PLONK_ERR_ARG, "offset is not a canonical BLS12-381 Fr value");
    PLONK_REQUIRE(bls_size_ok(ctx, log_n) && bls_size_ok(ctx, log_n + 2), PLONK_ERR_ARG,
                  "the BLS12-381 coset extension covers 2^8 .. 2^24 values (4n <= 2^26), not 2^%u", log_n);
    const size_t n = (size_t)1 << log_n, big = n << 2;
    void* coeffs;
    PLONK_TRY(ctx_scratch(ctx, 2, batch * n * sizeof(BlsFr), &coeffs));
    PLONK_TRY(wave_run<BlsFrField>(ctx, (const BlsFr*)d_in, (BlsFr*)coeffs, log_n, true, batch, n, n, n, nullptr, nullptr, true));
    const BlsFr* pw;
    PLONK_TRY(bls_power_table(ctx, off, fp_one<BlsFrParams>(), n, &pw));
    return wave_run<BlsFrField>(ctx, (const BlsFr*)coeffs, (BlsFr*)d_out, log_n + 2, false, batch, n, n, big, pw, nullptr, false);
}

// poly.py:169-177 over this field: M coset values -> ifft -> v_i (1/offset)^i; the 1/M of the ifft rides in the power table
int plonk_bls_fr_coset_to_coeffs(plonk_ctx* ctx, const void* d_in, void* d_out, unsigned log_m, const uint8_t offset_le32[32], size_t batch) {
    PLONK_REQUIRE(ctx && d_in && d_out && offset_le32, PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    if (!batch) return PLONK_OK;
    BlsFr off;
    PLONK_REQUIRE(bls_from_le32(offset_le32, &off), PLONK_ERR_ARG, "offset is not a canonical BLS12-381 Fr value");
    PLONK_REQUIRE(bls_size_ok(ctx, log_m), PLONK_ERR_ARG, "the BLS12-381 transform covers 2^8 .. 2^26 points (the wave kernels' sizes), not 2^%u", log_m);
    const size_t M = (size_t)1 << log_m;
    const BlsFr* pw;
    PLONK_TRY(bls_power_table(ctx, fp_inv(off), fp_inv(BlsFrField::from_u64((uint64_t)M)), M, &pw));
    return wave_run<BlsFrField>(ctx, (const BlsFr*)d_in, (BlsFr*)d_out, log_m, true, batch, M, M, M, nullptr, pw, false);
}

int plonk_bls_fr_ntt(plonk_ctx* ctx, const void* d_in, void* d_out, unsigned log_n, int inverse, size_t batch) {
    PLONK_REQUIRE(ctx && d_in && d_out, PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    if (!batch) return PLONK_OK;
    unsigned r1, r2;
    PLONK_REQUIRE(ntt_wave_plan(ctx, log_n, false, &r1, &r2), PLONK_ERR_ARG,
                  "the BLS12-381 transform covers 2^8 .. 2^26 points (the wave kernels' sizes), not 2^%u", log_n);
    const size_t N = (size_t)1 << log_n;
    return wave_run<BlsFrField>(ctx, (const BlsFr*)d_in, (BlsFr*)d_out, log_n, inverse != 0, batch, N, N, N, nullptr, nullptr, inverse != 0);
}

}  // extern "C"
