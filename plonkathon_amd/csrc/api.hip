// api.hip — the extern "C" surface declared in include/plonk_hip.h: context, device memory,
// conversions and the thin wrappers that turn one reference-level operation into kernel launches.
#include <stdlib.h>
#include <string.h>

#include "plonk_internal.h"

static thread_local char g_err[512] = "";

void plonk_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

// No cache of our own: the integrator (or another library on this thread) may call hipSetDevice between two entry
// points, so HIP's own per-thread record is the only one that can be trusted.  hipGetDevice is a thread-local read.
int plonk_use_device(int device) {
    int current = -1;
    if (hipGetDevice(&current) == hipSuccess && current == device) return PLONK_OK;
    PLONK_CHECK_HIP(hipSetDevice(device));
    return PLONK_OK;
}

int ctx_scratch(plonk_ctx* ctx, int slot, size_t bytes, void** out) {
    if (ctx->scratch_bytes[slot] < bytes) {
        if (ctx->scratch[slot]) {
            // the old buffer may still be in use by enqueued kernels
            PLONK_CHECK_HIP(hipStreamSynchronize(ctx->stream));
            hipFree(ctx->scratch[slot]);
            ctx->scratch[slot] = nullptr;
            ctx->scratch_bytes[slot] = 0;
        }
        size_t want = bytes + bytes / 4;
        void* p = nullptr;
        if (!plonk_dev_malloc(&p, want)) {
            plonk_set_error("hipMalloc of %zu scratch bytes failed", want);
            return PLONK_ERR_NOMEM;
        }
        ctx->scratch[slot] = p;
        ctx->scratch_bytes[slot] = want;
    }
    *out = ctx->scratch[slot];
    return PLONK_OK;
}

int ctx_copy_stream(plonk_ctx* ctx) {
    if (!ctx->copy_stream) PLONK_CHECK_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    return PLONK_OK;
}

static int prof_event(plonk_ctx* ctx, hipEvent_t* e) {
    if (!ctx->event_pool.empty()) {
        *e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return PLONK_OK;
    }
    PLONK_CHECK_HIP(hipEventCreate(e));
    return PLONK_OK;
}

int prof_begin(plonk_ctx* ctx, const char* name, double algo_bytes) {
    if (!ctx->profiling) return PLONK_OK;
    plonk_ctx::ProfRec r;
    r.name = name;
    r.algo_bytes = algo_bytes;
    PLONK_TRY(prof_event(ctx, &r.a));
    PLONK_TRY(prof_event(ctx, &r.b));
    PLONK_CHECK_HIP(hipEventRecord(r.a, ctx->stream));
    ctx->prof.push_back(r);
    return PLONK_OK;
}

int prof_end(plonk_ctx* ctx) {
    if (!ctx->profiling || ctx->prof.empty()) return PLONK_OK;
    PLONK_CHECK_HIP(hipEventRecord(ctx->prof.back().b, ctx->stream));
    return PLONK_OK;
}

uint64_t plonk_fnv1a64(const void* data, size_t n) {
    const uint8_t* p = (const uint8_t*)data;
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; i++) h = (h ^ p[i]) * 0x100000001b3ull;
    return h;
}

Fr fr_from_le32(const uint8_t* b) {
    Fr a;
    memcpy(a.v, b, 32);
    return fp_to_mont(a);
}

bool le32_below_modulus(const uint8_t* b, bool fq) {
    uint32_t v[8];
    memcpy(v, b, 32);
    for (int i = 7; i >= 0; i--) {
        uint32_t m = fq ? FqParams::mod(i) : FrParams::mod(i);
        if (v[i] < m) return true;
        if (v[i] > m) return false;
    }
    return false;
}

// .ptau coordinates are Montgomery residues with R = 2^256 (setup.py:39-40); ours use R = 2^PLONK_MONT_BITS.
__global__ void fq_rescale_kernel(Fq* data, size_t n, unsigned doublings) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Fq x = fp_load(data + i);
        for (unsigned k = 0; k < doublings; k++) x = fp_dbl(x);
        fp_store(data + i, x);
    }
}

extern "C" {

const char* plonk_last_error(void) { return g_err; }
int plonk_abi_version(void) { return PLONK_ABI_VERSION; }

int plonk_device_count(int* out_count) {
    PLONK_REQUIRE(out_count, PLONK_ERR_ARG, "out_count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) n = 0;
    *out_count = n;
    return PLONK_OK;
}

int plonk_ctx_create(int device, plonk_ctx** out_ctx) {
    PLONK_REQUIRE(out_ctx, PLONK_ERR_ARG, "out_ctx is NULL");
    int n = 0;
    PLONK_CHECK_HIP(hipGetDeviceCount(&n));
    PLONK_REQUIRE(device >= 0 && device < n, PLONK_ERR_ARG, "device %d out of range (%d visible)", device, n);
    PLONK_TRY(plonk_use_device(device));
    plonk_ctx* ctx = new plonk_ctx();
    ctx->device = device;
    if (const char* e = getenv("PLONK_NTT_ADAPTIVE_TILES")) ctx->ntt_adaptive_tiles = atoi(e) != 0;  // A/B knob (default on)
    PLONK_CHECK_HIP(hipStreamCreate(&ctx->stream));
    PLONK_CHECK_HIP(hipEventCreate(&ctx->ev_a));
    PLONK_CHECK_HIP(hipEventCreate(&ctx->ev_b));
    *out_ctx = ctx;
    return PLONK_OK;
}

int plonk_ctx_destroy(plonk_ctx* ctx) {
    if (!ctx) return PLONK_OK;
    plonk_use_device(ctx->device);
    hipStreamSynchronize(ctx->stream);
    for (void* p : ctx->owned) hipFree(p);
    for (auto& kv : ctx->power_tables) hipFree(kv.second);
    for (int s = 0; s < PLONK_SCRATCH_SLOTS; s++)
        if (ctx->scratch[s]) hipFree(ctx->scratch[s]);
    for (auto& r : ctx->prof) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    for (auto e : ctx->event_pool) hipEventDestroy(e);
    hipEventDestroy(ctx->ev_a);
    hipEventDestroy(ctx->ev_b);
    if (ctx->copy_stream) hipStreamDestroy(ctx->copy_stream);
    hipStreamDestroy(ctx->stream);
    delete ctx;
    return PLONK_OK;
}

int plonk_ctx_sync(plonk_ctx* ctx) {
    PLONK_REQUIRE(ctx, PLONK_ERR_ARG, "ctx is NULL");
    PLONK_ENTER(ctx);
    PLONK_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return PLONK_OK;
}

int plonk_ctx_device_name(plonk_ctx* ctx, char* buf, size_t buf_len) {
    PLONK_REQUIRE(ctx && buf && buf_len, PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    hipDeviceProp_t prop;
    PLONK_CHECK_HIP(hipGetDeviceProperties(&prop, ctx->device));
    snprintf(buf, buf_len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return PLONK_OK;
}

// ---- memory ------------------------------------------------------------------------------------
// page-locked host memory: what an asynchronous host-to-device copy needs to be asynchronous
int plonk_host_alloc(plonk_ctx* ctx, size_t bytes, void** out_hptr) {
    PLONK_REQUIRE(ctx && out_hptr, PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    void* p = nullptr;
    if (!plonk_host_malloc(&p, bytes ? bytes : 1)) {
        plonk_set_error("hipHostMalloc of %zu bytes failed", bytes);
        return PLONK_ERR_NOMEM;
    }
    *out_hptr = p;
    return PLONK_OK;
}
int plonk_host_free(plonk_ctx* ctx, void* hptr) {
    PLONK_REQUIRE(ctx, PLONK_ERR_ARG, "ctx is NULL");
    PLONK_ENTER(ctx);
    if (hptr) PLONK_CHECK_HIP(hipHostFree(hptr));
    return PLONK_OK;
}

int plonk_mem_alloc(plonk_ctx* ctx, size_t bytes, void** out_dptr) {
    PLONK_REQUIRE(ctx && out_dptr, PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    void* p = nullptr;
    if (!plonk_dev_malloc(&p, bytes ? bytes : 32)) {
        plonk_set_error("hipMalloc(%zu) failed", bytes);
        return PLONK_ERR_NOMEM;
    }
    *out_dptr = p;
    return PLONK_OK;
}
int plonk_mem_info(plonk_ctx* ctx, size_t* out_free, size_t* out_total) {
    PLONK_REQUIRE(ctx && out_free && out_total, PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    PLONK_CHECK_HIP(hipMemGetInfo(out_free, out_total));
    return PLONK_OK;
}
int plonk_mem_free(plonk_ctx* ctx, void* dptr) {
    PLONK_REQUIRE(ctx, PLONK_ERR_ARG, "ctx is NULL");
    PLONK_ENTER(ctx);
    if (dptr) PLONK_CHECK_HIP(hipFree(dptr));
    return PLONK_OK;
}
int plonk_mem_h2d(plonk_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
    PLONK_REQUIRE(ctx && (bytes == 0 || (d_dst && h_src)), PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    if (!bytes) return PLONK_OK;
    PLONK_CHECK_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    PLONK_CHECK_HIP(hipStreamSynchronize(ctx->stream));  // the caller's buffer is not retained
    return PLONK_OK;
}
int plonk_mem_d2h(plonk_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
    PLONK_REQUIRE(ctx && (bytes == 0 || (h_dst && d_src)), PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    if (!bytes) return PLONK_OK;
    PLONK_CHECK_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    PLONK_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return PLONK_OK;
}
int plonk_mem_d2d(plonk_ctx* ctx, void* d_dst, const void* d_src, size_t bytes) {
    PLONK_REQUIRE(ctx && (bytes == 0 || (d_dst && d_src)), PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    if (!bytes) return PLONK_OK;
    PLONK_CHECK_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return PLONK_OK;
}
int plonk_mem_zero(plonk_ctx* ctx, void* d_dst, size_t bytes) {
    PLONK_REQUIRE(ctx && (bytes == 0 || d_dst), PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    if (!bytes) return PLONK_OK;
    PLONK_CHECK_HIP(hipMemsetAsync(d_dst, 0, bytes, ctx->stream));
    return PLONK_OK;
}

// ---- Fr vectors --------------------------------------------------------------------------------
int plonk_fr_upload(plonk_ctx* ctx, void* d_dst, const uint8_t* h_src_le32, size_t count) {
    PLONK_REQUIRE(ctx && (count == 0 || (d_dst && h_src_le32)), PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    if (!count) return PLONK_OK;
    // the range check `value < r` runs on the device, fused with the Montgomery conversion (a host loop over a
    // batch of witnesses cost as much as packing them)
    void* flag;
    PLONK_TRY(ctx_scratch(ctx, 3, 64, &flag));
    PLONK_CHECK_HIP(hipMemcpyAsync(d_dst, h_src_le32, count * 32, hipMemcpyHostToDevice, ctx->stream));
    PLONK_TRY(k_fr_to_mont_checked(ctx, (Fr*)d_dst, count, (unsigned long long*)flag));
    unsigned long long first_bad = 0;
    PLONK_CHECK_HIP(hipMemcpyAsync(&first_bad, flag, sizeof first_bad, hipMemcpyDeviceToHost, ctx->stream));
    PLONK_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    PLONK_REQUIRE(first_bad == ~0ull, PLONK_ERR_ARG, "element %llu is not a canonical Fr value (>= r)", first_bad);
    return PLONK_OK;
}

int plonk_fr_download(plonk_ctx* ctx, uint8_t* h_dst_le32, const void* d_src, size_t count) {
    PLONK_REQUIRE(ctx && (count == 0 || (h_dst_le32 && d_src)), PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    if (!count) return PLONK_OK;
    void* tmp;
    PLONK_TRY(ctx_scratch(ctx, 3, count * 32, &tmp));
    PLONK_TRY(k_fr_from_mont(ctx, (const Fr*)d_src, (Fr*)tmp, count));
    PLONK_CHECK_HIP(hipMemcpyAsync(h_dst_le32, tmp, count * 32, hipMemcpyDeviceToHost, ctx->stream));
    PLONK_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return PLONK_OK;
}

// ---- NTT family --------------------------------------------------------------------------------
int plonk_ntt_configure(plonk_ctx* ctx, unsigned tile_log, unsigned single_pass_log, unsigned radix_log) {
    PLONK_REQUIRE(ctx, PLONK_ERR_ARG, "ctx is NULL");
    PLONK_ENTER(ctx);
    if (!tile_log) tile_log = 12;
    if (!single_pass_log) single_pass_log = 11;
    if (!radix_log) radix_log = 10;
    PLONK_REQUIRE(tile_log >= 2 && tile_log <= 12, PLONK_ERR_ARG, "tile_log must be in [2, 12]");
    PLONK_REQUIRE(single_pass_log <= 11 && single_pass_log <= tile_log, PLONK_ERR_ARG, "single_pass_log must be <= min(11, tile_log)");
    PLONK_REQUIRE(radix_log >= 1 && radix_log <= 10 && radix_log <= tile_log, PLONK_ERR_ARG, "radix_log must be in [1, min(10, tile_log)]");
    ctx->ntt_tile_log = tile_log;
    ctx->ntt_single_log = single_pass_log;
    ctx->ntt_radix_log = radix_log;
    ctx->ntt_cfg_epoch++;
    return PLONK_OK;
}

int plonk_ntt_select_kernel(plonk_ctx* ctx, unsigned kind) {
    PLONK_REQUIRE(ctx, PLONK_ERR_ARG, "ctx is NULL");
    PLONK_ENTER(ctx);
    PLONK_REQUIRE(kind <= 8 && kind != 2 && kind != 3, PLONK_ERR_ARG,
                  "kernel kind must be 0 (auto), 1 or 4 (the LDS kernel: radix-2 stages), 5 (in-register wave kernels wherever they apply), 6 (wave kernels, "
                  "never the two-element latency forms), 7 (wave kernels, the latency forms wherever they exist) or 8 (wave kernels, 2^12 on its 1024-thread form)");
    ctx->ntt_kind = kind;
    ctx->ntt_cfg_epoch++;
    return PLONK_OK;
}

int plonk_ntt_set_split(plonk_ctx* ctx, unsigned log_n, unsigned log_r1) {
    PLONK_REQUIRE(ctx, PLONK_ERR_ARG, "ctx is NULL");
    PLONK_ENTER(ctx);
    PLONK_REQUIRE(log_n >= 14 && log_n <= 26, PLONK_ERR_ARG, "two-pass wave transforms cover 2^14 .. 2^26 (got 2^%u)", log_n);
    PLONK_REQUIRE(log_r1 == 0 || (log_r1 >= 7 && log_r1 <= 13 && log_n - log_r1 >= 7 && log_n - log_r1 <= 13), PLONK_ERR_ARG,
                  "2^%u = 2^%u x 2^%u: both factors must lie in 2^7 .. 2^13", log_n, log_r1, log_n - log_r1);
    ctx->ntt_split[log_n] = (unsigned char)log_r1;
    ctx->ntt_cfg_epoch++;
    return PLONK_OK;
}

int plonk_ntt_get_split(plonk_ctx* ctx, unsigned log_n, unsigned* out_log_r1) {
    PLONK_REQUIRE(out_log_r1, PLONK_ERR_ARG, "bad argument");
    unsigned r1 = 0, r2 = 0;
    PLONK_REQUIRE(ntt_wave_plan(ctx, log_n, false, &r1, &r2) && r2, PLONK_ERR_ARG, "2^%u is not a two-pass wave transform (2^14 .. 2^26)", log_n);
    *out_log_r1 = r1;
    return PLONK_OK;
}

int plonk_ntt_set_table_budget(plonk_ctx* ctx, size_t bytes) {
    PLONK_REQUIRE(ctx, PLONK_ERR_ARG, "ctx is NULL");
    ctx->ntt_table_budget = bytes;
    ctx->ntt_cfg_epoch++;
    return PLONK_OK;
}

int plonk_fr_ntt(plonk_ctx* ctx, const void* d_in, void* d_out, unsigned log_n, int inverse, size_t batch) {
    PLONK_REQUIRE(ctx && d_in && d_out, PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    const size_t N = (size_t)1 << log_n;
    return ntt_run(ctx, (const Fr*)d_in, (Fr*)d_out, log_n, inverse != 0, batch, N, N, N, nullptr, nullptr, inverse != 0);
}

int plonk_fr_ntt_dist_columns(plonk_ctx* ctx, const void* d_in, void* d_out, unsigned log_n, unsigned log_world, unsigned rank, int inverse) {
    PLONK_REQUIRE(ctx && d_in && d_out && d_in != d_out && rank < (1u << log_world), PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    return ntt_dist_columns(ctx, (const Fr*)d_in, (Fr*)d_out, log_n, log_world, rank, inverse != 0);
}

int plonk_fr_ntt_dist_rows(plonk_ctx* ctx, const void* d_in, void* d_out, unsigned log_n, unsigned log_world, unsigned rank, int inverse) {
    PLONK_REQUIRE(ctx && d_in && d_out && d_in != d_out && rank < (1u << log_world), PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    return ntt_dist_rows(ctx, (const Fr*)d_in, (Fr*)d_out, log_n, log_world, rank, inverse != 0);
}

// power table first * base^i, i < n, cached per (base, first, n)
extern "C++" int get_power_table(plonk_ctx* ctx, const Fr& base, const Fr& first, size_t n, const Fr** out) {
    std::string key((const char*)base.v, 32);
    key.append((const char*)first.v, 32);
    key.append((const char*)&n, sizeof n);
    auto it = ctx->power_tables.find(key);
    if (it == ctx->power_tables.end()) {
        if (ctx->power_tables.size() >= 64) {  // bound the cache: offsets are per-proof challenges in API mode
            PLONK_CHECK_HIP(hipStreamSynchronize(ctx->stream));
            for (auto& kv : ctx->power_tables) hipFree(kv.second);
            ctx->power_tables.clear();
        }
        void* p = nullptr;
        if (!plonk_dev_malloc(&p, n * sizeof(Fr))) {
            plonk_set_error("hipMalloc of a %zu-entry power table failed", n);
            return PLONK_ERR_NOMEM;
        }
        PLONK_TRY(k_fr_powers(ctx, base, first, (Fr*)p, n));
        it = ctx->power_tables.emplace(key, (Fr*)p).first;
    }
    *out = it->second;
    return PLONK_OK;
}

int plonk_fr_coset_ntt_from_coeffs(plonk_ctx* ctx, const void* d_coeffs, void* d_out, unsigned log_n,
                                   unsigned log_expand, const uint8_t offset_le32[32], size_t batch) {
    PLONK_REQUIRE(ctx && d_coeffs && d_out && offset_le32, PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    PLONK_REQUIRE(le32_below_modulus(offset_le32, false), PLONK_ERR_ARG, "offset is not a canonical Fr value");
    const size_t n = (size_t)1 << log_n, big = n << log_expand;
    const Fr* pw;
    PLONK_TRY(get_power_table(ctx, fr_from_le32(offset_le32), fp_one<FrParams>(), n, &pw));
    // coefficients c_i * offset^i, zero-padded to 2^(log_n+log_expand), forward NTT     poly.py:160-163
    return ntt_run(ctx, (const Fr*)d_coeffs, (Fr*)d_out, log_n + log_expand, false, batch, n, n, big, pw, nullptr, false);
}

int plonk_fr_coset_extend(plonk_ctx* ctx, const void* d_in, void* d_out, unsigned log_n,
                          const uint8_t offset_le32[32], size_t batch) {
    PLONK_REQUIRE(ctx && d_in && d_out && offset_le32, PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    const size_t n = (size_t)1 << log_n;
    void* coeffs;
    PLONK_TRY(ctx_scratch(ctx, 2, batch * n * sizeof(Fr), &coeffs));
    // poly.py:159 — ifft to coefficients first
    PLONK_TRY(ntt_run(ctx, (const Fr*)d_in, (Fr*)coeffs, log_n, true, batch, n, n, n, nullptr, nullptr, true));
    return plonk_fr_coset_ntt_from_coeffs(ctx, coeffs, d_out, log_n, 2, offset_le32, batch);
}

int plonk_fr_coset_to_coeffs(plonk_ctx* ctx, const void* d_in, void* d_out, unsigned log_m,
                             const uint8_t offset_le32[32], size_t batch) {
    PLONK_REQUIRE(ctx && d_in && d_out && offset_le32, PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    PLONK_REQUIRE(le32_below_modulus(offset_le32, false), PLONK_ERR_ARG, "offset is not a canonical Fr value");
    const size_t M = (size_t)1 << log_m;
    // poly.py:172-176 — ifft, then v_i * (1/offset)^i; the 1/M of the ifft is folded into the table
    Fr inv_off = fp_inv(fr_from_le32(offset_le32));
    Fr m_mont = fp_zero<FrParams>();
    m_mont.v[0] = (uint32_t)M;
    m_mont.v[1] = (uint32_t)((uint64_t)M >> 32);
    Fr m_inv = fp_inv(fp_to_mont(m_mont));
    const Fr* pw;
    PLONK_TRY(get_power_table(ctx, inv_off, m_inv, M, &pw));
    return ntt_run(ctx, (const Fr*)d_in, (Fr*)d_out, log_m, true, batch, M, M, M, nullptr, pw, false);
}

// ---- pointwise ---------------------------------------------------------------------------------
int plonk_fr_pointwise(plonk_ctx* ctx, int op, const void* d_a, const void* d_b, void* d_out, size_t count) {
    PLONK_REQUIRE(ctx && (count == 0 || (d_a && d_b && d_out)), PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    PLONK_REQUIRE(op >= PLONK_OP_ADD && op <= PLONK_OP_DIV, PLONK_ERR_ARG, "unknown pointwise op %d", op);
    if (op == PLONK_OP_DIV) {
        void* inv;
        PLONK_TRY(ctx_scratch(ctx, 2, count * sizeof(Fr), &inv));
        PLONK_TRY(k_fr_batch_inverse(ctx, (const Fr*)d_b, (Fr*)inv, count));
        return k_fr_pointwise(ctx, PLONK_OP_MUL, (const Fr*)d_a, (const Fr*)inv, (Fr*)d_out, count);
    }
    return k_fr_pointwise(ctx, op, (const Fr*)d_a, (const Fr*)d_b, (Fr*)d_out, count);
}

int plonk_fr_scalar_op(plonk_ctx* ctx, int op, const void* d_a, const uint8_t scalar_le32[32], void* d_out,
                       size_t count, int constant_term_only) {
    PLONK_REQUIRE(ctx && scalar_le32 && (count == 0 || (d_a && d_out)), PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    PLONK_REQUIRE(op >= PLONK_OP_ADD && op <= PLONK_OP_DIV, PLONK_ERR_ARG, "unknown scalar op %d", op);
    PLONK_REQUIRE(le32_below_modulus(scalar_le32, false), PLONK_ERR_ARG, "scalar is not a canonical Fr value");
    Fr s = fr_from_le32(scalar_le32);
    if (op == PLONK_OP_DIV) {
        s = fp_inv(s);  // x / 0 == 0
        op = PLONK_OP_MUL;
    }
    return k_fr_pointwise_scalar(ctx, op, (const Fr*)d_a, s, (Fr*)d_out, count, constant_term_only ? 1 : count);
}

int plonk_fr_rotate(plonk_ctx* ctx, const void* d_in, void* d_out, size_t count, size_t shift) {
    PLONK_REQUIRE(ctx && d_in && d_out && d_in != d_out, PLONK_ERR_ARG, "bad argument (rotate is out-of-place)");
    PLONK_ENTER(ctx);
    PLONK_REQUIRE(shift < count, PLONK_ERR_ARG, "shift %zu must be < length %zu", shift, count);
    return k_fr_rotate(ctx, (const Fr*)d_in, (Fr*)d_out, count, shift, 1);
}

int plonk_fr_batch_inverse(plonk_ctx* ctx, const void* d_in, void* d_out, size_t count) {
    PLONK_REQUIRE(ctx && (count == 0 || (d_in && d_out)), PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    return k_fr_batch_inverse(ctx, (const Fr*)d_in, (Fr*)d_out, count);
}

int plonk_fr_powers(plonk_ctx* ctx, const uint8_t first_le32[32], const uint8_t base_le32[32], size_t count, void* d_out) {
    PLONK_REQUIRE(ctx && first_le32 && base_le32 && (count == 0 || d_out), PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    PLONK_REQUIRE(le32_below_modulus(first_le32, false) && le32_below_modulus(base_le32, false), PLONK_ERR_ARG,
                  "first / base is not a canonical Fr value");
    return k_fr_powers(ctx, fr_from_le32(base_le32), fr_from_le32(first_le32), (Fr*)d_out, count);
}

int plonk_fr_equal(plonk_ctx* ctx, const void* d_a, const void* d_b, size_t count, int* out_equal) {
    PLONK_REQUIRE(ctx && out_equal && (count == 0 || d_a), PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    *out_equal = 1;
    if (!count) return PLONK_OK;
    void* flag;
    PLONK_TRY(ctx_scratch(ctx, 3, 64, &flag));
    PLONK_TRY(k_fr_count_diff(ctx, (const Fr*)d_a, (const Fr*)d_b, count, (unsigned long long*)flag));
    unsigned long long diff = 0;
    PLONK_CHECK_HIP(hipMemcpyAsync(&diff, flag, sizeof diff, hipMemcpyDeviceToHost, ctx->stream));
    PLONK_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    *out_equal = diff == 0;
    return PLONK_OK;
}

int plonk_fr_barycentric(plonk_ctx* ctx, const void* d_vals, unsigned log_n, const uint8_t x_le32[32],
                         uint8_t out_le32[32]) {
    PLONK_REQUIRE(ctx && d_vals && x_le32 && out_le32, PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    PLONK_REQUIRE(log_n <= PLONK_FR_TWO_ADICITY, PLONK_ERR_ARG, "size 2^%u exceeds the 2-adicity of Fr", log_n);
    PLONK_REQUIRE(le32_below_modulus(x_le32, false), PLONK_ERR_ARG, "x is not a canonical Fr value");
    const Fr* roots;
    PLONK_TRY(ntt_get_roots(ctx, log_n, false, &roots));
    void* tmp;
    PLONK_TRY(ctx_scratch(ctx, 2, 2 * sizeof(Fr), &tmp));
    Fr* dx = (Fr*)tmp;
    Fr x = fr_from_le32(x_le32);
    PLONK_CHECK_HIP(hipMemcpyAsync(dx, &x, sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    PLONK_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    Fr nn = fp_zero<FrParams>();
    nn.v[0] = (uint32_t)((uint64_t)1 << log_n);
    Fr n_inv = fp_inv(fp_to_mont(nn));
    PLONK_TRY(k_fr_barycentric(ctx, (const Fr*)d_vals, roots, log_n, dx, 0, n_inv, dx + 1, 1));
    PLONK_TRY(k_fr_from_mont(ctx, dx + 1, dx + 1, 1));
    PLONK_CHECK_HIP(hipMemcpyAsync(out_le32, dx + 1, 32, hipMemcpyDeviceToHost, ctx->stream));
    PLONK_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return PLONK_OK;
}

// n_polys <= 16 polynomials of 2^log_n Lagrange values in separate buffers, polynomial k at its own point xs[k]: ONE kernel
// (a workgroup per polynomial) and ONE host synchronisation — the six evaluations of round 4 (prover.py:228-239) cost six
// round trips through plonk_fr_barycentric
int plonk_fr_barycentric_many(plonk_ctx* ctx, size_t n_polys, const void* const* d_vals, unsigned log_n, const uint8_t* xs_le32, uint8_t* out_le32) {
    PLONK_REQUIRE(ctx && d_vals && xs_le32 && out_le32 && n_polys >= 1 && n_polys <= 16, PLONK_ERR_ARG, "bad argument (1 .. 16 polynomials)");
    PLONK_ENTER(ctx);
    PLONK_REQUIRE(log_n <= PLONK_FR_TWO_ADICITY, PLONK_ERR_ARG, "size 2^%u exceeds the 2-adicity of Fr", log_n);
    Fr* xs = ctx->host_tmp;  // (context-owned: an asynchronous copy from pageable memory must not outlive its source)
    const Fr* polys[16];
    for (size_t k = 0; k < n_polys; k++) {
        PLONK_REQUIRE(d_vals[k], PLONK_ERR_ARG, "polynomial %zu is NULL", k);
        PLONK_REQUIRE(le32_below_modulus(xs_le32 + 32 * k, false), PLONK_ERR_ARG, "x[%zu] is not a canonical Fr value", k);
        xs[k] = fr_from_le32(xs_le32 + 32 * k);
        polys[k] = (const Fr*)d_vals[k];
    }
    const Fr* roots;
    PLONK_TRY(ntt_get_roots(ctx, log_n, false, &roots));
    void* tmp;
    PLONK_TRY(ctx_scratch(ctx, 2, 32 * sizeof(Fr), &tmp));
    Fr* dx = (Fr*)tmp;
    PLONK_CHECK_HIP(hipMemcpyAsync(dx, xs, n_polys * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    Fr nn = fp_zero<FrParams>();
    nn.v[0] = (uint32_t)((uint64_t)1 << log_n);
    const Fr n_inv = fp_inv(fp_to_mont(nn));
    PLONK_TRY(k_fr_barycentric_ptrs(ctx, polys, roots, log_n, dx, n_inv, dx + 16, n_polys));
    PLONK_TRY(k_fr_from_mont(ctx, dx + 16, dx + 16, n_polys));
    PLONK_CHECK_HIP(hipMemcpyAsync(out_le32, dx + 16, 32 * n_polys, hipMemcpyDeviceToHost, ctx->stream));
    PLONK_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return PLONK_OK;
}

// d_out[i] = constant + sum_k scalars[k] * d_terms[k][i], i < count; n_terms <= 20; d_out may be one of the terms
int plonk_fr_lincomb(plonk_ctx* ctx, size_t n_terms, const void* const* d_terms, const uint8_t* scalars_le32, const uint8_t constant_le32[32],
                     void* d_out, size_t count) {
    PLONK_REQUIRE(ctx && n_terms <= 20 && (n_terms == 0 || (d_terms && scalars_le32)) && constant_le32 && (count == 0 || d_out), PLONK_ERR_ARG,
                  "bad argument (at most 20 terms)");
    PLONK_ENTER(ctx);
    Fr sc[20];
    const Fr* terms[20];
    for (size_t k = 0; k < n_terms; k++) {
        PLONK_REQUIRE(d_terms[k] || !count, PLONK_ERR_ARG, "term %zu is NULL", k);
        PLONK_REQUIRE(le32_below_modulus(scalars_le32 + 32 * k, false), PLONK_ERR_ARG, "scalar %zu is not a canonical Fr value", k);
        sc[k] = fr_from_le32(scalars_le32 + 32 * k);
        terms[k] = (const Fr*)d_terms[k];
    }
    PLONK_REQUIRE(le32_below_modulus(constant_le32, false), PLONK_ERR_ARG, "the constant is not a canonical Fr value");
    return k_fr_lincomb(ctx, terms, sc, (unsigned)n_terms, fr_from_le32(constant_le32), (Fr*)d_out, count);
}

// ---- G1 ----------------------------------------------------------------------------------------
static int srs_alloc(plonk_ctx* ctx, size_t n_points, plonk_srs** out) {
    plonk_srs* s = new plonk_srs();
    s->n_points = n_points;
    s->device = ctx->device;
    void* p = nullptr;
    if (!plonk_dev_malloc(&p, n_points * sizeof(G1Affine))) {
        delete s;
        plonk_set_error("hipMalloc of %zu G1 bases failed", n_points);
        return PLONK_ERR_NOMEM;
    }
    s->bases = (G1Affine*)p;
    *out = s;
    (void)ctx;
    return PLONK_OK;
}

// Registry key of a base set: FNV-1a of the loaded bytes (a 64-bit hash is NOT an identity: the lookup-table registry compares
// the bases themselves before it shares a table, msm.hip: lut_verified).
// TEST HOOK, unsupported: with PLONK_ENABLE_TEST_HOOKS=1, PLONK_TEST_SRS_KEY replaces the hash by a constant so that the tests can
// file two different base sets under ONE key and watch the registry tell them apart.  The salt (which keeps .ptau and affine
// loads in separate key spaces) stays in force, and without the first variable the second is ignored: a stray setting in a
// production environment cannot collapse the registry.
static uint64_t srs_content_key(const void* bytes, size_t n, uint64_t salt) {
    const char* hooks = getenv("PLONK_ENABLE_TEST_HOOKS");  // (an SRS load is rare: two getenv calls cost nothing here)
    if (hooks && !strcmp(hooks, "1"))
        if (const char* e = getenv("PLONK_TEST_SRS_KEY")) return strtoull(e, nullptr, 0) ^ salt;
    return plonk_fnv1a64(bytes, n) ^ salt;
}

int plonk_srs_load_ptau(plonk_ctx* ctx, const uint8_t* g1_mont_le, size_t n_points, plonk_srs** out_srs) {
    PLONK_REQUIRE(ctx && g1_mont_le && out_srs && n_points, PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    for (size_t i = 0; i < 2 * n_points; i++)  // setup.py:36 `assert max(values) < b.field_modulus`
        PLONK_REQUIRE(le32_below_modulus(g1_mont_le + 32 * i, true), PLONK_ERR_ARG,
                      "SRS coordinate %zu is >= the BN254 base-field modulus", i);
    plonk_srs* s;
    PLONK_TRY(srs_alloc(ctx, n_points, &s));
    PLONK_CHECK_HIP(hipMemcpyAsync(s->bases, g1_mont_le, n_points * 64, hipMemcpyHostToDevice, ctx->stream));
    {
        size_t nc = 2 * n_points;
        unsigned g = (unsigned)((nc + 255) / 256);
        if (g > 2048) g = 2048;
        PLONK_LAUNCH(fq_rescale_kernel, dim3(g), dim3(256), 0, ctx->stream, (Fq*)s->bases, nc, (unsigned)(PLONK_MONT_BITS - 256));
        PLONK_CHECK_HIP(hipGetLastError());
    }
    PLONK_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    s->fixed = true;
    s->content_key = srs_content_key(g1_mont_le, n_points * 64, 0);
    *out_srs = s;
    return PLONK_OK;
}

int plonk_srs_load_affine(plonk_ctx* ctx, const uint8_t* xy_le, size_t n_points, plonk_srs** out_srs) {
    PLONK_REQUIRE(ctx && xy_le && out_srs && n_points, PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    std::vector<Fq> host(2 * n_points);
    for (size_t i = 0; i < 2 * n_points; i++) {
        PLONK_REQUIRE(le32_below_modulus(xy_le + 32 * i, true), PLONK_ERR_ARG,
                      "coordinate %zu is >= the BN254 base-field modulus", i);
        Fq a;
        memcpy(a.v, xy_le + 32 * i, 32);
        host[i] = fp_to_mont(a);  // host-side: API-mode lincombs are small
    }
    plonk_srs* s;
    PLONK_TRY(srs_alloc(ctx, n_points, &s));
    PLONK_CHECK_HIP(hipMemcpyAsync(s->bases, host.data(), n_points * 64, hipMemcpyHostToDevice, ctx->stream));
    PLONK_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    s->content_key = srs_content_key(xy_le, n_points * 64, 0x9e3779b97f4a7c15ull);  // canonical bytes: a different key space
    *out_srs = s;
    return PLONK_OK;
}

int plonk_srs_free(plonk_ctx* ctx, plonk_srs* srs) {
    PLONK_REQUIRE(ctx, PLONK_ERR_ARG, "ctx is NULL");
    PLONK_ENTER(ctx);
    if (!srs) return PLONK_OK;
    PLONK_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    for (auto& kv : srs->lagrange) plonk_srs_free(ctx, kv.second);
    if (srs->bases) hipFree(srs->bases);
    if (srs->table) hipFree(srs->table);
    msm_srs_release(srs);  // the lookup table is shared: freed with its last user
    delete srs;
    return PLONK_OK;
}

int plonk_srs_lagrange(plonk_ctx* ctx, plonk_srs* srs, unsigned log_n, plonk_srs** out_view) {
    PLONK_REQUIRE(ctx && srs && out_view, PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    return msm_lagrange_srs(ctx, srs, log_n, out_view);
}

int plonk_srs_size(const plonk_srs* srs, size_t* out_n) {
    PLONK_REQUIRE(srs && out_n, PLONK_ERR_ARG, "bad argument");
    *out_n = srs->n_points;
    return PLONK_OK;
}

int plonk_msm_configure(plonk_ctx* ctx, unsigned window_bits, unsigned groups) {
    PLONK_REQUIRE(ctx, PLONK_ERR_ARG, "ctx is NULL");
    PLONK_ENTER(ctx);
    PLONK_REQUIRE(window_bits == 0 || (window_bits >= 2 && window_bits <= 13), PLONK_ERR_ARG,
                  "window_bits must be 0 (default) or in [2, 13]");
    ctx->msm_window_bits = window_bits;
    ctx->msm_groups = groups;
    return PLONK_OK;
}

int plonk_srs_lookup_bits(const plonk_srs* srs, unsigned* out_bits) {
    PLONK_REQUIRE(srs && out_bits, PLONK_ERR_ARG, "bad argument");
    *out_bits = srs->lookup_bits;
    return PLONK_OK;
}

int plonk_srs_lookup_info(const plonk_srs* srs, unsigned* out_bits, size_t* out_bytes, double* out_build_s, int* out_sharers) {
    PLONK_REQUIRE(srs && out_bits && out_bytes && out_build_s && out_sharers, PLONK_ERR_ARG, "bad argument");
    return msm_lookup_info(srs, out_bits, out_bytes, out_build_s, out_sharers);
}

int plonk_srs_lookup_layout(const plonk_srs* srs, unsigned* out_kind, unsigned* out_additions_per_base) {
    PLONK_REQUIRE(srs && out_kind && out_additions_per_base, PLONK_ERR_ARG, "bad argument");
    return msm_lookup_layout(srs, out_kind, out_additions_per_base);
}

int plonk_srs_lookup_top(const plonk_srs* srs, unsigned* out_top_bits, unsigned* out_bases_per_group) {
    PLONK_REQUIRE(srs && out_top_bits && out_bases_per_group, PLONK_ERR_ARG, "bad argument");
    return msm_lookup_top(srs, out_top_bits, out_bases_per_group);
}

int plonk_msm_lookup_configure(plonk_ctx* ctx, int mode, unsigned bits, size_t budget_bytes) {
    PLONK_REQUIRE(ctx, PLONK_ERR_ARG, "ctx is NULL");
    PLONK_ENTER(ctx);
    const bool windows = (mode & 16) != 0, top = (mode & 32) != 0;
    mode &= ~(16 | 32);
    PLONK_REQUIRE(mode >= 0 && mode <= 2, PLONK_ERR_ARG,
                  "mode must be 0 (auto), 1 (off) or 2 (force), optionally + 16 (window tables) or + 32 (the comb of `bits` teeth with top tables)");
    PLONK_REQUIRE(!top || (!windows && bits && msm_comb_takes_top(bits)), PLONK_ERR_ARG,
                  "mode + 32 needs a comb (not + 16) of an explicit number of teeth with 254 mod teeth = 1 or 2 (7, 9, 11, 12, 14, 18, 21, 23 ..)");
    const unsigned max_bits = windows ? 17u : 24u;
    PLONK_REQUIRE(bits == 0 || (bits >= 2 && bits <= max_bits), PLONK_ERR_ARG, "bits must be 0 (auto) or in [2, %u]", max_bits);
    PLONK_REQUIRE(mode != 2 || bits, PLONK_ERR_ARG, "mode 2 needs an explicit number of bits");
    ctx->msm_lookup_mode = mode;
    ctx->msm_lookup_kind = windows ? MSM_TABLE_WINDOWS : MSM_TABLE_COMB;
    ctx->msm_lookup_bits = bits;
    ctx->msm_lookup_top = top;
    ctx->msm_lookup_budget = budget_bytes;
    return PLONK_OK;
}

int plonk_g1_msm(plonk_ctx* ctx, plonk_srs* srs, const void* d_scalars, size_t n, size_t batch,
                 size_t scalar_stride, uint8_t* h_out_xy_le, uint8_t* h_out_is_identity) {
    PLONK_REQUIRE(ctx && srs && d_scalars && h_out_xy_le && h_out_is_identity, PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    if (!batch) return PLONK_OK;
    void* res;
    PLONK_TRY(ctx_scratch(ctx, 2, batch * 64 + batch + 64, &res));
    Fq* d_xy = (Fq*)res;
    uint8_t* d_flags = (uint8_t*)res + batch * 64;
    PLONK_TRY(msm_run_device(ctx, srs, (const Fr*)d_scalars, n, batch, scalar_stride, d_xy, d_flags));
    PLONK_CHECK_HIP(hipMemcpyAsync(h_out_xy_le, d_xy, batch * 64, hipMemcpyDeviceToHost, ctx->stream));
    PLONK_CHECK_HIP(hipMemcpyAsync(h_out_is_identity, d_flags, batch, hipMemcpyDeviceToHost, ctx->stream));
    PLONK_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return PLONK_OK;
}

// ---- per-kernel profiling ------------------------------------------------------------------------
int plonk_profile_enable(plonk_ctx* ctx, int on) {
    PLONK_REQUIRE(ctx, PLONK_ERR_ARG, "ctx is NULL");
    PLONK_ENTER(ctx);
    ctx->profiling = on != 0;
    return PLONK_OK;
}

int plonk_profile_read(plonk_ctx* ctx, const char* kernel, double* total_ms, uint64_t* launches, double* algo_bytes) {
    PLONK_REQUIRE(ctx && kernel && total_ms && launches && algo_bytes, PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    PLONK_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    double ms = 0, bytes = 0;
    uint64_t n = 0;
    for (auto& r : ctx->prof) {
        const size_t kl = strlen(kernel);  // "name" matches exactly, "prefix*" every record whose name starts with prefix
        if (kl && kernel[kl - 1] == '*' ? strncmp(r.name, kernel, kl - 1) != 0 : strcmp(r.name, kernel) != 0) continue;
        float t = 0;
        PLONK_CHECK_HIP(hipEventElapsedTime(&t, r.a, r.b));
        ms += t;
        bytes += r.algo_bytes;
        n++;
    }
    *total_ms = ms;
    *launches = n;
    *algo_bytes = bytes;
    return PLONK_OK;
}

int plonk_profile_reset(plonk_ctx* ctx) {
    PLONK_REQUIRE(ctx, PLONK_ERR_ARG, "ctx is NULL");
    PLONK_ENTER(ctx);
    PLONK_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    for (auto& r : ctx->prof) {
        ctx->event_pool.push_back(r.a);
        ctx->event_pool.push_back(r.b);
    }
    ctx->prof.clear();
    return PLONK_OK;
}

// ---- timing ------------------------------------------------------------------------------------
int plonk_timer_start(plonk_ctx* ctx) {
    PLONK_REQUIRE(ctx, PLONK_ERR_ARG, "ctx is NULL");
    PLONK_ENTER(ctx);
    PLONK_CHECK_HIP(hipEventRecord(ctx->ev_a, ctx->stream));
    return PLONK_OK;
}
int plonk_timer_stop_ms(plonk_ctx* ctx, float* out_ms) {
    PLONK_REQUIRE(ctx && out_ms, PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    PLONK_CHECK_HIP(hipEventRecord(ctx->ev_b, ctx->stream));
    PLONK_CHECK_HIP(hipEventSynchronize(ctx->ev_b));
    PLONK_CHECK_HIP(hipEventElapsedTime(out_ms, ctx->ev_a, ctx->ev_b));
    return PLONK_OK;
}

}  // extern "C"
