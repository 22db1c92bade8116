// ntt_wave_host.h — host side of the wave NTT kernels (ntt_wave.h), generic over the field: tables, launches and the
// one- / two-pass driver.  A field is a traits type F with
//   typedef ... P;                                            the Montgomery parameters (fp.h)
//   static Fp<P> root_of_unity(unsigned log_n, bool inverse)  w = g^((m-1)/2^log_n), Montgomery form (host)
//   static Fp<P> from_u64(uint64_t)                           Montgomery form of a small integer (host)
//   static int packed_roots(ctx, log_n, inverse, const Fp<P>** out)            w^k, k < 2^log_n, on the device
//   static int packed_lo_hi(ctx, log_n, inverse, const Fp<P>** lo, const Fp<P>** hi)   w^e = lo[e & 1023] * hi[e >> 10]
//   static int powers(ctx, base, first, Fp<P>* out, n)        out[i] = first * base^i on the device
//   static WaveTables& tables(plonk_ctx*)                     this field's table cache of the context
// ntt.hip instantiates it for BN254 Fr (the prover's field), ntt_bls.hip for BLS12-381 Fr (standalone transform).
#pragma once
#include <stdlib.h>
#include <string.h>

#include "ntt_wave.h"

// the two-pass plan of a size (ntt.hip; field independent)
bool ntt_wave_plan(const plonk_ctx* ctx, unsigned log_n, bool latency, unsigned* log_r1, unsigned* log_r2);

template <class P> static int wave_limb_table(plonk_ctx* ctx, std::map<unsigned, int32_t*>& cache, unsigned key, const Fp<P>* packed, size_t n,
                                              const int32_t** out) {
    auto it = cache.find(key);
    if (it == cache.end()) {
        void* d = nullptr;
        Ninv261 ninv;
        fpl_ninv261<P>(ninv.l);
        if (!plonk_dev_malloc(&d, n * NTT_SHOUP_STRIDE * sizeof(int32_t))) {
            plonk_set_error("hipMalloc of a %zu-entry twiddle table failed", n);
            return PLONK_ERR_NOMEM;
        }
        ctx->owned.push_back(d);
        PLONK_LAUNCH(ntt_limb_table_kernel<P>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, packed, (int32_t*)d, n, ninv);
        PLONK_CHECK_HIP(hipGetLastError());
        it = cache.emplace(key, (int32_t*)d).first;
    }
    *out = it->second;
    return PLONK_OK;
}

// the twiddles of the wave kernel serving 2^log_n, in program order (built once per size and direction)
template <class F> static int wave_program_table(plonk_ctx* ctx, unsigned log_n, unsigned log_e, bool inverse, const int32_t** out) {
    typedef typename F::P P;
    WaveTables& T = F::tables(ctx);
    const unsigned key = log_n | (inverse ? 256u : 0u) | (log_e << 12);  // (2^9 exists with eight and with two elements per thread)
    auto it = T.prog.find(key);
    if (it == T.prog.end()) {
        const Fp<P>* packed;
        PLONK_TRY(F::packed_roots(ctx, log_n, inverse, &packed));
        const unsigned nlds = wavel_nlds(log_n, log_e), stages = wavel_tw_stages(log_e, nlds);
        const size_t entries = wavel_tw_offset(log_e, nlds, stages);
        void* d = nullptr;
        if (!plonk_dev_malloc(&d, entries * NTT_SHOUP_STRIDE * sizeof(int32_t))) {
            plonk_set_error("hipMalloc of a %zu-entry twiddle table failed", entries);
            return PLONK_ERR_NOMEM;
        }
        ctx->owned.push_back(d);
        Ninv261 ninv;
        fpl_ninv261<P>(ninv.l);
        for (unsigned st = 0; st < stages; st++) {
            const unsigned nb = wavel_tw_nb(log_e, nlds, st), count = wavel_tw_count(log_e, nlds, st), mult = wavel_tw_mult(log_e, nlds, st);
            for (unsigned f = 1; f <= count; f++) {
                int32_t* block = (int32_t*)d + ((size_t)wavel_tw_offset(log_e, nlds, st) + (size_t)(f - 1) * nb) * NTT_SHOUP_STRIDE;
                PLONK_LAUNCH(ntt_program_block_kernel<P>, dim3((nb + 63) / 64), dim3(64), 0, ctx->stream, packed, log_n, nb, f, mult, block, ninv);
            }
        }
        PLONK_CHECK_HIP(hipGetLastError());
        it = T.prog.emplace(key, (int32_t*)d).first;
    }
    *out = it->second;
    return PLONK_OK;
}

// inter-pass twiddle tables as Shoup pairs; scaled: the hi table times 1/N (the inverse transform's factor, folded in)
template <class F> static int wave_lo_hi(plonk_ctx* ctx, unsigned log_n, bool inverse, bool scaled, const int32_t** lo, const int32_t** hi) {
    typedef typename F::P P;
    WaveTables& T = F::tables(ctx);
    const Fp<P>*plo, *phi;
    PLONK_TRY(F::packed_lo_hi(ctx, log_n, inverse, &plo, &phi));
    const unsigned key = log_n | (inverse ? 256u : 0u);
    const unsigned log_lo = log_n < NTT_TW_LO_LOG ? log_n : NTT_TW_LO_LOG;
    const size_t nhi = log_n > NTT_TW_LO_LOG ? ((size_t)1 << (log_n - NTT_TW_LO_LOG)) : 1;
    PLONK_TRY(wave_limb_table<P>(ctx, T.lo, key, plo, (size_t)1 << log_lo, lo));
    if (!scaled) return wave_limb_table<P>(ctx, T.hi, key, phi, nhi, hi);
    if (T.hi.find(key | 512u) == T.hi.end()) {  // (1/N) * w_hi^k, built once
        Fp<P> whi = F::root_of_unity(log_n, inverse);
        for (unsigned i = 0; i < NTT_TW_LO_LOG; i++) whi = fp_sqr(whi);
        void* tmp = nullptr;  // (not a scratch slot: callers hold those across this call)
        if (!plonk_dev_malloc(&tmp, nhi * sizeof(Fp<P>))) {
            plonk_set_error("hipMalloc of a %zu-entry twiddle table failed", nhi);
            return PLONK_ERR_NOMEM;
        }
        int rc = F::powers(ctx, whi, fp_inv(F::from_u64((uint64_t)1 << log_n)), (Fp<P>*)tmp, nhi);
        if (rc == PLONK_OK) rc = wave_limb_table<P>(ctx, T.hi, key | 512u, (const Fp<P>*)tmp, nhi, hi);
        hipStreamSynchronize(ctx->stream);  // the packed copy must outlive the conversion kernel only
        hipFree(tmp);
        return rc;
    }
    return wave_limb_table<P>(ctx, T.hi, key | 512u, (const Fp<P>*)nullptr, nhi, hi);
}

// the column pass's inter-pass twiddles as one table in usage order (ntt_interpass_table_kernel), when the context's
// table budget allows its N * 80 bytes; *out = null otherwise (the caller falls back to the two small tables)
template <class F>
static int wave_interpass_table(plonk_ctx* ctx, unsigned log_n, unsigned log_r1, unsigned log_e1, bool inverse, bool scaled, const int32_t** out) {
    typedef typename F::P P;
    WaveTables& T = F::tables(ctx);
    *out = nullptr;
    if (!ctx->ntt_table_budget || log_e1 == 3) return PLONK_OK;  // switched off (tables built earlier stay allocated, unused); not for the E = 8 column kernels
    const unsigned key = log_n | (inverse ? 256u : 0u) | (scaled ? 512u : 0u) | (log_r1 << 12) | (log_e1 << 20);
    auto it = T.interpass.find(key);
    if (it == T.interpass.end()) {
        const size_t bytes = ((size_t)NTT_SHOUP_STRIDE * sizeof(int32_t)) << log_n;
        if (ctx->ntt_tables_bytes + bytes > ctx->ntt_table_budget) return PLONK_OK;
        const Fp<P>*plo, *phi;
        PLONK_TRY(F::packed_lo_hi(ctx, log_n, inverse, &plo, &phi));
        void* d = nullptr;
        if (!plonk_dev_malloc(&d, bytes)) {
            (void)hipGetLastError();  // not an error of the transform: the two small tables serve it
            return PLONK_OK;
        }
        ctx->owned.push_back(d);
        ctx->ntt_tables_bytes += bytes;
        Ninv261 ninv;
        fpl_ninv261<P>(ninv.l);
        const Fp<P> scale = scaled ? fp_inv(F::from_u64((uint64_t)1 << log_n)) : fp_one<P>();
        PLONK_LAUNCH(ntt_interpass_table_kernel<P>, dim3((unsigned)((((size_t)1 << log_n) + 255) / 256)), dim3(256), 0, ctx->stream, plo, phi, scale, log_n,
                     log_r1, log_e1, (int32_t*)d, ninv);
        PLONK_CHECK_HIP(hipGetLastError());
        it = T.interpass.emplace(key, (int32_t*)d).first;
    }
    *out = it->second;
    return PLONK_OK;
}

// fpl_reduce_small's table of j * m: 2 FPL_RS_J + 1 entries of 12 words, built on the host once per context and field
template <class F> static int wave_jm(plonk_ctx* ctx, const int32_t** out) {
    WaveTables& T = F::tables(ctx);
    if (!T.jm) {
        int32_t host[(2 * FPL_RS_J + 1) * 12];
        for (int j = -FPL_RS_J; j <= FPL_RS_J; j++) fpl_jm_entry<typename F::P>(j, host + (j + FPL_RS_J) * 12);
        void* d = nullptr;
        if (!plonk_dev_malloc(&d, sizeof host)) {
            plonk_set_error("hipMalloc of the NTT range-reduction table failed");
            return PLONK_ERR_NOMEM;
        }
        ctx->owned.push_back(d);
        PLONK_CHECK_HIP(hipMemcpy(d, host, sizeof host, hipMemcpyHostToDevice));
        T.jm = (const int32_t*)d;
    }
    *out = T.jm;
    return PLONK_OK;
}

// the radix-8 / radix-4 roots w_8^k of a transform direction as Shoup pairs (kernel arguments), and the rest zeroed
template <class F> static void wave_params_init(NttWaveT<typename F::P>* p, unsigned log_n, bool inverse) {
    typedef typename F::P P;
    memset(p, 0, sizeof *p);
    p->log_n = log_n;
    uint32_t ninv[9];
    fpl_ninv261<P>(ninv);
    const Fp<P> w8 = F::root_of_unity(3, inverse), w4 = fp_sqr(w8);
    p->w8[0] = fpl_shoup_from_mont(w8, ninv);
    p->w8[1] = fpl_shoup_from_mont(w4, ninv);
    p->w8[2] = fpl_shoup_from_mont(fp_mul(w4, w8), ninv);
}

template <class F, unsigned LOG_E, unsigned NLDS> static int wave_launch_as(plonk_ctx* ctx, const NttWaveT<typename F::P>& q, unsigned grid_x, unsigned grid_y) {
    typedef typename F::P P;
    constexpr unsigned nt = 1u << wavel_log_t(NLDS);
    const size_t shmem = NLDS ? (size_t)(LOG_E >= 2 ? 4 : 1) * nt * 36 : 0;  // one round of the wave-bit exchange: 4 elements (E = 2: one) of 9 words per thread
    WaveTables& T = F::tables(ctx);
    if (NLDS == 2 && LOG_E >= 2 && !T.attr_set[LOG_E >= 2 ? LOG_E - 2 : 0]) {  // 144 KiB: above the default limit; a per-device attribute, tracked per context
        PLONK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ntt_wavel_kernel<P, LOG_E, NLDS>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(144 * 1024)));
        T.attr_set[LOG_E - 2] = true;
    }
    PLONK_REQUIRE(NLDS != 3 || q.mode != 1, PLONK_ERR_STATE, "the 512-thread 2^12 kernel cannot run a column pass");
    if (NLDS == 3 && !T.attr_set[3]) {  // the 512-thread 2^12 kernel: 72 KiB
        PLONK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ntt_wavel_kernel<P, LOG_E, NLDS>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(72 * 1024)));
        T.attr_set[3] = true;
    }
    void (*kern)(NttWaveT<P>) = ntt_wavel_kernel<P, LOG_E, NLDS>;  // (a template-id's comma would split the macro's arguments)
    if constexpr (LOG_E <= 2) {
        if (q.tw_always == 2u) {  // column pass on the full inter-pass table
            kern = ntt_wavel_column_kernel<P, LOG_E, NLDS>;
            if (NLDS == 2 && !T.attr_set[2]) {
                PLONK_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(144 * 1024)));
                T.attr_set[2] = true;
            }
        }
    }
    PLONK_LAUNCH(kern, dim3(grid_x, grid_y), dim3(nt), shmem, ctx->stream, q);
    return PLONK_OK;
}

// log_r = 8, 10, 12: 4 elements per thread; 9, 11, 13 (and 12 in its 512-thread form): 8; 7, and 9 in its latency form: 2
template <class F> static int wave_launch(plonk_ctx* ctx, const NttWaveT<typename F::P>& p, unsigned log_r, unsigned log_e, unsigned grid_x, unsigned grid_y) {
    NttWaveT<typename F::P> q = p;
    if (!q.jm) PLONK_TRY(wave_jm<F>(ctx, &q.jm));  // (wave_run's cached plans carry it)
    switch (log_r | (log_e << 8)) {
        case 8 | (2 << 8): return wave_launch_as<F, 2, 0>(ctx, q, grid_x, grid_y);
        case 10 | (2 << 8): return wave_launch_as<F, 2, 1>(ctx, q, grid_x, grid_y);
        case 12 | (2 << 8): return wave_launch_as<F, 2, 2>(ctx, q, grid_x, grid_y);
        case 9 | (3 << 8): return wave_launch_as<F, 3, 0>(ctx, q, grid_x, grid_y);
        case 11 | (3 << 8): return wave_launch_as<F, 3, 1>(ctx, q, grid_x, grid_y);
        case 13 | (3 << 8): return wave_launch_as<F, 3, 2>(ctx, q, grid_x, grid_y);
        case 12 | (3 << 8): return wave_launch_as<F, 3, 3>(ctx, q, grid_x, grid_y);  // 512 threads x 8 elements
        case 7 | (1 << 8): return wave_launch_as<F, 1, 0>(ctx, q, grid_x, grid_y);
        case 9 | (1 << 8): return wave_launch_as<F, 1, 1>(ctx, q, grid_x, grid_y);
    }
    plonk_set_error("no wave kernel for a 2^%u-point transform with 2^%u elements per thread", log_r, log_e);
    return PLONK_ERR_ARG;
}

// Everything of a transform's launches that does not depend on the call's buffers: the split, the kernel arguments with their
// table pointers and constants (the radix-8 roots as Shoup pairs, 1/N, the range-reduction table).  Built on the first
// call of a (size, direction, scaling, table choice) and cached per context and field: the host-side field arithmetic —
// a root of unity by repeated squaring, a 261-bit Hensel lift, a modular inversion, ~20 us in all — used to run on EVERY
// call, and a lone 2^16 transform is ~10 us of device time (0.031 ms measured).
// PLONK_NTT_W12 = 4: keep 2^12 on the 1024-thread 4-element kernel everywhere (A/B runs)
static inline bool wave_w12_as_512x8() {
    static const bool on = [] {
        const char* e = getenv("PLONK_NTT_W12");
        return !(e && atoi(e) == 4);
    }();
    return on;
}

static inline bool wave_tmp_transposed() {
    static const bool on = [] {
        const char* e = getenv("PLONK_NTT_TMP_TRANSPOSED");
        return !(e && atoi(e) == 0);
    }();
    return on;
}

template <class P> struct WavePlan {
    unsigned log_r1 = 0, log_r2 = 0;
    unsigned log_e1 = 0, log_e2 = 0;  // elements per thread (log2) of the kernels of the two passes (log_e1 alone for a single pass)
    NttWaveT<P> a, c;  // single pass: a;  two passes: a = columns, c = rows
};

// big: the call has at least 2^20 elements (the 512-thread form of 2^12 pays from there: profiles/r04_f_ntt_2e12_forms_ab.jsonl)
template <class F>
static int wave_plan_get(plonk_ctx* ctx, unsigned log_n, bool inverse, bool scale_by_n_inv, bool want_full, bool latency, bool big,
                         const WavePlan<typename F::P>** out) {
    typedef typename F::P P;
    typedef Fp<P> E;
    WaveTables& T = F::tables(ctx);
    if (T.plan_epoch != ctx->ntt_cfg_epoch) {  // a split, budget or kernel choice changed: plans are rebuilt
        T.plans.clear();
        T.plan_epoch = ctx->ntt_cfg_epoch;
    }
    const unsigned key = log_n | (inverse ? 256u : 0u) | (scale_by_n_inv ? 512u : 0u) | (want_full ? 1024u : 0u) | (latency ? 2048u : 0u) | (big ? 4096u : 0u);
    auto it = T.plans.find(key);
    if (it != T.plans.end()) {
        *out = static_cast<const WavePlan<P>*>(it->second.get());
        return PLONK_OK;
    }
    std::shared_ptr<WavePlan<P>> plan(new WavePlan<P>());
    const size_t N = (size_t)1 << log_n;
    unsigned log_r1 = 0, log_r2 = 0;
    PLONK_REQUIRE(ntt_wave_plan(ctx, log_n, latency, &log_r1, &log_r2), PLONK_ERR_ARG, "no wave-kernel plan for 2^%u points", log_n);
    plan->log_r1 = log_r1;
    plan->log_r2 = log_r2;
    // 2^12 runs as 512 threads x 8 elements (two workgroups per CU) as a single pass and as a row pass of calls with at least 2^20
    // elements: 18.0 -> 19.2 G elements/s at 512 transforms, 17.7 -> 20.7 at 4096, row passes +2 .. 5 %; a small call is faster on
    // the 1024-thread kernel's shorter chain (a lone 2^12: 0.043 against 0.048 ms).  A column pass stays on the 1024-thread
    // 4-element kernel (the one-table inter-pass twiddles need its register room; ntt_wave.h)
    const bool w12 = big && wave_w12_as_512x8() && ctx->ntt_kind != 8;  // (kind 8: 2^12 on the 1024-thread form everywhere — tests, A/B)
    const unsigned log_e1 = plan->log_e1 = (log_r1 == 12 && w12 && !log_r2) ? 3u : wavel_log_e(log_r1, latency);
    const unsigned log_e2 = plan->log_e2 = !log_r2 ? 0u : ((log_r2 == 12 && w12) ? 3u : wavel_log_e(log_r2, latency));
    NttWaveT<P> p;
    wave_params_init<F>(&p, log_n, inverse);
    PLONK_TRY(wave_jm<F>(ctx, &p.jm));
    E n_inv = fp_zero<P>();
    if (scale_by_n_inv) n_inv = fp_inv(F::from_u64((uint64_t)N));
    if (!log_r2) {
        PLONK_TRY(wave_program_table<F>(ctx, log_n, log_e1, inverse, &p.roots));
        p.out_scalar = n_inv;
        p.has_out_scalar = scale_by_n_inv;
        plan->a = p;
    } else {
        // (measured, profiles/r03_m_ntt_sweep.jsonl: the table wins 4-10 % wherever the column pass fills the chip; a lone 2^18 —
        // one workgroup per CU, every load latency exposed — is 5 % faster on the small, L2-resident tables)
        const int32_t* full = nullptr;
        if (want_full) PLONK_TRY(wave_interpass_table<F>(ctx, log_n, log_r1, log_e1, inverse, scale_by_n_inv, &full));
        if (full) {
            p.tw_lo = full;
            p.tw_always = 2u;
        } else {
            PLONK_TRY(wave_lo_hi<F>(ctx, log_n, inverse, scale_by_n_inv, &p.tw_lo, &p.tw_hi));
            p.tw_always = scale_by_n_inv ? 1u : 0u;
        }
        NttWaveT<P> a = p;
        // the intermediate buffer is this library's own scratch: kept transposed (PLONK_NTT_TMP_TRANSPOSED=0: as [k1][c], A/B runs)
        // Measured (profiles/r04_g_ntt_tmp_transposed_ab.jsonl): lone transforms level, batches 3 - 9 % faster, the column pass's
        // WRITE_SIZE 1.28 - 1.38 x -> 1.02 x its bytes at 2^22 / 2^24; 2^25 = 2^12 x 2^13 is 5 % slower (its 1024-thread row kernel
        // gathers eight elements per thread 128 KiB apart) and keeps the [k1][c] form, like 2^26 (the 2^13 kernel cannot store
        // transposed: ntt_wave.h)
        const bool tr = wave_tmp_transposed() && log_r1 != 13 && log_n <= 24;
        a.mode = 1;
        a.fan = tr ? NTT_TMP_TRANSPOSED : 0u;
        a.log_other = log_r2;
        a.out_bstride = N;
        PLONK_TRY(wave_program_table<F>(ctx, log_r1, log_e1, inverse, &a.roots));
        NttWaveT<P> c = p;
        c.mode = 2;
        c.log_other = log_r1;
        c.chunk_stride = tr ? 1u << log_r1 : 0u;  // (chunk_log = 0: position c of row k1 at c R1 + k1)
        c.in_bstride = N;
        c.in_len = (unsigned)N;
        c.has_out_scalar = 0;  // 1/N went into the column pass's inter-pass twiddles (tw_hi)
        c.tw_always = 0;
        PLONK_TRY(wave_program_table<F>(ctx, log_r2, log_e2, inverse, &c.roots));
        plan->a = a;
        plan->c = c;
    }
    PLONK_CHECK_HIP(hipGetLastError());
    *out = plan.get();
    T.plans.emplace(key, std::static_pointer_cast<void>(plan));
    return PLONK_OK;
}

// the wave kernels take a fan whose strides are 0 or the transform size (ntt_wave.h: NttWaveT::fan)
// (and the 2^13 kernel takes no per-copy out_scale: ntt_wave.h, FAN_OUT_SCALE)
static inline bool wave_fan_ok(const NttFan& f, size_t N, bool has_out_scale) {
    return (f.in_stride == 0 || f.in_stride == N) && (f.out_stride == 0 || f.out_stride == N) && (f.scale_stride == 0 || f.scale_stride == N) &&
           !(N == ((size_t)1 << 13) && has_out_scale && f.scale_stride);
}

// one transform per batch entry: a single launch for 2^8 .. 2^13, columns then rows through scratch slot 0 for 2^14 .. 2^26
template <class F>
static int wave_run(plonk_ctx* ctx, const Fp<typename F::P>* in, Fp<typename F::P>* out, unsigned log_n, bool inverse, size_t batch, size_t in_len,
                    size_t in_bstride, size_t out_bstride, const Fp<typename F::P>* in_scale, const Fp<typename F::P>* out_scale, bool scale_by_n_inv,
                    const NttFan* fan = nullptr) {
    typedef typename F::P P;
    typedef Fp<P> E;
    const size_t N = (size_t)1 << log_n;
    PLONK_REQUIRE(!fan || (log_n <= 13 && wave_fan_ok(*fan, N, out_scale != nullptr)), PLONK_ERR_ARG,
                  "a fanned transform needs a single-pass size (2^7 .. 2^13) and strides of 0 or N");
    const bool want_full = log_n >= 14 && (log_n <= 16 || ((size_t)batch << log_n) >= ((size_t)1 << 19));
    // a small job waits for one wave's instruction chain: the two-element forms (plonk_ntt_select_kernel 6 / 7: never / always — tests, A/B runs)
    const bool latency = ctx->ntt_kind == 6 ? false : (ctx->ntt_kind == 7 || ((size_t)batch << log_n) <= ((size_t)1 << 18));
    const WavePlan<P>* plan;
    const bool big = ctx->ntt_kind == 6 || ((size_t)batch << log_n) >= ((size_t)1 << 20);
    PLONK_TRY(wave_plan_get<F>(ctx, log_n, inverse, scale_by_n_inv, want_full, latency, big, &plan));
    const unsigned log_r1 = plan->log_r1, log_r2 = plan->log_r2;
    const unsigned in_len32 = (unsigned)(in_len < N ? in_len : N);
    if (!log_r2) {
        NttWaveT<P> p = plan->a;
        p.in_bstride = in_bstride;
        p.out_bstride = out_bstride;
        p.in_len = in_len32;
        p.in_scale = in_scale;
        p.out_scale = out_scale;
        if (fan) p.fan = (fan->in_stride ? NTT_FAN_IN : 0u) | (fan->out_stride ? NTT_FAN_OUT : 0u) | (fan->scale_stride ? NTT_FAN_SCALE : 0u);
        // an in-place transform is safe: every thread has read all of its inputs before any thread stores (the stages
        // in between are separated by barriers for L > 0; for L = 0 the single wave runs in lock step)
        PLONK_TRY(prof_begin(ctx, "ntt_pass", 64.0 * (double)N * (double)batch));
        for (size_t b0 = 0; b0 < batch; b0 += (size_t)1 << 30) {  // grid.x carries the batch
            const size_t nb = batch - b0 < ((size_t)1 << 30) ? batch - b0 : (size_t)1 << 30;
            p.in = in + b0 * in_bstride;
            p.out = out + b0 * out_bstride;
            PLONK_TRY(wave_launch<F>(ctx, p, log_n, plan->log_e1, (unsigned)nb, fan ? fan->count : 1));
        }
        PLONK_TRY(prof_end(ctx));
        PLONK_CHECK_HIP(hipGetLastError());  // a refused launch (thread-local, no synchronisation)
        return PLONK_OK;
    }
    // two passes through a scratch copy: columns (R1 points each, stride R2), then rows (R2 points each)
    PLONK_REQUIRE(batch <= 65535, PLONK_ERR_ARG, "NTT batch %zu exceeds 65535", batch);
    void* sc;
    PLONK_TRY(ctx_scratch(ctx, 0, batch * N * sizeof(E), &sc));
    E* tmp = (E*)sc;
    NttWaveT<P> c = plan->c;
    c.in = tmp;
    c.out = out;
    c.out_bstride = out_bstride;
    c.out_scale = out_scale;
    NttWaveT<P> a = plan->a;
    a.in = in;
    a.out = tmp;
    a.in_bstride = in_bstride;
    a.in_len = in_len32;
    a.in_scale = in_scale;
    PLONK_TRY(prof_begin(ctx, "ntt_pass_columns", 32.0 * (double)N * (double)batch));
    PLONK_TRY(wave_launch<F>(ctx, a, log_r1, plan->log_e1, 1u << log_r2, (unsigned)batch));
    PLONK_TRY(prof_end(ctx));
    PLONK_TRY(prof_begin(ctx, "ntt_pass_rows", 32.0 * (double)N * (double)batch));
    PLONK_TRY(wave_launch<F>(ctx, c, log_r2, plan->log_e2, 1u << log_r1, (unsigned)batch));
    PLONK_TRY(prof_end(ctx));
    PLONK_CHECK_HIP(hipGetLastError());
    return PLONK_OK;
}
