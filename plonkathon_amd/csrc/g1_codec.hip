// g1_codec.hip — plonk_g1_compress / plonk_g1_decompress: the compressed G1 encoding of g1_codec.h, batched on the device.
// Decompression solves y^2 = x^3 + 3 over BN254 Fq: p = 3 (mod 4), so y = (x^3 + 3)^((p + 1) / 4) when a root exists.
#include "plonk_internal.h"
#include "g1_codec.h"

struct FqExponent { uint32_t e[8]; };

// status: bit 0 = a coordinate is not below p
__global__ void g1_compress_kernel(const uint32_t* xy, size_t n, uint8_t* out, uint32_t* bad) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x[8], y[8];
    for (int k = 0; k < 8; k++) {
        x[k] = xy[16 * i + k];
        y[k] = xy[16 * i + 8 + k];
    }
    bool ok = true;
    for (int h = 0; h < 2; h++) {
        const uint32_t* v = h ? y : x;
        bool lt = false, eq = true;
        for (int k = 7; k >= 0 && eq; k--)
            if (v[k] != FqParams::mod(k)) {
                lt = v[k] < FqParams::mod(k);
                eq = false;
            }
        ok = ok && lt;
    }
    if (!ok) atomicOr(bad, 1u);
    uint8_t o[32];
    g1c_compress<FqParams>(x, y, o);
    for (int k = 0; k < 32; k++) out[32 * i + k] = o[k];
}

// status[i]: 0 ok, 1 malformed (flag bits 00, x >= p, or infinity with x != 0), 2 = x^3 + 3 is not a square (not on the curve)
__global__ void g1_decompress_kernel(const uint8_t* in, size_t n, FqExponent sqrt_exp, uint32_t* xy, uint8_t* status) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t* b = in + 32 * i;
    const unsigned flag = b[0] & 0xC0u;
    Fq x;
    for (int k = 0; k < 8; k++) {
        const uint8_t* w = b + 4 * (7 - k);
        x.v[k] = ((uint32_t)(k == 7 ? (w[0] & 0x3Fu) : w[0]) << 24) | ((uint32_t)w[1] << 16) | ((uint32_t)w[2] << 8) | (uint32_t)w[3];
    }
    uint32_t* o = xy + 16 * i;
    for (int k = 0; k < 16; k++) o[k] = 0;
    bool lt = false, eq = true;
    for (int k = 7; k >= 0 && eq; k--)
        if (x.v[k] != FqParams::mod(k)) {
            lt = x.v[k] < FqParams::mod(k);
            eq = false;
        }
    if (flag == 0 || !lt) {
        status[i] = 1;
        return;
    }
    if (flag == PLONK_G1C_INFINITY) {
        status[i] = fp_is_zero(x) ? 0 : 1;
        return;
    }
    const Fq xm = fp_to_mont(x);
    Fq three = fp_zero<FqParams>();
    three.v[0] = 3;
    const Fq rhs = fp_add(fp_mul(fp_sqr(xm), xm), fp_to_mont(three));
    Fq y = fp_pow_limbs(rhs, sqrt_exp.e);
    if (!fp_eq(fp_sqr(y), rhs)) {
        status[i] = 2;
        return;
    }
    Fq yc = fp_from_mont(y);
    if (g1c_is_larger_half<FqParams>(yc.v) != (flag == PLONK_G1C_LARGEST)) yc = fp_from_mont(fp_neg(y));
    for (int k = 0; k < 8; k++) {
        o[k] = x.v[k];
        o[8 + k] = yc.v[k];
    }
    status[i] = 0;
}

extern "C" {

int plonk_g1_compress(plonk_ctx* ctx, const uint8_t* h_xy_le, size_t count, uint8_t* h_out32) {
    PLONK_REQUIRE(ctx && (count == 0 || (h_xy_le && h_out32)), PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    if (!count) return PLONK_OK;
    void* buf;
    PLONK_TRY(ctx_scratch(ctx, 3, count * 96 + 64, &buf));
    uint8_t* d_in = (uint8_t*)buf;
    uint8_t* d_out = d_in + count * 64;
    uint32_t* d_bad = reinterpret_cast<uint32_t*>(d_in + ((count * 96 + 15) / 16) * 16);
    PLONK_CHECK_HIP(hipMemcpyAsync(d_in, h_xy_le, count * 64, hipMemcpyHostToDevice, ctx->stream));
    PLONK_CHECK_HIP(hipMemsetAsync(d_bad, 0, 4, ctx->stream));
    PLONK_LAUNCH(g1_compress_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, ctx->stream, (const uint32_t*)d_in, count, d_out, d_bad);
    PLONK_CHECK_HIP(hipGetLastError());
    uint32_t bad = 0;
    PLONK_CHECK_HIP(hipMemcpyAsync(h_out32, d_out, count * 32, hipMemcpyDeviceToHost, ctx->stream));
    PLONK_CHECK_HIP(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    PLONK_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    PLONK_REQUIRE(!bad, PLONK_ERR_ARG, "a coordinate is not a canonical Fq value (>= p)");
    return PLONK_OK;
}

int plonk_g1_decompress(plonk_ctx* ctx, const uint8_t* h_in32, size_t count, uint8_t* h_out_xy_le, uint8_t* h_status) {
    PLONK_REQUIRE(ctx && (count == 0 || (h_in32 && h_out_xy_le && h_status)), PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    if (!count) return PLONK_OK;
    void* buf;
    PLONK_TRY(ctx_scratch(ctx, 3, count * 97 + 64, &buf));
    uint8_t* d_xy = (uint8_t*)buf;  // 64-byte records first: 4-byte aligned words
    uint8_t* d_in = d_xy + count * 64;
    uint8_t* d_st = d_in + count * 32;
    FqExponent ex;  // (p + 1) / 4
    uint64_t carry = 1;
    uint32_t t[8];
    for (int i = 0; i < 8; i++) {
        const uint64_t v = (uint64_t)FqParams::mod(i) + carry;
        t[i] = (uint32_t)v;
        carry = v >> 32;
    }
    for (int i = 0; i < 8; i++) ex.e[i] = (t[i] >> 2) | (i < 7 ? t[i + 1] << 30 : 0);
    PLONK_CHECK_HIP(hipMemcpyAsync(d_in, h_in32, count * 32, hipMemcpyHostToDevice, ctx->stream));
    PLONK_LAUNCH(g1_decompress_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, ctx->stream, (const uint8_t*)d_in, count, ex,
                 (uint32_t*)d_xy, d_st);
    PLONK_CHECK_HIP(hipGetLastError());
    PLONK_CHECK_HIP(hipMemcpyAsync(h_out_xy_le, d_xy, count * 64, hipMemcpyDeviceToHost, ctx->stream));
    PLONK_CHECK_HIP(hipMemcpyAsync(h_status, d_st, count, hipMemcpyDeviceToHost, ctx->stream));
    PLONK_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    return PLONK_OK;
}

}  // extern "C"
