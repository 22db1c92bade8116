/* A C caller of the C-ABI (include/plonk_hip.h) with no Python in between: the K1 known-answer test of the
 * reference (test.py:18-28: commit to the Lagrange vector 1..8) and an NTT round trip.  Built by
 * tests/test_c_abi.py against libplonk_hip.so (-m gpu) or the emulated build of the same sources (CPU suite).
 * usage: abi_demo <srs_2048.ptau>   -> prints "K1 <x decimal-free hex> <y hex>" and "roundtrip ok"           */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "plonk_hip.h"

#define CHECK(call)                                                                 \
    do {                                                                            \
        int rc_ = (call);                                                           \
        if (rc_ != PLONK_OK) {                                                      \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, plonk_last_error());      \
            return 1;                                                               \
        }                                                                           \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    static uint8_t ptau[80 + 64 * 2048];
    if (fread(ptau, 1, sizeof ptau, f) != sizeof ptau) return 2;
    fclose(f);
    const size_t powers = (size_t)1 << ptau[60]; /* setup.py:27 */
    if (powers != 2048) return 2;

    plonk_ctx* ctx;
    CHECK(plonk_ctx_create(0, &ctx));
    plonk_srs* srs;
    CHECK(plonk_srs_load_ptau(ctx, ptau + 80, powers, &srs)); /* setup.py:29-41 */

    uint8_t vals[8 * 32];
    memset(vals, 0, sizeof vals);
    for (int i = 0; i < 8; i++) vals[32 * i] = (uint8_t)(i + 1); /* canonical little-endian 1..8 */
    void *d_lag, *d_coef, *d_back;
    CHECK(plonk_mem_alloc(ctx, sizeof vals, &d_lag));
    CHECK(plonk_mem_alloc(ctx, sizeof vals, &d_coef));
    CHECK(plonk_mem_alloc(ctx, sizeof vals, &d_back));
    CHECK(plonk_fr_upload(ctx, d_lag, vals, 8));
    CHECK(plonk_fr_ntt(ctx, d_lag, d_coef, 3, 1, 1)); /* Setup.commit: values.ifft(), setup.py:68 */
    uint8_t xy[64], is_identity[1];
    CHECK(plonk_g1_msm(ctx, srs, d_coef, 8, 1, 8, xy, is_identity)); /* ec_lincomb, setup.py:69-72 */
    printf("K1 ");
    for (int c = 0; c < 2; c++) {
        for (int i = 31; i >= 0; i--) printf("%02x", xy[32 * c + i]);
        printf(c ? "\n" : " ");
    }
    CHECK(plonk_fr_ntt(ctx, d_coef, d_back, 3, 0, 1)); /* fft(ifft(x)) == x */
    uint8_t back[8 * 32];
    CHECK(plonk_fr_download(ctx, back, d_back, 8));
    if (memcmp(back, vals, sizeof vals) != 0 || is_identity[0]) {
        fprintf(stderr, "round trip mismatch\n");
        return 1;
    }
    printf("roundtrip ok\n");
    CHECK(plonk_mem_free(ctx, d_lag));
    CHECK(plonk_mem_free(ctx, d_coef));
    CHECK(plonk_mem_free(ctx, d_back));
    CHECK(plonk_srs_free(ctx, srs));
    CHECK(plonk_ctx_destroy(ctx));
    return 0;
}
