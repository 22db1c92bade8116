// TEST INFRASTRUCTURE ONLY — the emulation build has no RCCL: plonk_comm_* exist so the C-ABI table binds, and
// work for a single rank (a gather of one rank is a copy).  Multi-rank CPU tests use a socket / gloo transport
// above the C-ABI (tests/test_distributed_*.py); the RCCL calls themselves are exercised on the GPU box.
#include <string.h>

#include "../../include/plonk_hip.h"

struct plonk_comm { int rank, world; };
void plonk_set_error(const char* fmt, ...);

extern "C" {
int plonk_comm_unique_id(uint8_t out_id[PLONK_COMM_ID_BYTES]) { memset(out_id, 0x5a, PLONK_COMM_ID_BYTES); return PLONK_OK; }
int plonk_comm_create(plonk_ctx*, const uint8_t*, int rank, int world, plonk_comm** out) {
    if (world != 1 || rank != 0) { plonk_set_error("the emulation build has no RCCL: world must be 1"); return PLONK_ERR_STATE; }
    *out = new plonk_comm{0, 1};
    return PLONK_OK;
}
int plonk_comm_destroy(plonk_comm* c) { delete c; return PLONK_OK; }
int plonk_comm_size(const plonk_comm* c, int* r, int* w) { *r = c->rank; *w = c->world; return PLONK_OK; }
int plonk_gather_results(plonk_comm*, const uint8_t* s, size_t n, uint8_t* r) { memcpy(r, s, n); return PLONK_OK; }
int plonk_gather_proofs_device(plonk_comm*, plonk_prover* const* provers, size_t n_provers, size_t batch, int compressed, uint8_t* h_recv) {
    const size_t rec = compressed ? 480 : 768, n = n_provers * batch;  // one rank: the layout of the real call, filled by downloads
    memset(h_recv + n * rec, 0, (n + 15) & ~(size_t)15);
    for (size_t k = 0; k < n_provers; k++) {
        int rc = compressed ? plonk_prover_download_compressed(provers[k], batch, h_recv + k * batch * rec, h_recv + n * rec + k * batch)
                            : plonk_prover_download(provers[k], batch, h_recv + k * batch * rec, h_recv + n * rec + k * batch);
        if (rc != PLONK_OK) return rc;
    }
    return PLONK_OK;
}
int plonk_comm_set_default_timeout(double s) { return s >= 0 ? PLONK_OK : PLONK_ERR_ARG; }
int plonk_comm_set_timeout(plonk_comm*, double s) { return s >= 0 ? PLONK_OK : PLONK_ERR_ARG; }
int plonk_device_peer_access(int device, int* row, size_t cap) {
    if (device != 0 || !row || !cap) return PLONK_ERR_ARG;  // the emulation reports one device
    row[0] = 1;
    return PLONK_OK;
}
int plonk_comm_max_f64(plonk_comm*, double*) { return PLONK_OK; }
int plonk_comm_barrier(plonk_comm*) { return PLONK_OK; }
int plonk_comm_last_gather_ms(plonk_comm*, float* a, float* b) { *a = *b = 0; return PLONK_OK; }
int plonk_comm_info(const plonk_comm*, char* path, size_t cap, int* version, uint64_t* n) {
    if (path && cap) { strncpy(path, "(emulation: no RCCL)", cap - 1); path[cap - 1] = 0; }
    if (version) *version = 0;
    if (n) *n = 0;
    return PLONK_OK;
}
int plonk_comm_all_to_all(plonk_comm*, const void* s, void* r, size_t n) { memcpy(r, s, n); return PLONK_OK; }
int plonk_fr_ntt_distributed(plonk_comm*, const void*, void*, unsigned, int) {
    plonk_set_error("the emulation build has no RCCL: run plonk_fr_ntt_dist_columns / _rows around another transport");
    return PLONK_ERR_STATE;
}
}
