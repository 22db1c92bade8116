// comm.hip — everything that crosses xGMI, RCCL behind the C-ABI:
//   * the one collective of the prover path: gathering the finished proofs (768 bytes each: 9 affine G1 + 6 Fr) of
//     independently proving GPUs (SURVEY.md §8(b) `plonk_gather_results`, §8(e)), plus a barrier and a max-reduction
//     for the benchmark clock;
//   * the transpose step of a transform split across GPUs (`plonk_comm_all_to_all`, `plonk_fr_ntt_distributed`,
//     SURVEY.md §8(f) N4): grouped point-to-point ncclSend / ncclRecv, since xGMI is point-to-point anyway.
// The reference is a single Python process and has no counterpart.  One process per GPU; rank 0
// draws the ncclUniqueId (plonk_comm_unique_id) and hands its 128 bytes to the other ranks out of band
// (plonkathon_amd/distributed.py does it over a loopback socket).  librccl is dlopen'ed on first use so that
// single-GPU users never load it.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>

#include <rccl/rccl.h>  // types and enums only: every call goes through the dlsym table below

#include "plonk_internal.h"

namespace {
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                          // optional (NCCL >= 2.4): the deadline's way out
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;   // optional
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string asked;  // the name dlopen accepted
};
Rccl g_rccl;

// Load order, explicit so that the process decides which RCCL it talks to and not whatever an earlier import left in the
// link map: $PLONK_RCCL_LIB (a path), then the ROCm installation's own library ($ROCM_PATH/lib, /opt/rocm/lib), then the
// bare sonames.  (A process that imported PyTorch first has torch's bundled librccl mapped; with a bare soname dlopen
// would hand that one back.  bench.py and the product import no torch: they get the system library either way.)
int rccl_load() {
    if (g_rccl.handle) return PLONK_OK;
    std::vector<std::string> names;
    if (const char* e = getenv("PLONK_RCCL_LIB"))
        if (*e) names.push_back(e);
    if (const char* e = getenv("ROCM_PATH"))
        if (*e) names.push_back(std::string(e) + "/lib/librccl.so.1");
    names.push_back("/opt/rocm/lib/librccl.so.1");
    names.push_back("librccl.so.1");
    names.push_back("librccl.so");
    void* h = nullptr;
    std::string tried;
    for (const std::string& nm : names) {
        if ((h = dlopen(nm.c_str(), RTLD_NOW | RTLD_LOCAL))) {
            g_rccl.asked = nm;
            break;
        }
        tried += nm + " ";
        if (getenv("PLONK_RCCL_LIB") && nm == getenv("PLONK_RCCL_LIB")) break;  // an explicit choice does not fall through
    }
    PLONK_REQUIRE(h, PLONK_ERR_STATE, "librccl could not be loaded (tried: %s): %s", tried.c_str(), dlerror());
#define PLONK_RCCL_SYM(field, sym)                                                        \
    *(void**)(&g_rccl.field) = dlsym(h, sym);                                             \
    PLONK_REQUIRE(g_rccl.field, PLONK_ERR_STATE, "librccl does not export %s", sym)
    PLONK_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
    PLONK_RCCL_SYM(CommInitRank, "ncclCommInitRank");
    PLONK_RCCL_SYM(CommDestroy, "ncclCommDestroy");
    PLONK_RCCL_SYM(AllGather, "ncclAllGather");
    PLONK_RCCL_SYM(AllReduce, "ncclAllReduce");
    PLONK_RCCL_SYM(Send, "ncclSend");
    PLONK_RCCL_SYM(Recv, "ncclRecv");
    PLONK_RCCL_SYM(GroupStart, "ncclGroupStart");
    PLONK_RCCL_SYM(GroupEnd, "ncclGroupEnd");
    PLONK_RCCL_SYM(GetVersion, "ncclGetVersion");
    PLONK_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef PLONK_RCCL_SYM
    *(void**)(&g_rccl.CommAbort) = dlsym(h, "ncclCommAbort");
    *(void**)(&g_rccl.CommGetAsyncError) = dlsym(h, "ncclCommGetAsyncError");
    g_rccl.handle = h;
    return PLONK_OK;
}

// Deadlines.  A collective waits for EVERY rank: with one rank dead or stuck, ncclCommInitRank and any
// hipStreamSynchronize behind a collective wait for ever, and an 8-GPU job ends in its launcher's timeout with nothing learned.
// Every wait of this file therefore has a deadline — process default: $PLONK_COMM_TIMEOUT_S, else 600 s; 0 = none — after which
// the communicator is aborted (ncclCommAbort) and the call returns PLONK_ERR_TIMEOUT naming rank and operation.
std::atomic<double> g_default_timeout_s{-1.0};
double default_timeout_s() {
    double t = g_default_timeout_s.load();
    if (t >= 0) return t;
    const char* e = getenv("PLONK_COMM_TIMEOUT_S");
    t = (e && *e) ? atof(e) : 600.0;
    if (t < 0) t = 0;
    g_default_timeout_s.store(t);
    return t;
}
}  // namespace

#define PLONK_CHECK_RCCL(expr)                                                                              \
    do {                                                                                                    \
        ncclResult_t r_ = (expr);                                                                           \
        if (r_ != ncclSuccess) {                                                                            \
            plonk_set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), __FILE__, __LINE__); \
            return PLONK_ERR_HIP;                                                                           \
        }                                                                                                   \
    } while (0)

struct plonk_comm {
    plonk_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    uint8_t* d_buf = nullptr;  // [send | recv] staging in HBM
    size_t cap = 0;
    std::vector<hipEvent_t> events;  // plonk_gather_proofs_device: one per prover stream
    hipEvent_t ev_free = nullptr;    //   "the send buffer may be overwritten": recorded on the communicator's stream
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr, ev_t2 = nullptr;  // timing: collective start / end, host copy end
    bool timed = false;
    uint64_t collectives = 0;  // RCCL calls issued through this communicator (tests: the one-rank legs really call RCCL)
    double timeout_s = 0;      // deadline of every wait behind a collective (0 = none)
};

// ncclCommAbort, once: the communicator is unusable afterwards (every entry point checks c->comm)
static void comm_abort(plonk_comm* c) {
    if (!c->comm) return;
    if (g_rccl.CommAbort) g_rccl.CommAbort(c->comm);  // (without the symbol the handle is dropped: the process is about to exit)
    c->comm = nullptr;
}

// hipStreamSynchronize(the communicator's stream) with the deadline: polls the stream and RCCL's asynchronous error state.
// The first polls spin (a step's all-gather is over in microseconds), later ones sleep 50 us.
static int comm_wait(plonk_comm* c, const char* what) {
    hipStream_t s = c->ctx->stream;
    if (c->timeout_s <= 0) {
        PLONK_CHECK_HIP(hipStreamSynchronize(s));
        return PLONK_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned polls = 0;; polls++) {
        hipError_t q = hipStreamQuery(s);
        if (q == hipSuccess) return PLONK_OK;
        if (q != hipErrorNotReady) {
            plonk_set_error("%s on rank %d of %d: the stream reports %s", what, c->rank, c->world, hipGetErrorString(q));
            return PLONK_ERR_HIP;
        }
        if (polls >= 256) {
            if (c->comm && g_rccl.CommGetAsyncError && (polls & 63) == 0) {
                ncclResult_t ar = ncclSuccess;
                if (g_rccl.CommGetAsyncError(c->comm, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress) {
                    plonk_set_error("%s on rank %d of %d: RCCL reports an asynchronous error: %s", what, c->rank, c->world, g_rccl.GetErrorString(ar));
                    comm_abort(c);
                    return PLONK_ERR_HIP;
                }
            }
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (el > c->timeout_s) {
                plonk_set_error("%s did not complete within %.0f s on rank %d of %d (device %d): another rank is dead or stuck — the "
                                "communicator has been aborted", what, c->timeout_s, c->rank, c->world, c->ctx->device);
                comm_abort(c);
                return PLONK_ERR_TIMEOUT;
            }
            usleep(50);
        }
    }
}
#define PLONK_COMM_LIVE(c) PLONK_REQUIRE((c)->comm, PLONK_ERR_STATE, "the communicator was aborted after a failed or timed-out collective")

static int comm_staging(plonk_comm* c, size_t bytes) {
    if (c->cap >= bytes) return PLONK_OK;
    if (c->d_buf) {
        PLONK_CHECK_HIP(hipStreamSynchronize(c->ctx->stream));
        hipFree(c->d_buf);
        c->d_buf = nullptr;
        c->cap = 0;
    }
    void* p = nullptr;
    if (!plonk_dev_malloc(&p, bytes)) {
        plonk_set_error("hipMalloc of %zu gather-staging bytes failed", bytes);
        return PLONK_ERR_NOMEM;
    }
    c->d_buf = (uint8_t*)p;
    c->cap = bytes;
    return PLONK_OK;
}

extern "C" {

int plonk_comm_unique_id(uint8_t out_id[PLONK_COMM_ID_BYTES]) {
    PLONK_REQUIRE(out_id, PLONK_ERR_ARG, "out_id is NULL");
    static_assert(sizeof(ncclUniqueId) == PLONK_COMM_ID_BYTES, "ncclUniqueId size");
    PLONK_TRY(rccl_load());
    ncclUniqueId id;
    PLONK_CHECK_RCCL(g_rccl.GetUniqueId(&id));
    memcpy(out_id, &id, sizeof id);
    return PLONK_OK;
}

int plonk_comm_create(plonk_ctx* ctx, const uint8_t id_bytes[PLONK_COMM_ID_BYTES], int rank, int world, plonk_comm** out) {
    PLONK_REQUIRE(ctx && id_bytes && out && world >= 1 && rank >= 0 && rank < world, PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(ctx);
    PLONK_TRY(rccl_load());
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof id);
    plonk_comm* c = new plonk_comm();
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    c->timeout_s = default_timeout_s();
    ncclResult_t r = ncclSuccess;
    if (c->timeout_s <= 0) {
        r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    } else {
        // ncclCommInitRank blocks until all `world` ranks have called it.  It runs on a helper thread so that this one can give up:
        // on a timeout the helper is left behind (it may still be inside RCCL) — the caller is expected to report and exit
        struct Job {
            std::mutex m;
            std::condition_variable cv;
            bool done = false;
            ncclResult_t r = ncclSuccess;
            ncclComm_t comm = nullptr;
        };
        auto job = std::make_shared<Job>();
        const int device = ctx->device;
        std::thread([job, id, world, rank, device]() {
            ncclComm_t cm = nullptr;
            ncclResult_t rr = hipSetDevice(device) == hipSuccess ? g_rccl.CommInitRank(&cm, world, id, rank) : ncclUnhandledCudaError;
            std::lock_guard<std::mutex> lk(job->m);
            job->comm = cm;
            job->r = rr;
            job->done = true;
            job->cv.notify_all();
        }).detach();
        std::unique_lock<std::mutex> lk(job->m);
        if (!job->cv.wait_for(lk, std::chrono::duration<double>(c->timeout_s), [&] { return job->done; })) {
            plonk_set_error("ncclCommInitRank(rank %d of %d, device %d) did not return within %.0f s: not every rank reached the "
                            "communicator (a rank died before it, or the ranks disagree on the unique id / world size)", rank, world, device, c->timeout_s);
            delete c;
            return PLONK_ERR_TIMEOUT;
        }
        r = job->r;
        c->comm = job->comm;
    }
    if (r != ncclSuccess) {
        plonk_set_error("ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world, ctx->device, g_rccl.GetErrorString(r));
        c->comm = nullptr;
        delete c;
        return PLONK_ERR_HIP;
    }
    if (hipEventCreateWithFlags(&c->ev_free, hipEventDisableTiming) != hipSuccess || hipEventCreate(&c->ev_t0) != hipSuccess ||
        hipEventCreate(&c->ev_t1) != hipSuccess || hipEventCreate(&c->ev_t2) != hipSuccess) {
        plonk_set_error("hipEventCreate for a communicator failed");
        plonk_comm_destroy(c);
        return PLONK_ERR_HIP;
    }
    *out = c;
    return PLONK_OK;
}

int plonk_comm_destroy(plonk_comm* c) {
    if (!c) return PLONK_OK;
    plonk_use_device(c->ctx->device);
    if (c->comm && comm_wait(c, "plonk_comm_destroy") == PLONK_OK && c->comm) g_rccl.CommDestroy(c->comm);  // (a timed-out wait has aborted it)
    if (c->d_buf) hipFree(c->d_buf);
    for (hipEvent_t e : c->events) hipEventDestroy(e);
    for (hipEvent_t e : {c->ev_free, c->ev_t0, c->ev_t1, c->ev_t2})
        if (e) hipEventDestroy(e);
    delete c;
    return PLONK_OK;
}

int plonk_comm_set_default_timeout(double seconds) {
    PLONK_REQUIRE(seconds >= 0, PLONK_ERR_ARG, "a timeout is >= 0 seconds (0 = wait for ever)");
    g_default_timeout_s.store(seconds);
    return PLONK_OK;
}

int plonk_comm_set_timeout(plonk_comm* c, double seconds) {
    PLONK_REQUIRE(c && seconds >= 0, PLONK_ERR_ARG, "bad argument");
    c->timeout_s = seconds;
    return PLONK_OK;
}

// out_row[p] = 1 iff `device` can map device p's memory (hipDeviceCanAccessPeer; the diagonal is 1): what RCCL's P2P / xGMI
// transport needs between two ranks of one node.  bench.py --preflight prints the row per rank.
int plonk_device_peer_access(int device, int* out_row, size_t cap) {
    PLONK_REQUIRE(out_row && cap, PLONK_ERR_ARG, "bad argument");
    int n = 0;
    PLONK_CHECK_HIP(hipGetDeviceCount(&n));
    PLONK_REQUIRE(device >= 0 && device < n, PLONK_ERR_ARG, "device %d of %d", device, n);
    for (int p = 0; p < n && (size_t)p < cap; p++) {
        int ok = 1;
        if (p != device) PLONK_CHECK_HIP(hipDeviceCanAccessPeer(&ok, device, p));
        out_row[p] = ok;
    }
    return PLONK_OK;
}

int plonk_comm_size(const plonk_comm* c, int* out_rank, int* out_world) {
    PLONK_REQUIRE(c && out_rank && out_world, PLONK_ERR_ARG, "bad argument");
    *out_rank = c->rank;
    *out_world = c->world;
    return PLONK_OK;
}

// h_recv[r * bytes_per_rank ..] = rank r's h_send, for every r: one ncclAllGather of uint8 on the context's stream
int plonk_gather_results(plonk_comm* c, const uint8_t* h_send, size_t bytes_per_rank, uint8_t* h_recv) {
    PLONK_REQUIRE(c && h_send && h_recv && bytes_per_rank, PLONK_ERR_ARG, "bad argument");
    PLONK_COMM_LIVE(c);
    PLONK_ENTER(c->ctx);
    const size_t total = bytes_per_rank * (size_t)c->world;
    PLONK_TRY(comm_staging(c, bytes_per_rank + total));
    hipStream_t s = c->ctx->stream;
    uint8_t *d_send = c->d_buf, *d_recv = c->d_buf + bytes_per_rank;
    PLONK_CHECK_HIP(hipMemcpyAsync(d_send, h_send, bytes_per_rank, hipMemcpyHostToDevice, s));
    PLONK_CHECK_RCCL(g_rccl.AllGather(d_send, d_recv, bytes_per_rank, ncclUint8, c->comm, s));
    c->collectives++;
    PLONK_CHECK_HIP(hipMemcpyAsync(h_recv, d_recv, total, hipMemcpyDeviceToHost, s));
    return comm_wait(c, "plonk_gather_results (ncclAllGather)");
}

// The gather of a step's proofs without the host round trip of plonk_gather_results (D -> H -> D -> all-gather -> D -> H):
// every prover of this rank packs its resident batch — records of 768 bytes, or 480 compressed — and its status bytes
// straight into the send buffer on its own stream, the communicator's stream waits for those streams' events, ONE
// ncclAllGather moves proofs and status bytes of all ranks over xGMI, and one copy brings them to the host.
// h_recv[r] = [n_provers * batch records | n_provers * batch status bytes, padded to 16] of rank r.
int plonk_gather_proofs_device(plonk_comm* c, plonk_prover* const* provers, size_t n_provers, size_t batch, int compressed, uint8_t* h_recv) {
    PLONK_REQUIRE(c && provers && n_provers && batch && h_recv, PLONK_ERR_ARG, "bad argument");
    PLONK_COMM_LIVE(c);
    PLONK_ENTER(c->ctx);
    const size_t rec = compressed ? 480 : 768;
    const size_t n = n_provers * batch;
    const size_t per_rank = n * rec + ((n + 15) & ~(size_t)15);
    const size_t total = per_rank * (size_t)c->world;
    PLONK_TRY(comm_staging(c, per_rank + total));
    hipStream_t s = c->ctx->stream;
    uint8_t *d_send = c->d_buf, *d_recv = c->d_buf + per_rank;
    // nothing is enqueued before every argument has been checked
    for (size_t k = 0; k < n_provers; k++)
        PLONK_REQUIRE(provers[k] && prover_ctx(provers[k])->device == c->ctx->device, PLONK_ERR_ARG, "prover %zu lives on another device than the communicator", k);
    while (c->events.size() < n_provers) {
        hipEvent_t e;
        PLONK_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->events.push_back(e);
    }
    PLONK_CHECK_HIP(hipMemsetAsync(d_send + n * rec, 0, per_rank - n * rec, s));
    // the send buffer may still be read by the previous gather: ONE "buffer free" event on the communicator's stream, which
    // every prover stream waits for — the provers then pack concurrently, each on its own stream (recording the event once per
    // prover, behind the wait on the previous prover's pack, serialised them) — and the communicator's stream waits for all
    PLONK_CHECK_HIP(hipEventRecord(c->ev_free, s));
    int rc = PLONK_OK;
    size_t packed = 0;
    for (; packed < n_provers && rc == PLONK_OK; packed++) {
        hipStream_t ps = prover_ctx(provers[packed])->stream;
        if (hipStreamWaitEvent(ps, c->ev_free, 0) != hipSuccess) {
            plonk_set_error("hipStreamWaitEvent on prover %zu's stream failed", packed);
            rc = PLONK_ERR_HIP;
            break;
        }
        rc = prover_pack_device(provers[packed], batch, compressed, d_send + packed * batch * rec, d_send + n * rec + packed * batch, c->events[packed]);
    }
    for (size_t k = 0; k < packed; k++) (void)hipStreamWaitEvent(s, c->events[k], 0);  // also on failure: the packs enqueued so far finish before the buffer is reused
    if (rc != PLONK_OK) {
        (void)hipStreamSynchronize(s);
        return rc;
    }
    // one rank is not a special case: the all-gather of a one-rank communicator is RCCL's own copy (and the only way this
    // call path can be exercised on a one-GPU box)
    PLONK_CHECK_HIP(hipEventRecord(c->ev_t0, s));
    PLONK_CHECK_RCCL(g_rccl.AllGather(d_send, d_recv, per_rank, ncclUint8, c->comm, s));
    c->collectives++;
    PLONK_CHECK_HIP(hipEventRecord(c->ev_t1, s));
    PLONK_CHECK_HIP(hipMemcpyAsync(h_recv, d_recv, total, hipMemcpyDeviceToHost, s));
    PLONK_CHECK_HIP(hipEventRecord(c->ev_t2, s));
    PLONK_TRY(comm_wait(c, "plonk_gather_proofs_device (ncclAllGather of the step's proofs)"));
    c->timed = true;
    return PLONK_OK;
}

// device time of the last plonk_gather_proofs_device: the ncclAllGather itself, and the copy of all ranks' records to the host
int plonk_comm_last_gather_ms(plonk_comm* c, float* out_allgather_ms, float* out_to_host_ms) {
    PLONK_REQUIRE(c && out_allgather_ms && out_to_host_ms, PLONK_ERR_ARG, "bad argument");
    PLONK_REQUIRE(c->timed, PLONK_ERR_STATE, "no plonk_gather_proofs_device has completed on this communicator");
    PLONK_ENTER(c->ctx);
    PLONK_CHECK_HIP(hipEventElapsedTime(out_allgather_ms, c->ev_t0, c->ev_t1));
    PLONK_CHECK_HIP(hipEventElapsedTime(out_to_host_ms, c->ev_t1, c->ev_t2));
    return PLONK_OK;
}

// which RCCL this process talks to: the file the loaded ncclGetUniqueId lives in (dladdr), ncclGetVersion's code
// (major * 10000 + minor * 100 + patch), and the number of RCCL collectives / point-to-point groups this communicator has
// issued (comm may be NULL: the library is loaded if it was not).
int plonk_comm_info(const plonk_comm* c, char* out_path, size_t path_cap, int* out_version, uint64_t* out_collectives) {
    PLONK_TRY(rccl_load());
    if (out_path && path_cap) {
        Dl_info di;
        const char* nm = (dladdr((void*)g_rccl.GetUniqueId, &di) && di.dli_fname) ? di.dli_fname : g_rccl.asked.c_str();
        strncpy(out_path, nm, path_cap - 1);
        out_path[path_cap - 1] = 0;
    }
    if (out_version) PLONK_CHECK_RCCL(g_rccl.GetVersion(out_version));
    if (out_collectives) *out_collectives = c ? c->collectives : 0;
    return PLONK_OK;
}

// *inout = max over ranks (bench.py: the step time is the slowest rank's); also serves as the barrier
int plonk_comm_max_f64(plonk_comm* c, double* inout) {
    PLONK_REQUIRE(c && inout, PLONK_ERR_ARG, "bad argument");
    PLONK_COMM_LIVE(c);
    PLONK_ENTER(c->ctx);
    PLONK_TRY(comm_staging(c, 64));
    hipStream_t s = c->ctx->stream;
    double* d = (double*)c->d_buf;
    PLONK_CHECK_HIP(hipMemcpyAsync(d, inout, sizeof(double), hipMemcpyHostToDevice, s));
    PLONK_CHECK_RCCL(g_rccl.AllReduce(d, d, 1, ncclDouble, ncclMax, c->comm, s));
    c->collectives++;
    PLONK_CHECK_HIP(hipMemcpyAsync(inout, d, sizeof(double), hipMemcpyDeviceToHost, s));
    return comm_wait(c, "plonk_comm_max_f64 / plonk_comm_barrier (ncclAllReduce)");
}

// d_recv[r * bytes_per_peer ..] = block `rank` of rank r's d_send, for every r: the transpose step of the distributed
// NTT, as one group of point-to-point ncclSend / ncclRecv pairs over xGMI (device buffers, the context's stream)
int plonk_comm_all_to_all(plonk_comm* c, const void* d_send, void* d_recv, size_t bytes_per_peer) {
    PLONK_REQUIRE(c && d_send && d_recv && bytes_per_peer, PLONK_ERR_ARG, "bad argument");
    PLONK_COMM_LIVE(c);
    PLONK_ENTER(c->ctx);
    hipStream_t s = c->ctx->stream;
    // (one rank included: a send / receive pair to oneself inside a group is RCCL's own copy)
    PLONK_CHECK_RCCL(g_rccl.GroupStart());
    for (int r = 0; r < c->world; r++) {
        PLONK_CHECK_RCCL(g_rccl.Send((const uint8_t*)d_send + (size_t)r * bytes_per_peer, bytes_per_peer, ncclUint8, r, c->comm, s));
        PLONK_CHECK_RCCL(g_rccl.Recv((uint8_t*)d_recv + (size_t)r * bytes_per_peer, bytes_per_peer, ncclUint8, r, c->comm, s));
    }
    PLONK_CHECK_RCCL(g_rccl.GroupEnd());
    c->collectives++;
    return PLONK_OK;
}

// One transform of 2^log_n points across the communicator's W = 2^k GPUs (four-step): local column
// transforms with the inter-pass twiddles, ONE all-to-all, local row transforms.  Layouts: see plonk_hip.h.
int plonk_fr_ntt_distributed(plonk_comm* c, const void* d_in, void* d_out, unsigned log_n, int inverse) {
    PLONK_REQUIRE(c && d_in && d_out, PLONK_ERR_ARG, "bad argument");
    PLONK_ENTER(c->ctx);
    unsigned log_w = 0;
    while ((1 << log_w) < c->world) log_w++;
    PLONK_REQUIRE((1 << log_w) == c->world, PLONK_ERR_ARG, "the communicator has %d ranks: a power of two is needed", c->world);
    unsigned r1, r2;
    PLONK_TRY(ntt_dist_plan(log_n, log_w, &r1, &r2));
    const size_t local = (size_t)1 << (log_n - log_w);
    void* sc;
    PLONK_TRY(ctx_scratch(c->ctx, 0, 2 * local * sizeof(Fr), &sc));
    Fr* cols = (Fr*)sc;
    Fr* recv = cols + local;
    PLONK_TRY(ntt_dist_columns(c->ctx, (const Fr*)d_in, cols, log_n, log_w, (unsigned)c->rank, inverse != 0));
    PLONK_TRY(plonk_comm_all_to_all(c, cols, recv, (local >> log_w) * sizeof(Fr)));
    return ntt_dist_rows(c->ctx, recv, (Fr*)d_out, log_n, log_w, (unsigned)c->rank, inverse != 0);
}

int plonk_comm_barrier(plonk_comm* c) {
    double zero = 0;
    return plonk_comm_max_f64(c, &zero);
}

}  // extern "C"
