// hip_emu.cpp — TEST INFRASTRUCTURE ONLY (see hip_emu.h).
#include "hip_emu.h"

#include <stdio.h>
#include <sys/mman.h>
#include <time.h>
#include <ucontext.h>

#include <vector>

namespace hipemu {
Idx3 g_threadIdx, g_blockIdx;
dim3 g_blockDim, g_gridDim;
unsigned char* g_dyn_smem = nullptr;

static const size_t kStack = 256 * 1024;
struct Fiber { ucontext_t ctx; void* stack = nullptr; bool done = true; };
static std::vector<Fiber> g_fibers;
static ucontext_t g_sched;
static const std::function<void()>* g_body = nullptr;
static int g_cur = -1, g_nthreads = 0;
static unsigned g_bar_count = 0, g_bar_gen = 0;
static unsigned g_wave_count[64], g_wave_gen[64];
static uint64_t g_wave_buf[64][64];

static void yield_to_sched() { swapcontext(&g_fibers[g_cur].ctx, &g_sched); }

static void fiber_entry() {
    (*g_body)();
    g_fibers[g_cur].done = true;
    swapcontext(&g_fibers[g_cur].ctx, &g_sched);
}

void syncthreads() {
    unsigned gen = g_bar_gen;
    if (++g_bar_count == (unsigned)g_nthreads) { g_bar_count = 0; g_bar_gen++; return; }
    while (g_bar_gen == gen) yield_to_sched();
}

static void wave_sync(int w, unsigned lanes) {
    unsigned gen = g_wave_gen[w];
    if (++g_wave_count[w] == lanes) { g_wave_count[w] = 0; g_wave_gen[w]++; return; }
    while (g_wave_gen[w] == gen) yield_to_sched();
}

static unsigned lanes_in_wave(int w) {
    int rem = g_nthreads - w * 64;
    return rem >= 64 ? 64u : (unsigned)rem;
}

uint64_t shfl64(uint64_t v, int src) {
    int w = g_cur >> 6, lane = g_cur & 63;
    unsigned n = lanes_in_wave(w);
    g_wave_buf[w][lane] = v;
    wave_sync(w, n);
    uint64_t r = g_wave_buf[w][(unsigned)(src & 63) < n ? (src & 63) : lane];
    wave_sync(w, n);
    return r;
}

uint64_t ballot(int pred) {
    int w = g_cur >> 6, lane = g_cur & 63;
    unsigned n = lanes_in_wave(w);
    g_wave_buf[w][lane] = pred ? 1 : 0;
    wave_sync(w, n);
    uint64_t r = 0;
    for (unsigned i = 0; i < n; i++) r |= (uint64_t)(g_wave_buf[w][i] & 1) << i;
    wave_sync(w, n);
    return r;
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
    int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads <= 0 || nthreads > 1024) { fprintf(stderr, "hipemu: bad block size %d\n", nthreads); abort(); }
    if ((int)g_fibers.size() < nthreads) g_fibers.resize(nthreads);
    for (int t = 0; t < nthreads; t++)
        if (!g_fibers[t].stack) {
            g_fibers[t].stack = mmap(nullptr, kStack, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
            if (g_fibers[t].stack == MAP_FAILED) { perror("hipemu mmap"); abort(); }
        }
    // dynamic LDS: a fresh allocation of EXACTLY the launch's size, so that the AddressSanitizer build (make asan) sees a
    // kernel that indexes past the LDS it asked for
    void* smem = nullptr;
    if (posix_memalign(&smem, 64, shmem ? shmem : 1) != 0) { perror("hipemu posix_memalign"); abort(); }
    g_dyn_smem = (unsigned char*)smem;
    g_blockDim = block;
    g_gridDim = grid;
    g_nthreads = nthreads;
    g_body = &body;
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) {
                g_blockIdx = {bx, by, bz};
                g_bar_count = 0;
                memset(g_wave_count, 0, sizeof g_wave_count);
                for (int t = 0; t < nthreads; t++) {
                    Fiber& f = g_fibers[t];
                    getcontext(&f.ctx);
                    f.ctx.uc_stack.ss_sp = f.stack;
                    f.ctx.uc_stack.ss_size = kStack;
                    f.ctx.uc_link = nullptr;
                    f.done = false;
                    makecontext(&f.ctx, fiber_entry, 0);
                }
                int alive = nthreads;
                // Scheduling order of the fibres between synchronisation points.  Any order is a legal GPU schedule for
                // code that only communicates through barriers, wave-level primitives and atomics, so a result that
                // changes under PLONK_EMU_SCHED=reverse (or random:<seed>) means a missing barrier / a race.
                static const char* sched_env = getenv("PLONK_EMU_SCHED");
                static unsigned sched_rng = sched_env && !strncmp(sched_env, "random", 6) ? (unsigned)atoi(sched_env + (sched_env[6] ? 7 : 6)) * 2654435761u + 12345u : 0u;
                static bool sched_said = false;
                if (sched_env && !sched_said) {
                    fprintf(stderr, "hipemu: fibre schedule = %s\n", sched_env);
                    sched_said = true;
                }
                std::vector<int> order(nthreads);
                for (int t = 0; t < nthreads; t++) order[t] = sched_env && !strcmp(sched_env, "reverse") ? nthreads - 1 - t : t;
                while (alive > 0) {
                    int progressed = 0;
                    if (sched_rng)
                        for (int t = nthreads - 1; t > 0; t--) {
                            sched_rng = sched_rng * 1664525u + 1013904223u;
                            std::swap(order[t], order[(sched_rng >> 8) % (unsigned)(t + 1)]);
                        }
                    for (int ti = 0; ti < nthreads; ti++) {
                        const int t = order[ti];
                        Fiber& f = g_fibers[t];
                        if (f.done) continue;
                        g_cur = t;
                        g_threadIdx = {(unsigned)t % block.x, ((unsigned)t / block.x) % block.y, (unsigned)t / (block.x * block.y)};
                        swapcontext(&g_sched, &f.ctx);
                        progressed = 1;
                        if (f.done) alive--;
                    }
                    if (!progressed) break;
                }
            }
    g_cur = -1;
    g_body = nullptr;
    g_dyn_smem = nullptr;
    free(smem);
}
}  // namespace hipemu

struct hipemuEvent { timespec ts; };

hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static thread_local int g_emu_device = 0;
hipError_t hipSetDevice(int d) { g_emu_device = d; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = g_emu_device; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    memset(p, 0, sizeof *p);
    snprintf(p->name, sizeof p->name, "hipemu (CPU fibers; tests only)");
    snprintf(p->gcnArchName, sizeof p->gcnArchName, "emu");
    p->multiProcessorCount = 1;
    return hipSuccess;
}
hipError_t hipMalloc(void** p, size_t n) { return posix_memalign(p, 256, n ? n : 256) ? hipErrorOutOfMemory : hipSuccess; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemuEvent(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { clock_gettime(CLOCK_MONOTONIC, &e->ts); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = (float)((b->ts.tv_sec - a->ts.tv_sec) * 1e3 + (b->ts.tv_nsec - a->ts.tv_nsec) * 1e-6);
    return hipSuccess;
}
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
