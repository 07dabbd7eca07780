// libcattus_pool.so: the C ABI of include/cattus_pool.h over RCCL.  Host code only (HIP runtime + RCCL calls, no kernel).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "../../include/cattus_pool.h"

#define POOL_API extern "C" __attribute__((visibility("default")))

static_assert(CATTUS_POOL_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id crosses the ABI as raw bytes");

namespace {
thread_local std::string g_err;
int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIP_OK(expr)                                                                              \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess) return fail(-3, "%s failed: %s", #expr, hipGetErrorString(e__));   \
    } while (0)
#define NCCL_OK(expr)                                                                             \
    do {                                                                                          \
        ncclResult_t r__ = (expr);                                                                \
        if (r__ != ncclSuccess && r__ != ncclInProgress) return fail(-3, "%s failed: %s", #expr, ncclGetErrorString(r__)); \
    } while (0)
struct DevMem {
    void* p = nullptr;
    ~DevMem() {
        if (p) (void)hipFree(p);
    }
};
}  // namespace

struct cattus_pool {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int rank = 0, world = 1, device = 0;
    double timeout_s = 600.0;  // deadline of every wait inside an entry point (cattus_pool_set_timeout)
    bool dead = false;         // the communicator was aborted (a deadline passed, RCCL reported an error): the handle only destroys
};

namespace {
using Clock = std::chrono::steady_clock;

// The communicator is NON-BLOCKING (ncclCommInitRankConfig, blocking = 0): no RCCL call parks the host thread, so a peer that
// died cannot hang this rank -- the reference's own TODO (training/self-play/src/self_play.rs:128).  Every wait is a poll with a
// deadline; when it passes, ncclCommAbort tears the communicator down (that also ends kernels of it still spinning on the
// device) and the entry point returns CATTUS_POOL_E_TIMEOUT.  The handle is dead from then on.
int abort_comm(cattus_pool* p, int code, const char* what) {
    if (p->comm) (void)ncclCommAbort(p->comm);
    p->comm = nullptr;
    p->dead = true;
    return fail(code, "%s (communicator aborted; destroy this pool and make a new one with the surviving ranks)", what);
}

// until RCCL has finished what the last non-blocking call started (ncclInProgress -> ncclSuccess), or the deadline
int wait_comm(cattus_pool* p, Clock::time_point deadline, const char* what) {
    for (;;) {
        ncclResult_t st = ncclSuccess;
        const ncclResult_t r = ncclCommGetAsyncError(p->comm, &st);
        if (r != ncclSuccess) return abort_comm(p, -3, ncclGetErrorString(r));
        if (st == ncclSuccess) return 0;
        if (st != ncclInProgress) return abort_comm(p, -3, ncclGetErrorString(st));
        if (Clock::now() >= deadline) {
            char buf[160];
            snprintf(buf, sizeof buf, "%s: no completion within %.3g s (a peer is gone or never called)", what, p->timeout_s);
            return abort_comm(p, CATTUS_POOL_E_TIMEOUT, buf);
        }
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
}

// until everything enqueued on the pool's stream has run, or the deadline
int wait_stream(cattus_pool* p, Clock::time_point deadline, const char* what) {
    for (;;) {
        const hipError_t q = hipStreamQuery(p->stream);
        if (q == hipSuccess) return 0;
        if (q != hipErrorNotReady) return abort_comm(p, -3, hipGetErrorString(q));
        if (p->comm) {  // an asynchronous RCCL error (a peer's process died) shows here first
            ncclResult_t st = ncclSuccess;
            if (ncclCommGetAsyncError(p->comm, &st) == ncclSuccess && st != ncclSuccess && st != ncclInProgress) return abort_comm(p, -3, ncclGetErrorString(st));
        }
        if (Clock::now() >= deadline) {
            char buf[160];
            snprintf(buf, sizeof buf, "%s: the collective did not complete within %.3g s (a peer is gone or never called)", what, p->timeout_s);
            const int rc = abort_comm(p, CATTUS_POOL_E_TIMEOUT, buf);
            (void)hipStreamSynchronize(p->stream);  // the aborted kernels leave the stream
            return rc;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}
Clock::time_point deadline_of(const cattus_pool* p) {
    return Clock::now() + std::chrono::duration_cast<Clock::duration>(std::chrono::duration<double>(p->timeout_s));
}
#define ALIVE(p)                                                                                          \
    do {                                                                                                  \
        if ((p)->dead) return fail(CATTUS_POOL_E_STATE, "this pool's communicator was aborted earlier");  \
    } while (0)
#define WAIT_OK(expr)          \
    do {                       \
        const int w__ = (expr); \
        if (w__) return w__;   \
    } while (0)
}  // namespace

POOL_API const char* cattus_pool_last_error(void) { return g_err.c_str(); }

POOL_API int cattus_pool_unique_id(uint8_t id[CATTUS_POOL_ID_BYTES]) {
    if (!id) return fail(-1, "id is NULL");
    ncclUniqueId u;
    NCCL_OK(ncclGetUniqueId(&u));
    memcpy(id, u.internal, CATTUS_POOL_ID_BYTES);
    return 0;
}

POOL_API int cattus_pool_create(const uint8_t id[CATTUS_POOL_ID_BYTES], int rank, int world, int device, cattus_pool** out) {
    if (!id || !out) return fail(-1, "NULL argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(-1, "rank %d of %d", rank, world);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(-3, "no usable HIP device %d", device);
    HIP_OK(hipSetDevice(device));
    cattus_pool* p = new cattus_pool;
    p->rank = rank, p->world = world, p->device = device;
    ncclUniqueId u;
    memcpy(u.internal, id, CATTUS_POOL_ID_BYTES);
    ncclConfig_t config = NCCL_CONFIG_INITIALIZER;
    config.blocking = 0;
    ncclResult_t r = ncclCommInitRankConfig(&p->comm, world, u, rank, &config);
    if (r != ncclSuccess && r != ncclInProgress) {
        delete p;
        return fail(-3, "ncclCommInitRankConfig failed: %s", ncclGetErrorString(r));
    }
    if (const int w = wait_comm(p, deadline_of(p), "cattus_pool_create")) {  // every rank has to call within the deadline
        delete p;
        return w;
    }
    if (hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) != hipSuccess) {
        (void)ncclCommAbort(p->comm);
        delete p;
        return fail(-3, "hipStreamCreate failed");
    }
    *out = p;
    return 0;
}

POOL_API int cattus_pool_set_timeout(cattus_pool* p, double seconds) {
    if (!p || !(seconds >= 0)) return fail(-1, "bad argument");
    p->timeout_s = seconds;
    return 0;
}

// Diagnostic (tests): a collective that does not complete -- what the pooling collectives are to the survivors when a peer died
// before calling them.  With one rank nothing can be left unmatched (RCCL refuses a lone self-receive as invalid usage), so the
// pool's stream is held by a host callback and an all-reduce is enqueued behind it: the wait sees a collective that is not
// finishing, the deadline passes, the communicator is aborted; then the callback is released, and the all-reduce's kernel -- now
// of an aborted communicator -- leaves the stream by itself.  Returns what the failure path returns: CATTUS_POOL_E_TIMEOUT.
namespace {
struct Gate {
    std::atomic<bool> open{false};
};
void hold_stream(void* arg) {
    Gate* g = static_cast<Gate*>(arg);
    while (!g->open.load(std::memory_order_acquire)) std::this_thread::sleep_for(std::chrono::microseconds(100));
}
}  // namespace

POOL_API int cattus_pool_debug_stalled_collective(cattus_pool* p) {
    if (!p) return fail(-1, "NULL argument");
    ALIVE(p);
    HIP_OK(hipSetDevice(p->device));
    DevMem d;
    HIP_OK(hipMalloc(&d.p, 256));
    Gate gate;
    const auto deadline = deadline_of(p);
    // the stream is released 200 ms behind the deadline whatever this thread is doing then (ncclCommAbort waits for the communicator's
    // enqueued kernel, which cannot start while the stream is held)
    std::thread releaser([&gate, deadline] {
        std::this_thread::sleep_until(deadline + std::chrono::milliseconds(200));
        gate.open.store(true, std::memory_order_release);
    });
    int rc = 0;
    if (hipLaunchHostFunc(p->stream, hold_stream, &gate) != hipSuccess) rc = fail(-3, "hipLaunchHostFunc failed");
    if (rc == 0) {
        const ncclResult_t r = ncclAllReduce(d.p, d.p, 32, ncclUint64, ncclSum, p->comm, p->stream);
        rc = (r == ncclSuccess || r == ncclInProgress) ? wait_comm(p, deadline, "stalled collective") : fail(-3, "ncclAllReduce failed: %s", ncclGetErrorString(r));
    }
    if (rc == 0) {
        // wait_stream without its final drain: the stream drains once the releaser has opened the gate
        for (;;) {
            const hipError_t q = hipStreamQuery(p->stream);
            if (q == hipSuccess) {
                rc = fail(-3, "the stalled collective completed before the deadline was noticed");
                break;
            }
            if (Clock::now() >= deadline) {
                char buf[160];
                snprintf(buf, sizeof buf, "stalled collective: the collective did not complete within %.3g s (a peer is gone or never called)", p->timeout_s);
                rc = abort_comm(p, CATTUS_POOL_E_TIMEOUT, buf);
                break;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }
    const std::string keep = g_err;
    gate.open.store(true, std::memory_order_release);
    releaser.join();
    (void)hipStreamSynchronize(p->stream);  // the callback returns, the aborted communicator's kernel leaves the stream
    g_err = keep;
    return rc;
}

POOL_API void cattus_pool_destroy(cattus_pool* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->comm) {
        // a clean communicator is finalised (bounded: a peer that never finalises costs the deadline, not a hang), else aborted
        const ncclResult_t r = ncclCommFinalize(p->comm);
        if ((r == ncclSuccess || r == ncclInProgress) && wait_comm(p, deadline_of(p), "cattus_pool_destroy") == 0) (void)ncclCommDestroy(p->comm);
        else if (p->comm) (void)ncclCommAbort(p->comm);
    }
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

POOL_API void cattus_pool_free(void* ptr) { free(ptr); }

POOL_API int cattus_pool_reduce_counters(cattus_pool* p, uint64_t* counters, uint32_t n) {
    if (!p || (!counters && n)) return fail(-1, "NULL argument");
    if (!n) return 0;
    ALIVE(p);
    HIP_OK(hipSetDevice(p->device));
    const auto deadline = deadline_of(p);
    DevMem d;
    HIP_OK(hipMalloc(&d.p, (size_t)n * 8));
    HIP_OK(hipMemcpyAsync(d.p, counters, (size_t)n * 8, hipMemcpyHostToDevice, p->stream));
    NCCL_OK(ncclAllReduce(d.p, d.p, n, ncclUint64, ncclSum, p->comm, p->stream));
    WAIT_OK(wait_comm(p, deadline, "cattus_pool_reduce_counters"));
    std::vector<uint64_t> back(n);  // the caller's array changes only when the collective has completed
    HIP_OK(hipMemcpyAsync(back.data(), d.p, (size_t)n * 8, hipMemcpyDeviceToHost, p->stream));
    WAIT_OK(wait_stream(p, deadline, "cattus_pool_reduce_counters"));
    memcpy(counters, back.data(), (size_t)n * 8);
    return 0;
}

POOL_API int cattus_pool_records(cattus_pool* p, const uint8_t* bytes, const uint32_t* meta, uint64_t n_local, uint32_t record_bytes,
                                 uint8_t** all_bytes, uint32_t** all_meta, uint64_t* n_total) {
    if (!p || !all_bytes || !all_meta || !n_total || (n_local && (!bytes || !meta))) return fail(-1, "NULL argument");
    if (!record_bytes) return fail(-1, "record_bytes is 0");
    *all_bytes = nullptr, *all_meta = nullptr, *n_total = 0;
    ALIVE(p);
    HIP_OK(hipSetDevice(p->device));
    const auto deadline = deadline_of(p);
    const int W = p->world;
    // 1. every rank's count (and record size, which must agree)
    DevMem dc;
    HIP_OK(hipMalloc(&dc.p, (size_t)W * 16));
    const uint64_t mine[2] = {n_local, record_bytes};
    HIP_OK(hipMemcpyAsync((char*)dc.p + (size_t)p->rank * 16, mine, 16, hipMemcpyHostToDevice, p->stream));
    NCCL_OK(ncclAllGather((char*)dc.p + (size_t)p->rank * 16, dc.p, 2, ncclUint64, p->comm, p->stream));
    WAIT_OK(wait_comm(p, deadline, "cattus_pool_records (counts)"));
    std::vector<uint64_t> counts((size_t)W * 2);
    HIP_OK(hipMemcpyAsync(counts.data(), dc.p, (size_t)W * 16, hipMemcpyDeviceToHost, p->stream));
    WAIT_OK(wait_stream(p, deadline, "cattus_pool_records (counts)"));
    uint64_t total = 0, nmax = 1;
    for (int r = 0; r < W; r++) {
        if (counts[2 * r + 1] != record_bytes) return fail(-1, "rank %d pools %llu-byte records, this rank %u-byte ones", r, (unsigned long long)counts[2 * r + 1], record_bytes);
        total += counts[2 * r], nmax = std::max(nmax, counts[2 * r]);
    }
    *n_total = total;
    // 2. payloads: [count][record bytes | 12 meta bytes], each rank's padded to the largest shard; rank 0 receives
    const size_t row = (size_t)record_bytes + 12, slab = (size_t)nmax * row;
    std::vector<uint8_t> stage(slab, 0);
    for (uint64_t i = 0; i < n_local; i++) {
        memcpy(&stage[i * row], bytes + i * record_bytes, record_bytes);
        memcpy(&stage[i * row + record_bytes], meta + 3 * i, 12);
    }
    DevMem dsend, drecv;
    HIP_OK(hipMalloc(&dsend.p, slab));
    HIP_OK(hipMemcpyAsync(dsend.p, stage.data(), slab, hipMemcpyHostToDevice, p->stream));
    if (p->rank == 0) HIP_OK(hipMalloc(&drecv.p, slab * W));
    NCCL_OK(ncclGroupStart());
    if (p->rank == 0)
        for (int r = 0; r < W; r++) NCCL_OK(ncclRecv((char*)drecv.p + (size_t)r * slab, slab, ncclUint8, r, p->comm, p->stream));
    NCCL_OK(ncclSend(dsend.p, slab, ncclUint8, 0, p->comm, p->stream));
    NCCL_OK(ncclGroupEnd());
    WAIT_OK(wait_comm(p, deadline, "cattus_pool_records (payloads)"));
    if (p->rank != 0) return wait_stream(p, deadline, "cattus_pool_records (payloads)");
    std::vector<uint8_t> all(slab * W);
    HIP_OK(hipMemcpyAsync(all.data(), drecv.p, slab * W, hipMemcpyDeviceToHost, p->stream));
    WAIT_OK(wait_stream(p, deadline, "cattus_pool_records (payloads)"));
    // 3. sorted by (game, ply), as cattus_amd.dist.pool_records returns them
    std::vector<const uint8_t*> rows;
    rows.reserve(total);
    for (int r = 0; r < W; r++)
        for (uint64_t i = 0; i < counts[2 * r]; i++) rows.push_back(&all[(size_t)r * slab + i * row]);
    auto key = [&](const uint8_t* q) {
        uint32_t m[2];
        memcpy(m, q + record_bytes, 8);
        return ((uint64_t)m[0] << 32) | m[1];
    };
    std::stable_sort(rows.begin(), rows.end(), [&](const uint8_t* a, const uint8_t* b) { return key(a) < key(b); });
    uint8_t* ob = (uint8_t*)malloc(std::max<size_t>(1, total * record_bytes));
    uint32_t* om = (uint32_t*)malloc(std::max<size_t>(1, total * 12));
    if (!ob || !om) {
        free(ob), free(om);
        return fail(-4, "out of memory");
    }
    for (uint64_t i = 0; i < total; i++) {
        memcpy(ob + i * record_bytes, rows[i], record_bytes);
        memcpy(om + 3 * i, rows[i] + record_bytes, 12);
    }
    *all_bytes = ob, *all_meta = om;
    return 0;
}
