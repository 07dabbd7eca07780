// libcattus_pool.so: the C ABI of include/cattus_pool.h over RCCL.  Host code only (HIP runtime + RCCL calls, no kernel).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
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
        if (r__ != ncclSuccess) return fail(-3, "%s failed: %s", #expr, ncclGetErrorString(r__)); \
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
};

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
    ncclResult_t r = ncclCommInitRank(&p->comm, world, u, rank);
    if (r != ncclSuccess) {
        delete p;
        return fail(-3, "ncclCommInitRank failed: %s", ncclGetErrorString(r));
    }
    if (hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking) != hipSuccess) {
        ncclCommDestroy(p->comm);
        delete p;
        return fail(-3, "hipStreamCreate failed");
    }
    *out = p;
    return 0;
}

POOL_API void cattus_pool_destroy(cattus_pool* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    if (p->comm) ncclCommDestroy(p->comm);
    delete p;
}

POOL_API void cattus_pool_free(void* ptr) { free(ptr); }

POOL_API int cattus_pool_reduce_counters(cattus_pool* p, uint64_t* counters, uint32_t n) {
    if (!p || (!counters && n)) return fail(-1, "NULL argument");
    if (!n) return 0;
    HIP_OK(hipSetDevice(p->device));
    DevMem d;
    HIP_OK(hipMalloc(&d.p, (size_t)n * 8));
    HIP_OK(hipMemcpyAsync(d.p, counters, (size_t)n * 8, hipMemcpyHostToDevice, p->stream));
    NCCL_OK(ncclAllReduce(d.p, d.p, n, ncclUint64, ncclSum, p->comm, p->stream));
    HIP_OK(hipMemcpyAsync(counters, d.p, (size_t)n * 8, hipMemcpyDeviceToHost, p->stream));
    HIP_OK(hipStreamSynchronize(p->stream));
    return 0;
}

POOL_API int cattus_pool_records(cattus_pool* p, const uint8_t* bytes, const uint32_t* meta, uint64_t n_local, uint32_t record_bytes,
                                 uint8_t** all_bytes, uint32_t** all_meta, uint64_t* n_total) {
    if (!p || !all_bytes || !all_meta || !n_total || (n_local && (!bytes || !meta))) return fail(-1, "NULL argument");
    if (!record_bytes) return fail(-1, "record_bytes is 0");
    *all_bytes = nullptr, *all_meta = nullptr, *n_total = 0;
    HIP_OK(hipSetDevice(p->device));
    const int W = p->world;
    // 1. every rank's count (and record size, which must agree)
    DevMem dc;
    HIP_OK(hipMalloc(&dc.p, (size_t)W * 16));
    const uint64_t mine[2] = {n_local, record_bytes};
    HIP_OK(hipMemcpyAsync((char*)dc.p + (size_t)p->rank * 16, mine, 16, hipMemcpyHostToDevice, p->stream));
    NCCL_OK(ncclAllGather((char*)dc.p + (size_t)p->rank * 16, dc.p, 2, ncclUint64, p->comm, p->stream));
    std::vector<uint64_t> counts((size_t)W * 2);
    HIP_OK(hipMemcpyAsync(counts.data(), dc.p, (size_t)W * 16, hipMemcpyDeviceToHost, p->stream));
    HIP_OK(hipStreamSynchronize(p->stream));
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
    if (p->rank != 0) {
        HIP_OK(hipStreamSynchronize(p->stream));
        return 0;
    }
    std::vector<uint8_t> all(slab * W);
    HIP_OK(hipMemcpyAsync(all.data(), drecv.p, slab * W, hipMemcpyDeviceToHost, p->stream));
    HIP_OK(hipStreamSynchronize(p->stream));
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
