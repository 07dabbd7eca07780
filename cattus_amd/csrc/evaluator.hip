// Host side of libcattus_hip.so: the C ABI declared in include/cattus_hip.h.
//
// Replaces NNetwork::run_net + Model::{new,run} (reference: engine/src/net/mod.rs:41-72,
// engine/src/net/model.rs:61-218) and Batcher::apply (engine/src/util/batch.rs:49-177).
// There is deliberately no CPU fallback: without a usable HIP device every entry point fails.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <shared_mutex>
#include <string>
#include <thread>
#include <vector>

#include <dlfcn.h>

#include "../../include/cattus_hip.h"
#include "../../include/cattus_hip_diag.h"
#include "kernels.h"

// Kernel arguments in device memory: by default the HIP runtime keeps the kernel-argument ring in host memory and the
// command processor fetches every launch's arguments over the host link before the first wave starts (~1.6 us per
// launch, 8 % of a 41-launch bf16 forward pass; profiles/r02_experiments.txt).  HIP_FORCE_DEV_KERNARG=1 moves the ring
// into HBM, but the runtime reads it once, when it initialises, and the variable belongs to the HOST process: this
// library does not touch the environment (a library constructor calling setenv races with the host's threads and
// changes the runtime for every other HIP user).  bench.py, the tests, the Python package (opt-out: CATTUS_NO_ENV_DEFAULTS=1)
// and bin/*_self_player export it; INTEGRATION.md tells a Rust host to do the same.  cattus_hip_create reports the
// setting it found in cattus_hip_runtime_note().
static std::string g_runtime_note;

using namespace cattus;

#define CATTUS_API extern "C" __attribute__((visibility("default")))

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t err__ = (expr);                                                                      \
        if (err__ != hipSuccess)                                                                        \
            return fail(CATTUS_E_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(err__), __FILE__, __LINE__); \
    } while (0)

constexpr size_t HEADER_BYTES = 64;
constexpr float BN_EPS = 1e-5f;
constexpr uint32_t FC_HIDDEN = 128;

inline uint16_t f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // keep NaN a NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                           // round to nearest even
    return (uint16_t)(u >> 16);
}

// BatchNorm (eval) folded into the preceding bias-free conv:
//   scale = gamma / sqrtf(var + eps);  w' = w * scale;  b' = beta - mean * scale
// (gamma = 1, beta = 0 where the reference builds BatchNorm2d(affine=False): net_utils.py:14,30,68,78).
// This file is compiled with -ffp-contract=off: the products and the subtraction round separately.
struct Folded {
    std::vector<float> w;  // [taps][cout][cin]
    std::vector<float> b;  // [cout]
};

Folded fold_conv(const float* w, uint32_t cout, uint32_t cin, uint32_t taps, const float* gamma, const float* beta,
                 const float* mean, const float* var) {
    Folded f;
    f.w.resize((size_t)taps * cout * cin);
    f.b.resize(cout);
    for (uint32_t co = 0; co < cout; co++) {
        const float g = gamma ? gamma[co] : 1.0f;
        const float be = beta ? beta[co] : 0.0f;
        const float scale = g / sqrtf(var[co] + BN_EPS);
        f.b[co] = be - mean[co] * scale;
        for (uint32_t ci = 0; ci < cin; ci++)
            for (uint32_t t = 0; t < taps; t++)
                f.w[((size_t)t * cout + co) * cin + ci] = w[((size_t)co * cin + ci) * taps + t] * scale;
    }
    return f;
}

// One device allocation that a tower's hot buffers are carved from (the Winograd tower: the transformed weights of all layers and
// the activation buffers).  What a forward pass of chess 20x256 touches -- 168 MB of U, 50 MB of activations -- is most of the 256 MB
// Infinity Cache, which is indexed by physical address: 110 separately allocated 2 MB pages land on its sets unevenly, some sets
// overflow, and the weight stream then comes from HBM in part (the same binary ran a layer in 31 or in 34 us from one process to
// the next, and towers of <= 17 blocks always in 31).  One contiguous block covers the sets evenly.
struct DevArena {
    char* base = nullptr;
    size_t cap = 0, used = 0;
    ~DevArena() {
        if (base) (void)hipFree(base);
    }
    void* take(size_t bytes) {  // nullptr when it does not fit (or there is no arena): the caller allocates on its own
        const size_t at = (used + 4095) & ~(size_t)4095;
        if (!base || at + bytes > cap) return nullptr;
        used = at + bytes;
        return base + at;
    }
};

struct DevBuf {
    void* p = nullptr;
    bool owned = true;  // false: carved from a DevArena, which frees it
    ~DevBuf() {
        if (p && owned) (void)hipFree(p);
    }
    int alloc(size_t bytes, DevArena* arena = nullptr) {
        if (p && owned) (void)hipFree(p);
        p = nullptr, owned = true;
        if (arena && (p = arena->take(bytes ? bytes : 16))) {
            owned = false;
            return CATTUS_OK;
        }
        hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
        if (e != hipSuccess) return fail(CATTUS_E_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return CATTUS_OK;
    }
    int upload(const void* src, size_t bytes, DevArena* arena = nullptr) {
        int rc = alloc(bytes, arena);
        if (rc) return rc;
        HIP_TRY(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
        return CATTUS_OK;
    }
    template <typename T>
    T* as() const {
        return reinterpret_cast<T*>(p);
    }
};

struct PinnedBuf {
    void* p = nullptr;
    ~PinnedBuf() {
        if (p) (void)hipHostFree(p);
    }
    int alloc(size_t bytes) {
        hipError_t e = hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault);
        if (e != hipSuccess) return fail(CATTUS_E_NOMEM, "hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return CATTUS_OK;
    }
    template <typename T>
    T* as() const {
        return reinterpret_cast<T*>(p);
    }
};

// page-locked host ranges handed out by cattus_hip_host_alloc: start -> length.  Every evaluation probes
// this (three pointers per batch, from several threads), allocation is rare: readers share the lock.
std::shared_mutex g_pin_mu;
std::map<const char*, size_t> g_pinned;
// Optional ROCTx ranges around every batch (CATTUS_ROCTX=1): they show up in `rocprofv3 --marker-trace`.
// The library is looked up at run time so that nothing links against the profiler.
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char* on = getenv("CATTUS_ROCTX");
        if (!on || on[0] != '1') return;
        for (const char* name : {"librocprofiler-sdk-roctx.so", "libroctx64.so"}) {
            if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
                push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
                pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (push && pop) return;
                push = nullptr, pop = nullptr;
            }
        }
    }
};
const Roctx& roctx() {
    static const Roctx r;
    return r;
}
struct RoctxRange {
    bool on;
    RoctxRange(const char* what, uint32_t n) : on(roctx().push != nullptr) {
        if (on) {
            char buf[64];
            snprintf(buf, sizeof buf, "%s n=%u", what, n);
            roctx().push(buf);
        }
    }
    ~RoctxRange() {
        if (on) roctx().pop();
    }
};

bool is_pinned(const void* p) {
    std::shared_lock<std::shared_mutex> lk(g_pin_mu);
    auto it = g_pinned.upper_bound((const char*)p);
    if (it == g_pinned.begin()) return false;
    --it;
    return (const char*)p < it->first + it->second;
}

struct ConvLayer {
    DevBuf w, b;
    DevBuf wf;  // f16x2: the same weights in MFMA fragment order (kernels.h, CONV_W_FRAG)
    DevBuf wu, bw;  // f16x2, Winograd form: G g G^T as scaled (hi, lo) pairs in fragment order, and its [biases | inverse scales]
    uint32_t cin = 0;  // as laid out on the device (padded for the MFMA path)
};

struct ServerBatch {
    uint64_t seq = 0;
    uint32_t count = 0, collected = 0;
    bool sealed = false, running = false, done = false;
    int status = CATTUS_OK;
    std::string error;
    std::chrono::steady_clock::time_point t0;
    std::vector<uint64_t> planes;
    std::vector<float> policy, value;
    std::vector<uint8_t> taken;  // per slot: its ticket has been waited for
};

constexpr int NLANES = CATTUS_HIP_LANES;

struct Lane {
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;  // blocking-sync event: the host thread sleeps until the batch is back
    DevBuf d_planes, x0, a, t, y, hv, h1, d_policy, d_value;
    DevBuf d_legal_idx, d_legal_cnt, d_probs;  // legal-move softmax operands, allocated on first use
    uint32_t legal_stride = 0;
    PinnedBuf h_planes, h_policy, h_value;
    // the Winograd tower in one launch (tower_wino4_kernel): this lane's layer table, hand-off counters and error word, and the
    // page-locked word the error word is copied to behind every such launch
    DevBuf tower_layers, tower_ready;  // tower_ready: [error word, pad to 64 B | a counter per (layer, board group)] -- one memset per batch
    PinnedBuf h_tower_err;
    uint32_t tower_nlayers = 0;
    std::mutex mu;  // held while a batch uses the lane
    ~Lane() {
        if (done) (void)hipEventDestroy(done);
        if (stream) (void)hipStreamDestroy(stream);
    }
};

}  // namespace

struct cattus_eval {
    DevArena arena;  // first member: destroyed last, behind every buffer carved from it
    cattus_net_desc d{};
    cattus_eval_config cfg{};
    bool tuned = false;  // MFMA NHWC tower vs generic NCHW f32 tower
    // SimpleTwoHeadedModel (training/cattus_train/net_utils.py:92-121; blob with filters == 0): planes -> f32 tensor -> two dense
    // layers + ReLU -> a dense tanh value head and a dense policy head, all in f32 whatever cfg.dtype says (the net is tiny)
    bool simple = false;
    DevBuf d1w, d1b, d2w, d2b, svw, svb, spw, spb;
    bool wait_spin = true;  // host wait for a batch: spinning hipStreamSynchronize, or a blocking event
    Act act = Act::F32;
    uint32_t hw = 0, bpad = 0, cpad0 = 0;
    uint32_t slots = 64;  // pixel slots per board of the tuned tower (kernels.h: tower_slots)
    uint32_t fpad = 0;    // filters as laid out on the device: rounded up to 64 on the tuned path (zero channels)
    // bf16 networks with <= 64 filters: the whole tower in one launch, activations resident in LDS
    // (tower64_lds_kernel; CATTUS_TOWER64=0 selects the per-layer launches, for A/B runs and the equality test)
    bool tower64 = false;
    bool resident_tower = true;  // diagnostic switch CATTUS_TOWER64=0: per-layer launches instead (A/B runs, the equality tests)
    DevBuf t64_layers;
    // f16x2 networks with <= 64 filters: the same, in split precision (tower64_split_kernel; weights from the register ring)
    bool tower64s = false;
    DevBuf t64s_bias;
    bool t64s_fuse_heads = true;  // CATTUS_T64S_HEADS=0: the head convs as their own launch on the tower's f32 rows (A/B, the equality test)
    int t64s_depth = 0;           // CATTUS_T64S_SHAPE=1|2|9: workgroup shape of the resident split tower (kernels.h; 0: by grid size)
    bool pack_separately = false;  // CATTUS_FUSED_STEM=0: plane pack as its own launch in front of the stem (A/B, tests)
    int t64_force_ch = 0;          // CATTUS_T64_CH=2|4: workgroup shape of the resident tower (A/B runs, the row-split test)
    // f16x2, 8x8 boards, filters a multiple of 128: every layer behind the stem in Winograd F(2x2, 3x3) form (kernels_wino.hip).
    // Chosen once, when the evaluator is created -- never per batch, so that a leaf's result does not depend on the batch it came
    // in: for max_batch > 128 (up to there the direct kernels' small tiles win or tie -- whole step of chess 20x256, direct | Winograd: 0.59 |
    // 0.89 ms at 64 leaves, 0.87 | 0.89 at 96, 0.97 | 0.90 at 128 full and ~0.87 | 0.89 at the ~93 a 128-leaf self-play batch holds; from 129 on
    // the direct form needs a second 128-row tile: 1.45 | 0.93 at 160, 1.53 | 1.01 at 192, 1.77 | 1.25 at 256, scripts/by_batch_forms.py);
    // cattus_eval_config.tower_form forbids / forces it.
    bool winograd = false;
    // which Winograd kernel: the 4-frequencies x 2x2-blocks one (kernels_wino4.hip) wherever it covers the shape, else the
    // 16-frequencies one (kernels_wino.hip); same bits; diagnostic switch CATTUS_WINO_KERNEL=k16|k4
    bool wino_k4 = true;
    bool wino_k8 = false;  // CATTUS_WINO_KERNEL=k8: the eight-wave kernel (kernels_wino8.hip), per layer
    // The Winograd tower as ONE launch (tower_wino4_kernel), one workgroup per CU, several tiles per workgroup and layer where a layer has
    // more tiles than the device has CUs (diagnostic switch
    // CATTUS_WINO_PERSIST=0: per-layer launches; CATTUS_WINO_SPIN=<polls>: the budget of a hand-off wait).  persist_ok falls when
    // a launch reported a wait that gave up: the batch is run again on the per-layer launches, and so is every later one.
    bool wino_persist = true;
    std::atomic<bool> persist_ok{true};
    uint32_t persist_spin = 1u << 18;
    uint32_t cus = 0;
    bool wino_inplace = true;      // CATTUS_WINO_INPLACE=0: a third activation buffer for the blocks' outputs (A/B runs)
    bool split_wfrag = true;       // CATTUS_SPLIT_W=0: f16x2 weights through the LDS ring (conv3x3_split_kernel) instead of the register ring
    // tile-forcing switches (CATTUS_CONV_CB, CATTUS_CONV_PBW: A/B runs, the tile-equality tests) and the f16 towers' saturation
    // counter: this evaluator's own -- a second evaluator in the process (model1 vs model2) neither re-tiles nor shares them
    ConvOpts conv_opts;
    DevBuf d_saturated;
    bool t64_layer_steps = true;   // CATTUS_T64_LS=0: three barriers per layer in the one-board resident tower
    int device = 0;

    ConvLayer stem;
    std::vector<std::unique_ptr<ConvLayer>> c1, c2;
    // heads: generic path keeps f32 [k][n]-transposed FC weights for the SIMT kernels; the tuned path
    // keeps K-contiguous, zero-padded matrices in the tower element type for the MFMA head GEMMs
    DevBuf head_w, head_b, w1t, b1, w2, b2, wpt, bp;
    uint32_t kvp = 0, kpp = 0;  // padded K of the value / policy FC (tuned path)

    // Two lanes = two independent sets of activation buffers, each with its own stream, so that two
    // host threads can have a batch in flight each: one lane's transfers, launch latency and completion
    // wake-up hide behind the other lane's kernels.  The weights are shared.
    Lane lanes[NLANES];
    std::atomic<unsigned> lane_rr{0};

    std::mutex stat_mu;
    cattus_stats stats{};
    uint64_t stats_tower_fallbacks = 0;  // batches the one-launch tower gave up on (run again on the per-layer launches)

    // leaf server
    std::mutex srv_mu;
    std::condition_variable srv_cv, done_cv;
    std::deque<std::unique_ptr<ServerBatch>> batches;  // oldest first; back() is collecting unless sealed
    uint64_t next_seq = 1;
    bool flush_req = false, stop = false;
    std::thread servers[NLANES];  // one per lane: two sealed batches can be on the device together

    ~cattus_eval() {
        {
            std::lock_guard<std::mutex> lk(srv_mu);
            stop = true;
        }
        srv_cv.notify_all();
        done_cv.notify_all();
        for (auto& t : servers)
            if (t.joinable()) t.join();
    }
};

namespace {

size_t blob_floats(const cattus_net_desc& d) {
    const size_t hw = (size_t)d.board * d.board, F = d.filters;
    if (F == 0) {  // SimpleTwoHeadedModel: two K x K dense layers, a 1 x K and an M x K head
        const size_t K = (size_t)d.planes * hw;
        return 2 * (K * K + K) + K + 1 + (size_t)d.moves * K + d.moves;
    }
    size_t n = F * d.planes * 9 + 4 * F;
    n += (size_t)d.blocks * (2 * F * F * 9 + 6 * F);
    n += (size_t)d.vhc * F + 2 * (size_t)d.vhc + (size_t)FC_HIDDEN * d.vhc * hw + FC_HIDDEN + FC_HIDDEN + 1;
    n += (size_t)d.phc * F + 2 * (size_t)d.phc + (size_t)d.moves * d.phc * hw + d.moves;
    return n;
}

// The Winograd kernel this evaluator would run a cin -> cout layer on covers that shape.
bool wino_shape_ok(const cattus_eval* e, uint32_t cin, uint32_t cout) {
    return e->wino_k4 ? wino4_supported(e->bpad, cin, cout, e->d.board) : wino_supported(e->bpad, cin, cout, e->d.board);
}

// Launches of the persistent tower never overlap on a device: each waits for the one before it (an event chain per device, process-wide:
// two evaluators -- model1 and model2 of a self-play round -- or the two lanes of one would otherwise share the CUs, neither launch
// would have all its workgroups resident, and each would wait for hand-offs from workgroups the other keeps out).  Other work of the
// lanes (copies, stem, heads) overlaps as before.
std::mutex g_chain_mu;
hipEvent_t g_chain[64] = {};
int chain_persistent_launch(int device, hipStream_t st, const std::function<void()>& launch) {
    std::lock_guard<std::mutex> lk(g_chain_mu);
    hipEvent_t& ev = g_chain[device & 63];
    if (!ev) {
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    } else {
        HIP_TRY(hipStreamWaitEvent(st, ev, 0));
    }
    launch();
    HIP_TRY(hipEventRecord(ev, st));
    return CATTUS_OK;
}

// Upload one folded 3x3 layer in the layout of the selected tower.  On the tuned path output channels are
// padded to `cout_pad` and input channels to the device layout of the producing layer (`cin_pad`, a multiple of
// one 128-byte row) with zero weights and zero bias: a padded channel computes relu(0) = 0, and as an input it
// adds fmaf(0, x, acc) = acc terms only, so the f32 chains of the real channels are bit for bit unchanged.
int upload_conv(cattus_eval* e, ConvLayer& L, const Folded& f, uint32_t cout, uint32_t cin, uint32_t cout_pad, uint32_t cin_pad) {
    int rc;
    if (!e->tuned) {
        L.cin = cin;
        if ((rc = L.b.upload(f.b.data(), cout * sizeof(float)))) return rc;
        return L.w.upload(f.w.data(), f.w.size() * sizeof(float));
    }
    L.cin = cin_pad;
    if (e->act == Act::F16S) {
        // Split precision (kernels.hip, K1s): a weight is the pair hi = f16(w'), lo = f16(w' - hi) of w' = w * 2^s, with
        // s chosen per output channel so that the channel's largest |w'| lies in [2^10, 2^11): the lo halves of all but
        // the channel's tiniest weights are then normal f16 numbers (22 significant bits per weight), nothing comes near
        // the f16 range limit, and 2^-s -- applied to the f32 accumulator in the epilogue -- undoes the scale exactly.
        // Rows are [hi of 32 input channels | lo of the same 32] per 128 bytes; the bias buffer is [cout_pad biases |
        // cout_pad inverse scales].
        std::vector<float> b((size_t)2 * cout_pad, 0.0f);
        memcpy(b.data(), f.b.data(), cout * sizeof(float));
        std::vector<_Float16> w((size_t)9 * cout_pad * 2 * cin_pad, (_Float16)0.0f);
        for (uint32_t co = 0; co < cout_pad; co++) {
            float m = 0.0f;
            if (co < cout)
                for (uint32_t t = 0; t < 9; t++)
                    for (uint32_t ci = 0; ci < cin; ci++) m = std::max(m, fabsf(f.w[((size_t)t * cout + co) * cin + ci]));
            int sh = 0;
            if (m > 0.0f && std::isfinite(m)) sh = std::min(100, std::max(-100, 10 - ilogbf(m)));
            b[cout_pad + co] = ldexpf(1.0f, -sh);
            if (co >= cout) continue;
            for (uint32_t t = 0; t < 9; t++)
                for (uint32_t ci = 0; ci < cin; ci++) {
                    const float ws = ldexpf(f.w[((size_t)t * cout + co) * cin + ci], sh);
                    const _Float16 hi = (_Float16)ws;
                    const _Float16 lo = (_Float16)(ws - (float)hi);
                    _Float16* row = &w[((size_t)t * cout_pad + co) * 2 * cin_pad + (size_t)(ci >> 5) * 64 + (ci & 31)];
                    row[0] = hi, row[32] = lo;
                }
        }
        if ((rc = L.b.upload(b.data(), b.size() * sizeof(float)))) return rc;
        if (e->winograd && &L != &e->stem && wino_shape_ok(e, cin_pad, cout_pad)) {
            // Winograd F(2x2, 3x3) form: U = G g G^T per (cout, cin) in float64, one power-of-two scale per output channel over
            // all 16 frequencies (largest |U 2^s| in [2^10, 2^11)), split into (hi, lo), in the kernel's fragment order
            static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
            std::vector<double> U((size_t)cout * cin * 16);
            std::vector<float> bwv((size_t)2 * cout_pad, 0.0f);
            memcpy(bwv.data(), f.b.data(), cout * sizeof(float));
            std::vector<_Float16> wu((size_t)16 * cout_pad * cin_pad * 2 + (size_t)WINO_RING_STAGES * 1024, (_Float16)0.0f);  // + the ring's overrun
            for (uint32_t co = 0; co < cout_pad; co++) {
                double m = 0.0;
                if (co < cout)
                    for (uint32_t ci = 0; ci < cin; ci++) {
                        double g[3][3], t[4][3];
                        for (int ky = 0; ky < 3; ky++)
                            for (int kx = 0; kx < 3; kx++) g[ky][kx] = f.w[((size_t)(ky * 3 + kx) * cout + co) * cin + ci];
                        for (int i = 0; i < 4; i++)
                            for (int kx = 0; kx < 3; kx++) t[i][kx] = G[i][0] * g[0][kx] + G[i][1] * g[1][kx] + G[i][2] * g[2][kx];
                        for (int i = 0; i < 4; i++)
                            for (int l = 0; l < 4; l++) {
                                const double u = t[i][0] * G[l][0] + t[i][1] * G[l][1] + t[i][2] * G[l][2];
                                U[((size_t)co * cin + ci) * 16 + i * 4 + l] = u;
                                m = std::max(m, fabs(u));
                            }
                    }
                int sh = 0;
                if (m > 0.0 && std::isfinite(m)) sh = std::min(100, std::max(-100, 10 - ilogb(m)));
                bwv[cout_pad + co] = ldexpf(1.0f, -sh);
                if (co >= cout) continue;
                for (uint32_t ci = 0; ci < cin; ci++)
                    for (uint32_t fq = 0; fq < 16; fq++) {
                        const float us = (float)ldexp(U[((size_t)co * cin + ci) * 16 + fq], sh);
                        const _Float16 hi = (_Float16)us;
                        wu[wino_frag_index(fq, co, ci, 0, cin_pad)] = hi;
                        wu[wino_frag_index(fq, co, ci, 1, cin_pad)] = (_Float16)(us - (float)hi);
                    }
            }
            if ((rc = L.bw.upload(bwv.data(), bwv.size() * sizeof(float)))) return rc;
            if ((rc = L.wu.upload(wu.data(), wu.size() * 2, &e->arena))) return rc;
        }
        if (e->split_wfrag) {  // the register-ring kernel's layout: a permutation of the rows above
            std::vector<_Float16> wf(w.size());
            for (uint32_t t = 0; t < 9; t++)
                for (uint32_t co = 0; co < cout_pad; co++) {
                    const _Float16* row = &w[((size_t)t * cout_pad + co) * 2 * cin_pad];
                    for (uint32_t ci = 0; ci < cin_pad; ci++) {
                        wf[split_frag_index(t, co, ci, 0, cin_pad)] = row[(size_t)(ci >> 5) * 64 + (ci & 31)];
                        wf[split_frag_index(t, co, ci, 1, cin_pad)] = row[(size_t)(ci >> 5) * 64 + (ci & 31) + 32];
                    }
                }
            return L.wf.upload(wf.data(), wf.size() * 2);
        }
        return L.w.upload(w.data(), w.size() * 2);
    }
    if (e->act == Act::F16) {
        // Single-term f16: w' = f16(w * 2^s), s per output channel as for the split tower (largest |w'| in [2^10, 2^11): no
        // weight of the channel becomes a subnormal unless it is 2^-24 of the largest), rows [9][cout_pad][cin_pad] as in the
        // bf16 tower; the bias buffer is [cout_pad biases | cout_pad inverse scales].
        std::vector<float> b((size_t)2 * cout_pad, 0.0f);
        memcpy(b.data(), f.b.data(), cout * sizeof(float));
        std::vector<_Float16> w((size_t)9 * cout_pad * cin_pad, (_Float16)0.0f);
        for (uint32_t co = 0; co < cout_pad; co++) {
            float m = 0.0f;
            if (co < cout)
                for (uint32_t t = 0; t < 9; t++)
                    for (uint32_t ci = 0; ci < cin; ci++) m = std::max(m, fabsf(f.w[((size_t)t * cout + co) * cin + ci]));
            int sh = 0;
            if (m > 0.0f && std::isfinite(m)) sh = std::min(100, std::max(-100, 10 - ilogbf(m)));
            b[cout_pad + co] = ldexpf(1.0f, -sh);
            if (co >= cout) continue;
            for (uint32_t t = 0; t < 9; t++)
                for (uint32_t ci = 0; ci < cin; ci++)
                    w[((size_t)t * cout_pad + co) * cin_pad + ci] = (_Float16)ldexpf(f.w[((size_t)t * cout + co) * cin + ci], sh);
        }
        if ((rc = L.b.upload(b.data(), b.size() * sizeof(float)))) return rc;
        return L.w.upload(w.data(), w.size() * 2);
    }
    std::vector<float> b(cout_pad, 0.0f);
    memcpy(b.data(), f.b.data(), cout * sizeof(float));
    if ((rc = L.b.upload(b.data(), b.size() * sizeof(float)))) return rc;
    if (e->act == Act::BF16) {
        std::vector<uint16_t> w((size_t)9 * cout_pad * cin_pad, 0);
        for (uint32_t t = 0; t < 9; t++)
            for (uint32_t co = 0; co < cout; co++)
                for (uint32_t ci = 0; ci < cin; ci++)
                    w[((size_t)t * cout_pad + co) * cin_pad + ci] = f32_to_bf16(f.w[((size_t)t * cout + co) * cin + ci]);
        return L.w.upload(w.data(), w.size() * 2);
    }
    std::vector<float> w((size_t)9 * cout_pad * cin_pad, 0.0f);
    for (uint32_t t = 0; t < 9; t++)
        for (uint32_t co = 0; co < cout; co++)
            memcpy(&w[((size_t)t * cout_pad + co) * cin_pad], &f.w[((size_t)t * cout + co) * cin], cin * sizeof(float));
    return L.w.upload(w.data(), w.size() * 4);
}

// SimpleTwoHeadedModel: weights transposed to [k][n] (coalesced along n in the dense kernel), buffers for the planes tensor
// and the two hidden layers.
int build_simple(cattus_eval* e, const float* p) {
    const cattus_net_desc& d = e->d;
    const size_t K = (size_t)d.planes * e->hw, M = d.moves, B = e->cfg.max_batch;
    auto upload_t = [&](DevBuf& buf, const float* w, size_t n_out) -> int {
        std::vector<float> t(K * n_out);
        for (size_t n = 0; n < n_out; n++)
            for (size_t k = 0; k < K; k++) t[k * n_out + n] = w[n * K + k];
        return buf.upload(t.data(), t.size() * 4);
    };
    int rc;
    if ((rc = upload_t(e->d1w, p, K))) return rc;
    p += K * K;
    if ((rc = e->d1b.upload(p, K * 4))) return rc;
    p += K;
    if ((rc = upload_t(e->d2w, p, K))) return rc;
    p += K * K;
    if ((rc = e->d2b.upload(p, K * 4))) return rc;
    p += K;
    if ((rc = upload_t(e->svw, p, 1))) return rc;
    p += K;
    if ((rc = e->svb.upload(p, 4))) return rc;
    p += 1;
    if ((rc = upload_t(e->spw, p, M))) return rc;
    p += M * K;
    if ((rc = e->spb.upload(p, M * 4))) return rc;
    for (Lane& L : e->lanes) {
        HIP_TRY(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&L.done, hipEventBlockingSync | hipEventDisableTiming));
        if ((rc = L.d_planes.alloc(B * d.planes * e->cfg.plane_words * 8))) return rc;
        if ((rc = L.x0.alloc(B * K * 4))) return rc;
        if ((rc = L.a.alloc(B * K * 4))) return rc;
        if ((rc = L.t.alloc(B * K * 4))) return rc;
        if ((rc = L.d_policy.alloc(B * M * 4))) return rc;
        if ((rc = L.d_value.alloc(B * 4))) return rc;
        if ((rc = L.h_planes.alloc(B * d.planes * e->cfg.plane_words * 8))) return rc;
        if ((rc = L.h_policy.alloc(B * M * 4))) return rc;
        if ((rc = L.h_value.alloc(B * 4))) return rc;
    }
    return CATTUS_OK;
}

int build(cattus_eval* e, const float* p) {
    const cattus_net_desc& d = e->d;
    if (e->simple) return build_simple(e, p);
    const uint32_t F = d.filters, hw = e->hw, FP = e->fpad;
    auto take = [&](size_t n) {
        const float* r = p;
        p += n;
        return r;
    };
    int rc;
    {
        const float* w = take((size_t)F * d.planes * 9);
        const float *g = take(F), *be = take(F), *mu = take(F), *var = take(F);
        const uint32_t kc = e->tuned ? (uint32_t)act_kc(e->act) : 1;
        if ((rc = upload_conv(e, e->stem, fold_conv(w, F, d.planes, 9, g, be, mu, var), F, d.planes, FP, (d.planes + kc - 1) / kc * kc))) return rc;
        e->cpad0 = e->stem.cin;
    }
    for (uint32_t i = 0; i < d.blocks; i++) {
        e->c1.emplace_back(new ConvLayer);
        e->c2.emplace_back(new ConvLayer);
        const float* w1 = take((size_t)F * F * 9);
        const float *mu1 = take(F), *var1 = take(F);
        if ((rc = upload_conv(e, *e->c1.back(), fold_conv(w1, F, F, 9, nullptr, nullptr, mu1, var1), F, F, FP, FP))) return rc;
        const float* w2 = take((size_t)F * F * 9);
        const float *g2 = take(F), *be2 = take(F), *mu2 = take(F), *var2 = take(F);
        if ((rc = upload_conv(e, *e->c2.back(), fold_conv(w2, F, F, 9, g2, be2, mu2, var2), F, F, FP, FP))) return rc;
    }
    // heads: value rows first, then policy rows, in one [vhc+phc][F] 1x1 conv
    std::vector<float> hw_w((size_t)(d.vhc + d.phc) * F), hw_b(std::max(32u, d.vhc + d.phc), 0.0f);  // bias padded to one 32-row tile
    const float* vw = take((size_t)d.vhc * F);
    const float *vmu = take(d.vhc), *vvar = take(d.vhc);
    Folded fv = fold_conv(vw, d.vhc, F, 1, nullptr, nullptr, vmu, vvar);
    const float* fc1_w = take((size_t)FC_HIDDEN * d.vhc * hw);
    const float* fc1_b = take(FC_HIDDEN);
    const float* fc2_w = take(FC_HIDDEN);
    const float* fc2_b = take(1);
    const float* pw = take((size_t)d.phc * F);
    const float *pmu = take(d.phc), *pvar = take(d.phc);
    Folded fp = fold_conv(pw, d.phc, F, 1, nullptr, nullptr, pmu, pvar);
    const float* pfc_w = take((size_t)d.moves * d.phc * hw);
    const float* pfc_b = take(d.moves);
    memcpy(hw_w.data(), fv.w.data(), fv.w.size() * 4);
    memcpy(hw_w.data() + fv.w.size(), fp.w.data(), fp.w.size() * 4);
    memcpy(hw_b.data(), fv.b.data(), fv.b.size() * 4);
    memcpy(hw_b.data() + fv.b.size(), fp.b.data(), fp.b.size() * 4);
    const uint32_t kv = d.vhc * hw, kp = d.phc * hw;
    if ((rc = e->head_b.upload(hw_b.data(), hw_b.size() * 4))) return rc;
    if ((rc = e->b1.upload(fc1_b, FC_HIDDEN * 4))) return rc;
    if ((rc = e->w2.upload(fc2_w, FC_HIDDEN * 4))) return rc;
    if ((rc = e->b2.upload(fc2_b, 4))) return rc;
    if ((rc = e->bp.upload(pfc_b, d.moves * 4))) return rc;
    if (e->tuned) {
        // K-contiguous matrices, K padded to 16 with zeros (zero terms do not change an fmaf chain)
        const uint32_t ocn = d.vhc + d.phc;
        e->kvp = (kv + 15) / 16 * 16;
        e->kpp = (kp + 15) / 16 * 16;
        const uint32_t m32 = (d.moves + 31) / 32 * 32;
        std::vector<float> cw((size_t)32 * FP, 0.0f), w1((size_t)FC_HIDDEN * e->kvp, 0.0f), wp((size_t)m32 * e->kpp, 0.0f);
        for (uint32_t oc = 0; oc < ocn; oc++) memcpy(&cw[(size_t)oc * FP], &hw_w[(size_t)oc * F], (size_t)F * 4);
        for (uint32_t j = 0; j < FC_HIDDEN; j++) memcpy(&w1[(size_t)j * e->kvp], &fc1_w[(size_t)j * kv], kv * 4);
        for (uint32_t m = 0; m < d.moves; m++) memcpy(&wp[(size_t)m * e->kpp], &pfc_w[(size_t)m * kp], kp * 4);
        const Act hact = head_act(e->act);  // the split tower's heads run in exact f32 on the last layer's f32 rows
        auto upload_t = [&](DevBuf& buf, const std::vector<float>& v) -> int {
            if (hact == Act::F32) return buf.upload(v.data(), v.size() * 4);
            std::vector<uint16_t> hb(v.size());
            for (size_t i = 0; i < v.size(); i++) hb[i] = f32_to_bf16(v[i]);
            return buf.upload(hb.data(), hb.size() * 2);
        };
        // the FC weights go up in MFMA fragment order (kernels.h, HeadsMfma): a wave's operand of one k-step is one KiB
        auto frag_order = [&](const std::vector<float>& v, uint32_t rows, uint32_t K) {
            const uint32_t kstep = hact == Act::F32 ? 8 : 16, half = kstep / 2;
            std::vector<float> o(v.size());
            for (uint32_t row = 0; row < rows; row++)
                for (uint32_t k = 0; k < K; k++)
                    o[((((size_t)(row >> 5) * (K / kstep) + k / kstep) * 2 + (k % kstep) / half) * 32 + (row & 31)) * half + k % half] =
                        v[(size_t)row * K + k];
            return o;
        };
        if ((rc = upload_t(e->head_w, cw))) return rc;
        if ((rc = upload_t(e->w1t, frag_order(w1, FC_HIDDEN, e->kvp)))) return rc;
        if ((rc = upload_t(e->wpt, frag_order(wp, m32, e->kpp)))) return rc;
    } else {
        std::vector<float> w1t((size_t)kv * FC_HIDDEN), wpt((size_t)kp * d.moves);
        for (uint32_t j = 0; j < FC_HIDDEN; j++)
            for (uint32_t k = 0; k < kv; k++) w1t[(size_t)k * FC_HIDDEN + j] = fc1_w[(size_t)j * kv + k];
        for (uint32_t m = 0; m < d.moves; m++)
            for (uint32_t k = 0; k < kp; k++) wpt[(size_t)k * d.moves + m] = pfc_w[(size_t)m * kp + k];
        if ((rc = e->head_w.upload(hw_w.data(), hw_w.size() * 4))) return rc;
        if ((rc = e->w1t.upload(w1t.data(), w1t.size() * 4))) return rc;
        if ((rc = e->wpt.upload(wpt.data(), wpt.size() * 4))) return rc;
        e->kvp = kv, e->kpp = kp;
    }

    if (e->tuned && e->act == Act::BF16 && FP == 64 && e->cpad0 == 64) {
        e->tower64 = e->resident_tower;
        std::vector<Tower64Layer> tl;
        tl.push_back(Tower64Layer{e->stem.w.p, e->stem.b.as<float>(), 0, 0});
        for (uint32_t i = 0; i < d.blocks; i++) {
            tl.push_back(Tower64Layer{e->c1[i]->w.p, e->c1[i]->b.as<float>(), 0, 0});
            tl.push_back(Tower64Layer{e->c2[i]->w.p, e->c2[i]->b.as<float>(), 1, 0});
        }
        if ((rc = e->t64_layers.upload(tl.data(), tl.size() * sizeof(Tower64Layer)))) return rc;
    }

    if (e->tuned && e->act == Act::F16S && FP == 64 && e->cpad0 == 32 && e->split_wfrag && 1 + 2 * d.blocks <= (uint32_t)T64S_MAX_LAYERS) {
        e->tower64s = e->resident_tower;
        std::vector<Tower64SplitLayer> tl;
        tl.push_back(Tower64SplitLayer{e->stem.wf.p, e->stem.b.as<float>(), 0, 1});
        for (uint32_t i = 0; i < d.blocks; i++) {
            tl.push_back(Tower64SplitLayer{e->c1[i]->wf.p, e->c1[i]->b.as<float>(), 0, 2});
            tl.push_back(Tower64SplitLayer{e->c2[i]->wf.p, e->c2[i]->b.as<float>(), 1, 2});
        }
        if ((rc = e->t64_layers.upload(tl.data(), tl.size() * sizeof(Tower64SplitLayer)))) return rc;
        // every layer's [64 biases | 64 inverse scales] in one table (the kernel copies it to LDS with independent loads)
        if ((rc = e->t64s_bias.alloc(tl.size() * 512))) return rc;
        for (size_t l = 0; l < tl.size(); l++)
            HIP_TRY(hipMemcpy((char*)e->t64s_bias.p + l * 512, tl[l].bias, 512, hipMemcpyDeviceToDevice));
    }

    // activations
    const size_t bp_ = e->bpad, B = e->cfg.max_batch;
    // bytes per channel of the tower buffers; the f16 towers' last layer writes f32 rows into one of them
    const size_t esz = e->tuned ? (act_f16_family(e->act) ? 4 : (size_t)act_bytes(e->act)) : 4;
    const size_t hesz = e->tuned ? (size_t)act_bytes(head_act(e->act)) : 4;  // element of the head activations
    const size_t slots = e->tuned ? e->slots : hw;
    const size_t FA = e->tuned ? FP : F;  // channels of the tower buffers
    for (Lane& L : e->lanes) {
        HIP_TRY(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&L.done, hipEventBlockingSync | hipEventDisableTiming));
        if ((rc = L.d_planes.alloc(B * d.planes * e->cfg.plane_words * 8))) return rc;
        if ((rc = L.x0.alloc(bp_ * slots * e->cpad0 * esz))) return rc;
        if ((rc = L.a.alloc(bp_ * slots * FA * esz, &e->arena))) return rc;
        if ((rc = L.t.alloc(bp_ * slots * FA * esz, &e->arena))) return rc;
        if ((rc = L.y.alloc(bp_ * slots * FA * esz, e->wino_inplace ? nullptr : &e->arena))) return rc;  // in place: the Winograd tower does not touch it
        const size_t hv_bytes = (size_t)(e->tuned ? (bp_ + 31) / 32 * 32 : bp_) * (e->kvp + e->kpp) * hesz;  // tuned: whole 32-leaf tiles
        if ((rc = L.hv.alloc(hv_bytes))) return rc;
        HIP_TRY(hipMemset(L.hv.p, 0, hv_bytes));  // pad columns (and leaves never written) must read as zero
        if ((rc = L.h1.alloc(bp_ * FC_HIDDEN * 4))) return rc;
        if ((rc = L.d_policy.alloc(B * d.moves * 4))) return rc;
        if ((rc = L.d_value.alloc(B * 4))) return rc;
        if ((rc = L.h_planes.alloc(B * d.planes * e->cfg.plane_words * 8))) return rc;
        if ((rc = L.h_policy.alloc(B * d.moves * 4))) return rc;
        if ((rc = L.h_value.alloc(B * 4))) return rc;
        if (e->tuned && e->act == Act::F16S && e->wino_k4 && e->wino_persist && e->wino_inplace && d.blocks > 0 && e->c1[0]->wu.p) {
            // the lane's layer table: conv1 of a block a -> t, conv2 t -> a over its own skip rows (the per-layer launches' in-place plan)
            std::vector<Wino4TowerLayer> tl;
            for (uint32_t i = 0; i < d.blocks; i++) {
                tl.push_back(Wino4TowerLayer{L.a.as<float>(), e->c1[i]->wu.p, e->c1[i]->bw.as<float>(), nullptr, L.t.as<float>()});
                tl.push_back(Wino4TowerLayer{L.t.as<float>(), e->c2[i]->wu.p, e->c2[i]->bw.as<float>(), L.a.as<float>(), L.a.as<float>()});
            }
            L.tower_nlayers = (uint32_t)tl.size();
            if ((rc = L.tower_layers.upload(tl.data(), tl.size() * sizeof(Wino4TowerLayer)))) return rc;
            if ((rc = L.tower_ready.alloc(64 + (size_t)tl.size() * (bp_ / 4) * 4))) return rc;
            if ((rc = L.h_tower_err.alloc(4))) return rc;
            *L.h_tower_err.as<unsigned>() = 0;
        }
    }
    return CATTUS_OK;
}

struct TowerTimer {
    std::vector<hipEvent_t> ev;  // pairs
    size_t used = 0;
};

// Enqueue the whole forward for n leaves whose planes are at d_planes; logits/values go to
// d_policy/d_value.  With `tt`, a HIP event pair brackets every tower conv launch.
int enqueue_forward(cattus_eval* e, Lane& L, const uint64_t* d_planes, uint32_t n, float* d_policy, float* d_value,
                    hipStream_t st, TowerTimer* tt = nullptr) {
    const cattus_net_desc& d = e->d;
    const uint32_t F = d.filters, hw = e->hw, S = d.board, w64 = e->cfg.plane_words;
    // timing pass: each tower launch gets its own (start, stop) event pair stamped by the kernel itself
    auto ev = [&](bool stop) -> hipEvent_t {
        if (!tt || tt->used >= tt->ev.size()) return nullptr;
        (void)stop;
        return tt->ev[tt->used++];
    };
    void *a = L.a.p, *t = L.t.p, *y = L.y.p;
    uint32_t nb = n;
    if (e->simple) {
        // SimpleTwoHeadedModel.forward (net_utils.py:112-121): flatten -> dense + ReLU -> dense + ReLU -> value: dense + tanh,
        // policy: dense + the non-finite scrub of net/mod.rs:56-61
        const uint32_t K = d.planes * hw;
        launch_planes_to_tensor_nchw(d_planes, n, d.planes, w64, S, n, L.x0.as<float>(), st);
        launch_dense(L.x0.as<float>(), K, e->d1w.as<float>(), e->d1b.as<float>(), n, K, K, (float*)a, 1, st);
        launch_dense((float*)a, K, e->d2w.as<float>(), e->d2b.as<float>(), n, K, K, (float*)t, 1, st);
        launch_dense((float*)t, K, e->svw.as<float>(), e->svb.as<float>(), n, K, 1, d_value, 2, st);
        launch_dense((float*)t, K, e->spw.as<float>(), e->spb.as<float>(), n, K, d.moves, d_policy, 0, st);
        hipError_t serr = hipGetLastError();
        if (serr != hipSuccess) return fail(CATTUS_E_DEVICE, "kernel launch failed: %s", hipGetErrorString(serr));
        return CATTUS_OK;
    }
    if (e->tuned) {
        const uint32_t bpw = ROWS_PER_WG / e->slots;  // boards per workgroup of the conv kernel
        const uint32_t FP = e->fpad;
        nb = (n + bpw - 1) / bpw * bpw;
        if (e->tower64) {
            Tower64Args ta{};
            ta.planes = d_planes, ta.layers = e->t64_layers.as<Tower64Layer>(), ta.out = nullptr;
            ta.n = n, ta.C = d.planes, ta.w64 = w64, ta.S = S, ta.nlayers = 1 + 2 * d.blocks;
            ta.head_w = e->head_w.p, ta.head_b = e->head_b.as<float>(), ta.hv = L.hv.p;
            ta.hv_pol = (uint32_t)((e->bpad + 31) / 32 * 32) * e->kvp, ta.kvp = e->kvp, ta.kpp = e->kpp, ta.vhc = d.vhc, ta.ocn = d.vhc + d.phc;
            const uint32_t rows = nb * e->slots;
            // 128-row workgroups; for 64-slot boards one board per workgroup while that leaves no CU with two of them
            hipEvent_t s0 = ev(false), s1 = ev(true);
            int ch = e->slots == 64 && rows / 64 <= 256 ? 4 : 2;
            if (e->t64_force_ch) ch = e->t64_force_ch == 4 && e->slots == 64 ? 4 : 2;  // A/B runs, the row-split test
            launch_tower64(ta, rows, ch, e->t64_layer_steps, st, s0, s1);
        } else if (e->tower64s) {
            // the whole split-precision tower in one launch; its output: plain f32 rows in `a` for the f32 head kernels
            Tower64SplitArgs ta{};
            ta.planes = d_planes, ta.layers = e->t64_layers.as<Tower64SplitLayer>(), ta.sat = e->conv_opts.saturated;
            ta.bias_all = e->t64s_bias.as<float>();
            ta.n = n, ta.C = d.planes, ta.w64 = w64, ta.S = S, ta.nlayers = 1 + 2 * d.blocks;
            if (e->t64s_fuse_heads) {
                ta.head_w = e->head_w.as<float>(), ta.head_b = e->head_b.as<float>(), ta.hv = L.hv.as<float>();
                ta.hv_pol = (uint32_t)((e->bpad + 31) / 32 * 32) * e->kvp, ta.kvp = e->kvp, ta.kpp = e->kpp, ta.vhc = d.vhc, ta.ocn = d.vhc + d.phc;
            } else {
                ta.out = (float*)a;
            }
            hipEvent_t s0 = ev(false), s1 = ev(true);
            launch_tower64_split(ta, nb * e->slots, e->t64s_depth, st, s0, s1);
        } else {
            // the stem conv expands the planes itself when they fit one 128-byte chunk (every game here); else K0 first
            const bool fused_stem = d.planes <= 32 && e->cpad0 == (uint32_t)act_kc(e->act) && !e->pack_separately;
            const StemInput stem_in{d_planes, n, d.planes, w64};
            if (!fused_stem) launch_pack_planes_nhwc(e->act, d_planes, n, nb, d.planes, w64, S, e->cpad0, L.x0.p, st);
            hipEvent_t s0 = ev(false), s1 = ev(true);
            // the split tower hands its last layer to the f32 head kernels as plain f32 rows
            const int last_flags = act_f16_family(e->act) ? CONV_OUT_F32 : 0;
            const bool wfrag = e->act == Act::F16S && e->split_wfrag;
            const int wflag = wfrag ? CONV_W_FRAG : 0;
            auto wptr = [&](const ConvLayer& c) { return wfrag ? c.wf.p : c.w.p; };
            launch_conv3x3_mfma(e->act, L.x0.p, wptr(e->stem), e->stem.b.as<float>(), nullptr, a, nb, e->cpad0, FP, S, st, s0, s1,
                                fused_stem ? &stem_in : nullptr,
                                wflag | (d.blocks == 0 ? last_flags : e->act == Act::F16S && e->c1[0]->wu.p ? CONV_OUT_F32 | CONV_WINO_IN : 0), e->conv_opts);
            // f16x2 with CATTUS_WINOGRAD=1 on 8x8 boards: every layer behind the stem in Winograd form, f32 rows between the layers
            const bool wino = d.blocks > 0 && e->c1[0]->wu.p != nullptr;
            auto conv = [&](const ConvLayer& c, const void* in, const void* res, void* out, int lflags) {
                hipEvent_t s0 = ev(false), s1 = ev(true);
                if (wino && e->wino_k8) launch_conv3x3_wino8((const float*)in, c.wu.p, c.bw.as<float>(), (const float*)res, (float*)out, nb, FP, FP, st, s0, s1, e->conv_opts.saturated);
                else if (wino && e->wino_k4) launch_conv3x3_wino4((const float*)in, c.wu.p, c.bw.as<float>(), (const float*)res, (float*)out, nb, FP, FP, st, s0, s1, e->conv_opts.saturated);
                else if (wino) launch_conv3x3_wino((const float*)in, c.wu.p, c.bw.as<float>(), (const float*)res, (float*)out, nb, FP, FP, st, s0, s1, e->conv_opts.saturated);
                else launch_conv3x3_mfma(e->act, in, wptr(c), c.b.as<float>(), res, out, nb, FP, FP, S, st, s0, s1, nullptr, wflag | lflags, e->conv_opts);
            };
            if (wino && !e->wino_k8 && L.tower_nlayers && e->persist_ok.load(std::memory_order_relaxed) && wino4_tower_fits(nb, FP, e->cus)) {
                // every layer behind the stem in ONE launch: counters and error word zeroed ahead of it on the same stream, the error word
                // copied to page-locked memory behind it (eval_host reads it when the batch is back; the device entry points at their next call)
                hipEvent_t s0 = ev(false), s1 = ev(true);
                unsigned* const tower_err = L.tower_ready.as<unsigned>();
                HIP_TRY(hipMemsetAsync(L.tower_ready.p, 0, 64 + (size_t)L.tower_nlayers * (nb / 4) * 4, st));
                int crc = chain_persistent_launch(e->device, st, [&] {
                    launch_tower_wino4(L.tower_layers.as<Wino4TowerLayer>(), L.tower_nlayers, tower_err + 16, tower_err,
                                       e->conv_opts.saturated, nb, FP, e->persist_spin, e->cus, st, s0, s1);
                });
                if (crc) return crc;
                HIP_TRY(hipMemcpyAsync(L.h_tower_err.p, tower_err, 4, hipMemcpyDeviceToHost, st));
            } else
            for (uint32_t i = 0; i < d.blocks; i++) {
                conv(*e->c1[i], a, nullptr, t, 0);
                if (wino && e->wino_inplace) {
                    // the block's output over its own skip rows, in place: the lane that adds a skip element is the lane that writes
                    // that element, behind all its reads -- two activation buffers instead of three (33.6 instead of 50 MB per lane
                    // of what a pass drags through the Infinity Cache beside the 168 MB of U)
                    conv(*e->c2[i], t, a, a, 0);
                    continue;
                }
                conv(*e->c2[i], t, a, y, i + 1 == d.blocks ? last_flags : 0);
                std::swap(a, y);
            }
        }
    } else {
        launch_planes_to_tensor_nchw(d_planes, n, d.planes, w64, S, n, L.x0.as<float>(), st);
        hipEvent_t s0 = ev(false), s1 = ev(true);
        launch_conv3x3_generic(L.x0.as<float>(), e->stem.w.as<float>(), e->stem.b.as<float>(), nullptr, (float*)a, n,
                               d.planes, F, S, st, s0, s1);
        for (uint32_t i = 0; i < d.blocks; i++) {
            s0 = ev(false), s1 = ev(true);
            launch_conv3x3_generic((float*)a, e->c1[i]->w.as<float>(), e->c1[i]->b.as<float>(), nullptr, (float*)t, n, F,
                                   F, S, st, s0, s1);
            s0 = ev(false), s1 = ev(true);
            launch_conv3x3_generic((float*)t, e->c2[i]->w.as<float>(), e->c2[i]->b.as<float>(), (float*)a, (float*)y, n,
                                   F, F, S, st, s0, s1);
            std::swap(a, y);
        }
    }
    const uint32_t kv = d.vhc * hw, kp = d.phc * hw;
    if (e->tuned) {
        HeadsMfma hd{};
        hd.conv_w = e->head_w.p, hd.conv_b = e->head_b.as<float>(), hd.hv = L.hv.p;
        hd.w1 = e->w1t.p, hd.b1 = e->b1.as<float>(), hd.h1 = L.h1.as<float>();
        hd.wp = e->wpt.p, hd.bp = e->bp.as<float>(), hd.policy = d_policy;
        hd.hw = hw, hd.vhc = d.vhc, hd.phc = d.phc, hd.kvp = e->kvp, hd.kpp = e->kpp, hd.M = d.moves;
        hd.w2 = e->w2.as<float>(), hd.b2 = e->b2.as<float>(), hd.value = d_value;
        hd.slots = e->slots;
        hd.hv_leaves = (e->bpad + 31) / 32 * 32;
        launch_heads_mfma(head_act(e->act), e->tower64 || (e->tower64s && e->t64s_fuse_heads) ? nullptr : a, n, e->fpad, hd, st);
    } else {
        TowerView tv;
        tv.x = a, tv.act = Act::F32, tv.sb = F * hw, tv.sk = hw, tv.sp = 1;
        launch_head_conv1x1(tv, e->head_w.as<float>(), e->head_b.as<float>(), n, F, d.vhc + d.phc, hw, L.hv.as<float>(), st);
        launch_value_fc1(L.hv.as<float>(), kv + kp, e->w1t.as<float>(), e->b1.as<float>(), n, kv, L.h1.as<float>(), st);
        launch_value_fc2_tanh(L.h1.as<float>(), e->w2.as<float>(), e->b2.as<float>(), n, d_value, st);
        launch_policy_fc(L.hv.as<float>(), kv + kp, kv, e->wpt.as<float>(), e->bp.as<float>(), n, kp, d.moves, d_policy, st);
    }
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(CATTUS_E_DEVICE, "kernel launch failed: %s", hipGetErrorString(err));
    return CATTUS_OK;
}

void account(cattus_eval* e, uint32_t n, double seconds) {
    std::lock_guard<std::mutex> lk(e->stat_mu);
    cattus_stats& s = e->stats;
    // RunningAverage::set with epsilon 0.99, starting from 0 (reference: engine/src/util/metric.rs:1-20,
    // engine/src/net/mod.rs:36): value = (1 - eps) * value + eps * new
    s.run_seconds_ema = 0.01 * s.run_seconds_ema + 0.99 * seconds;
    s.run_seconds_total += seconds;
    s.batches += 1;
    s.positions += n;
    if (n == e->cfg.max_batch) s.full_batches += 1;
}

struct LegalArgs {
    const uint16_t* idx;  // [n][stride] policy indices of the legal moves
    const uint16_t* cnt;  // [n]
    uint32_t stride;
    float* probs;  // [n][stride]
};

// Blocking host-buffer evaluation; caller holds no lock.  With `lg` the logits stay on the device and
// the per-leaf softmax over the legal moves comes back instead (policy is not written).
int eval_host(cattus_eval* e, const uint64_t* planes, uint32_t n, float* policy, float* value, const LegalArgs* lg = nullptr) {
    const cattus_net_desc& d = e->d;
    // a free lane if there is one, else queue on one of them in turn
    std::unique_lock<std::mutex> lk;
    Lane* lane = nullptr;
    for (Lane& cand : e->lanes) {
        std::unique_lock<std::mutex> tl(cand.mu, std::try_to_lock);
        if (tl.owns_lock()) {
            lane = &cand, lk = std::move(tl);
            break;
        }
    }
    if (!lane) {
        lane = &e->lanes[e->lane_rr.fetch_add(1) % NLANES];
        lk = std::unique_lock<std::mutex>(lane->mu);
    }
    Lane& L = *lane;
    const RoctxRange range(lg ? "cattus_hip_eval_legal" : "cattus_hip_eval", n);
    HIP_TRY(hipSetDevice(e->device));
    const auto t0 = std::chrono::steady_clock::now();
    const size_t pbytes = (size_t)n * d.planes * e->cfg.plane_words * 8;
    // Buffers obtained from cattus_hip_host_alloc are page-locked: DMA straight from / into them.
    // Anything else goes through the evaluator's own pinned staging buffers.
    const bool direct = is_pinned(planes) && (lg || is_pinned(policy)) && is_pinned(value);
    if (lg && (L.legal_stride < lg->stride || !L.d_probs.p)) {
        const size_t B = e->cfg.max_batch;
        int arc;
        if ((arc = L.d_legal_idx.alloc(B * lg->stride * 2)) || (arc = L.d_legal_cnt.alloc(B * 2)) ||
            (arc = L.d_probs.alloc(B * lg->stride * 4)))
            return arc;
        L.legal_stride = lg->stride;
    }
    const void* src_planes = planes;
    if (!direct) {
        memcpy(L.h_planes.p, planes, pbytes);
        src_planes = L.h_planes.p;
    }
    HIP_TRY(hipMemcpyAsync(L.d_planes.p, src_planes, pbytes, hipMemcpyHostToDevice, L.stream));
    auto copy_out = [&]() -> int {
        if (lg) {
            HIP_TRY(hipMemcpyAsync(L.d_legal_idx.p, lg->idx, (size_t)n * lg->stride * 2, hipMemcpyHostToDevice, L.stream));
            HIP_TRY(hipMemcpyAsync(L.d_legal_cnt.p, lg->cnt, (size_t)n * 2, hipMemcpyHostToDevice, L.stream));
            if (launch_legal_softmax(L.d_policy.as<float>(), d.moves, L.d_legal_idx.as<uint16_t>(), L.d_legal_cnt.as<uint16_t>(),
                                     lg->stride, n, L.d_probs.as<float>(), L.stream))
                return fail(CATTUS_E_INVALID, "legal stride %u exceeds 1024", lg->stride);
            HIP_TRY(hipMemcpyAsync(lg->probs, L.d_probs.p, (size_t)n * lg->stride * 4, hipMemcpyDeviceToHost, L.stream));
        } else {
            HIP_TRY(hipMemcpyAsync(direct ? (void*)policy : L.h_policy.p, L.d_policy.p, (size_t)n * d.moves * 4, hipMemcpyDeviceToHost, L.stream));
        }
        HIP_TRY(hipMemcpyAsync(direct ? (void*)value : L.h_value.p, L.d_value.p, (size_t)n * 4, hipMemcpyDeviceToHost, L.stream));
        return CATTUS_OK;
    };
    // Forward + copies + wait.  hipStreamSynchronize spins on the completion signal (lowest latency: +3 % self-play throughput on a 16-CPU
    // share with 8 search threads); CATTUS_HIP_WAIT=block sleeps on a blocking-sync event instead, which frees the core each waiting
    // thread would burn (for hosts short of CPUs).
    auto run_batch = [&]() -> int {
        int rc = enqueue_forward(e, L, L.d_planes.as<uint64_t>(), n, L.d_policy.as<float>(), L.d_value.as<float>(), L.stream);
        if (rc) return rc;
        if ((rc = copy_out())) return rc;
        if (e->wait_spin) {
            HIP_TRY(hipStreamSynchronize(L.stream));
        } else {
            HIP_TRY(hipEventRecord(L.done, L.stream));
            HIP_TRY(hipEventSynchronize(L.done));
        }
        return CATTUS_OK;
    };
    int rc = run_batch();
    if (rc) return rc;
    if (L.h_tower_err.p && *L.h_tower_err.as<volatile unsigned>() != 0) {
        // a hand-off wait of the one-launch tower ran out of its budget (its workgroups were not all resident: somebody else's kernels
        // on this device): the launch finished on rows that were not ready.  Nothing of it is kept: this evaluator goes back to the
        // per-layer launches, for this batch (again, from the planes) and for every later one.
        *L.h_tower_err.as<volatile unsigned>() = 0;
        e->persist_ok.store(false, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> sl(e->stat_mu);
            e->stats_tower_fallbacks += 1;
        }
        if ((rc = run_batch())) return rc;
    }
    if (!direct) {
        if (!lg) memcpy(policy, L.h_policy.p, (size_t)n * d.moves * 4);
        memcpy(value, L.h_value.p, (size_t)n * 4);
    }
    account(e, n, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    return CATTUS_OK;
}

ServerBatch* new_batch(cattus_eval* e) {
    auto b = std::make_unique<ServerBatch>();
    b->seq = e->next_seq++;
    b->planes.resize((size_t)e->cfg.max_batch * e->d.planes * e->cfg.plane_words);
    e->batches.push_back(std::move(b));
    return e->batches.back().get();
}

void server_loop(cattus_eval* e) {
    std::unique_lock<std::mutex> lk(e->srv_mu);
    for (;;) {
        // pick the oldest batch that is sealed and not yet run; seal the collecting one on deadline/flush
        ServerBatch* run = nullptr;
        for (auto& b : e->batches)
            if (b->sealed && !b->done && !b->running) {
                run = b.get();
                break;
            }
        if (!run && !e->batches.empty()) {
            ServerBatch* cur = e->batches.back().get();
            if (!cur->sealed && !cur->running && cur->count > 0) {
                const auto deadline = cur->t0 + std::chrono::microseconds(e->cfg.flush_us);
                if (e->flush_req || std::chrono::steady_clock::now() >= deadline) {
                    cur->sealed = true;
                    run = cur;
                } else if (!e->stop) {
                    e->srv_cv.wait_until(lk, deadline);
                    continue;
                }
            }
        }
        if (!run) {
            e->flush_req = false;
            if (e->stop) return;
            e->srv_cv.wait(lk);
            continue;
        }
        run->running = true;
        const uint32_t n = run->count;
        run->policy.resize((size_t)n * e->d.moves);
        run->value.resize(n);
        run->taken.assign(n, 0);
        lk.unlock();
        const int rc = eval_host(e, run->planes.data(), n, run->policy.data(), run->value.data());
        std::string err = rc ? g_last_error : std::string();
        lk.lock();
        run->status = rc;
        run->error = err;
        run->done = true;
        e->done_cv.notify_all();
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------ ABI

CATTUS_API const char* cattus_hip_last_error(void) { return g_last_error.c_str(); }
CATTUS_API const char* cattus_hip_version(void) { return "cattus_hip 0.2 (gfx950)"; }
CATTUS_API const char* cattus_hip_runtime_note(void) { return g_runtime_note.c_str(); }

CATTUS_API const char* cattus_hip_tower_kernel(const cattus_eval* e) {
    if (!e) return "";
    if (e->simple) return "policy_fc_kernel";
    if (!e->tuned) return "conv3x3_generic_kernel";
    if (e->tower64) return "tower64_lds_kernel";
    if (e->tower64s) return "tower64_split_kernel";
    if (e->act == Act::F16S) return e->d.blocks > 0 && e->c1[0]->wu.p ? (e->wino_k8 ? "conv3x3_wino8_kernel" : e->wino_k4 ? (e->lanes[0].tower_nlayers && e->persist_ok.load() && wino4_tower_fits(e->bpad, e->fpad, e->cus) ? "tower_wino4_kernel" : "conv3x3_wino4_kernel") : "conv3x3_wino_kernel") : e->split_wfrag ? "conv3x3_splitw_kernel" : "conv3x3_split_kernel";
    return "conv3x3_mfma_v2_kernel";
}

namespace {

// Diagnostic switches of cattus_hip_create_diag (include/cattus_hip_diag.h): "KEY=VALUE;KEY=VALUE".  cattus_hip_create passes none,
// and the library reads no A/B switch from the environment: what a host links is one code path per configuration.
struct Switches {
    std::map<std::string, std::string> kv;
    int parse(const char* text) {
        if (!text) return CATTUS_OK;
        static const char* const known[] = {"CATTUS_CONV_CB", "CATTUS_CONV_PBW", "CATTUS_FUSED_STEM", "CATTUS_T64_CH", "CATTUS_T64_LS", "CATTUS_SPLIT_W",
                                            "CATTUS_T64S_HEADS", "CATTUS_T64S_SHAPE", "CATTUS_TOWER64", "CATTUS_FORCE_GENERIC", "CATTUS_WINO_INPLACE",
                                            "CATTUS_ARENA", "CATTUS_WINO_KERNEL", "CATTUS_WINO_PERSIST", "CATTUS_WINO_SPIN"};
        const std::string s(text);
        size_t at = 0;
        while (at < s.size()) {
            size_t end = s.find(';', at);
            if (end == std::string::npos) end = s.size();
            const std::string item = s.substr(at, end - at);
            at = end + 1;
            if (item.empty()) continue;
            const size_t eq = item.find('=');
            if (eq == std::string::npos || eq == 0) return fail(CATTUS_E_INVALID, "diagnostic switch '%s' is not KEY=VALUE", item.c_str());
            const std::string key = item.substr(0, eq);
            bool ok = false;
            for (const char* k : known) ok = ok || key == k;
            if (!ok) return fail(CATTUS_E_INVALID, "unknown diagnostic switch '%s'", key.c_str());
            kv[key] = item.substr(eq + 1);
        }
        return CATTUS_OK;
    }
    const char* get(const char* key) const {
        auto it = kv.find(key);
        return it == kv.end() ? nullptr : it->second.c_str();
    }
};

int create_impl(const void* weights, size_t nbytes, const cattus_eval_config* cfg_in, const char* switches, cattus_eval** out) {
    if (!out) return fail(CATTUS_E_INVALID, "out is NULL");
    *out = nullptr;
    if (!weights || !cfg_in) return fail(CATTUS_E_INVALID, "weights/cfg is NULL");
    // struct_size versions the configuration: 24 bytes = the fields up to flush_us (tower_form = AUTO), 28 = with tower_form
    cattus_eval_config cfg_copy{};
    if (cfg_in->struct_size == offsetof(cattus_eval_config, tower_form)) memcpy(&cfg_copy, cfg_in, offsetof(cattus_eval_config, tower_form));
    else if (cfg_in->struct_size == sizeof(cattus_eval_config)) cfg_copy = *cfg_in;
    else return fail(CATTUS_E_INVALID, "cfg.struct_size mismatch");
    cfg_copy.struct_size = sizeof(cattus_eval_config);
    const cattus_eval_config* cfg = &cfg_copy;
    if (cfg->tower_form > CATTUS_TOWER_WINOGRAD) return fail(CATTUS_E_INVALID, "unknown tower_form %u", cfg->tower_form);
    Switches sw;
    if (int src = sw.parse(switches)) return src;
    if (nbytes < HEADER_BYTES || memcmp(weights, "CATTUSW1", 8) != 0) return fail(CATTUS_E_INVALID, "not a cattus weight blob");
    uint32_t h[9];
    memcpy(h, (const char*)weights + 8, sizeof h);
    if (h[0] != 1 || h[8] != FC_HIDDEN) return fail(CATTUS_E_INVALID, "unsupported blob version %u / hidden %u", h[0], h[8]);
    cattus_net_desc d{h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8]};
    const bool simple = d.filters == 0;  // SimpleTwoHeadedModel: no conv tower (cattus_amd/weights.py)
    if (d.board < 1 || d.board > 11 || !d.planes || !d.moves || (!simple && (!d.vhc || !d.phc)) || (simple && (d.blocks || d.vhc || d.phc)))
        return fail(CATTUS_E_INVALID, "bad network shape in blob header");
    if (simple && (size_t)d.planes * d.board * d.board > 2048) return fail(CATTUS_E_UNSUPPORTED, "SimpleTwoHeadedModel wider than 2048 features");
    if (nbytes != HEADER_BYTES + 4 * blob_floats(d)) return fail(CATTUS_E_INVALID, "blob size %zu does not match its header", nbytes);
    if (cfg->max_batch < 1 || cfg->max_batch > (1u << 20)) return fail(CATTUS_E_INVALID, "max_batch out of range");
    if ((uint64_t)cfg->plane_words * 64 < (uint64_t)d.board * d.board || cfg->plane_words > 2)
        return fail(CATTUS_E_INVALID, "plane_words %u cannot hold a %ux%u board", cfg->plane_words, d.board, d.board);
    if (d.planes * cfg->plane_words > 128) return fail(CATTUS_E_UNSUPPORTED, "more than 128 plane words per leaf");
    if (cfg->dtype != CATTUS_DTYPE_F32 && cfg->dtype != CATTUS_DTYPE_BF16 && cfg->dtype != CATTUS_DTYPE_F16X2 && cfg->dtype != CATTUS_DTYPE_F16)
        return fail(CATTUS_E_INVALID, "unknown dtype %u", cfg->dtype);
    if (!simple && (size_t)d.phc * d.board * d.board * 8 * 4 > 64 * 1024) return fail(CATTUS_E_UNSUPPORTED, "policy head too wide for the FC kernel");

    int ndev = 0;
    hipError_t herr = hipGetDeviceCount(&ndev);
    if (herr != hipSuccess || ndev <= 0)
        return fail(CATTUS_E_DEVICE, "no HIP device available (%s); this library has no CPU path", hipGetErrorString(herr));
    if (cfg->device < 0 || cfg->device >= ndev) return fail(CATTUS_E_INVALID, "device %d out of range (%d devices)", cfg->device, ndev);
    HIP_TRY(hipSetDevice(cfg->device));
    {
        // the tower kernels' dynamic-LDS opt-in: once per device, before anything can be launched on it
        static std::mutex prep_mu;
        static uint64_t prepared = 0;
        std::lock_guard<std::mutex> lk(prep_mu);
        const uint64_t bit = 1ull << (cfg->device & 63);
        if (!(prepared & bit)) {
            const hipError_t perr = prepare_device();
            if (perr != hipSuccess) return fail(CATTUS_E_DEVICE, "hipFuncSetAttribute failed: %s", hipGetErrorString(perr));
            prepared |= bit;
        }
    }
    // every environment switch is read here, once per evaluator: nothing on the evaluation path calls getenv
    {
        static std::mutex note_mu;
        std::lock_guard<std::mutex> lk(note_mu);
        const char* ka = getenv("HIP_FORCE_DEV_KERNARG");
        g_runtime_note = ka && ka[0] == '1' ? "HIP_FORCE_DEV_KERNARG=1: kernel arguments in device memory"
                                            : "HIP_FORCE_DEV_KERNARG is not 1: kernel arguments travel over the host link (about +8 % per batch); "
                                              "export HIP_FORCE_DEV_KERNARG=1 before the process initialises HIP";
    }
    const char* wait_mode = getenv("CATTUS_HIP_WAIT");  // operational, not A/B: how the host thread waits for a batch
    const char* conv_cb_env = sw.get("CATTUS_CONV_CB");
    const char* conv_pbw_env = sw.get("CATTUS_CONV_PBW");
    const char* fused_stem_env = sw.get("CATTUS_FUSED_STEM");
    const char* t64_ch_env = sw.get("CATTUS_T64_CH");
    const char* t64_ls_env = sw.get("CATTUS_T64_LS");
    const char* split_w_env = sw.get("CATTUS_SPLIT_W");
    const char* t64s_heads_env = sw.get("CATTUS_T64S_HEADS");
    const char* t64s_d_env = sw.get("CATTUS_T64S_SHAPE");
    const char* tower64_env = sw.get("CATTUS_TOWER64");

    std::unique_ptr<cattus_eval> e(new (std::nothrow) cattus_eval);
    if (!e) return fail(CATTUS_E_NOMEM, "out of memory");
    e->d = d;
    e->cfg = *cfg;
    if (e->cfg.flush_us == 0) e->cfg.flush_us = 200;
    e->device = cfg->device;
    e->wait_spin = !(wait_mode && strcmp(wait_mode, "block") == 0);
    e->pack_separately = fused_stem_env && fused_stem_env[0] == '0';
    e->t64_force_ch = t64_ch_env ? atoi(t64_ch_env) : 0;
    e->t64_layer_steps = !(t64_ls_env && atoi(t64_ls_env) == 0);
    e->split_wfrag = !(split_w_env && split_w_env[0] == '0');
    // the tower's form is part of the configuration (a leaf's bits must not depend on the batch it came in, so never per batch):
    // AUTO = Winograd for max_batch > 128 where the shape allows it; WINOGRAD on a shape it does not cover is refused below
    e->winograd = cfg->tower_form == CATTUS_TOWER_WINOGRAD || (cfg->tower_form == CATTUS_TOWER_AUTO && cfg->max_batch > 128);
    e->resident_tower = !(tower64_env && tower64_env[0] == '0');
    e->t64s_fuse_heads = !(t64s_heads_env && t64s_heads_env[0] == '0');
    e->t64s_depth = t64s_d_env ? atoi(t64s_d_env) : 0;
    if (e->t64s_depth != 1 && e->t64s_depth != 2 && e->t64s_depth != 9) e->t64s_depth = 0;
    e->conv_opts.cb = conv_cb_env ? atoi(conv_cb_env) : 0;
    e->conv_opts.pbw = conv_pbw_env ? atoi(conv_pbw_env) : 0;
    e->hw = d.board * d.board;
    // The MFMA tower covers every board up to 11x11 and any filter count (channels are padded to 64 with zeros);
    // the two 1x1 head convs share one 32-row MFMA tile.  Wider heads take the generic f32 path (one thread
    // per output, same arithmetic order), which otherwise serves as a checker only (CATTUS_FORCE_GENERIC=1).
    const char* force_generic = sw.get("CATTUS_FORCE_GENERIC");
    e->simple = simple;
    e->tuned = !simple && d.vhc + d.phc <= 32 && !(force_generic && force_generic[0] == '1');
    e->act = simple ? Act::F32
             : cfg->dtype == CATTUS_DTYPE_BF16 ? Act::BF16
             : cfg->dtype == CATTUS_DTYPE_F16X2 ? Act::F16S
             : cfg->dtype == CATTUS_DTYPE_F16 ? Act::F16
                                              : Act::F32;
    if (!e->tuned && e->act != Act::F32)
        return fail(CATTUS_E_UNSUPPORTED, "bf16 / f16 / f16x2 need the MFMA tower: value + policy head channels <= 32 (got %u + %u)", d.vhc, d.phc);
    if (act_f16_family(e->act)) {
        int src = e->d_saturated.alloc(sizeof(unsigned));
        if (src) return src;
        HIP_TRY(hipMemset(e->d_saturated.p, 0, sizeof(unsigned)));
        e->conv_opts.saturated = e->d_saturated.as<unsigned>();
    }
    if (act_f16_family(e->act)) {
        // the f16 towers' stems expand the planes themselves (no separate plane pack exists for them)
        if (d.planes > 32) return fail(CATTUS_E_UNSUPPORTED, "f16x2 / f16 take at most 32 input planes (got %u)", d.planes);
        e->pack_separately = false;
    }
    e->slots = tower_slots(d.board);
    e->fpad = e->tuned ? (d.filters + COUT_PER_WG - 1) / COUT_PER_WG * COUT_PER_WG : d.filters;
    const uint32_t bpw = e->tuned ? ROWS_PER_WG / e->slots : 1;
    e->bpad = (cfg->max_batch + bpw - 1) / bpw * bpw;
    const char* inplace_env = sw.get("CATTUS_WINO_INPLACE");
    e->wino_inplace = !(inplace_env && inplace_env[0] == '0');
    const char* arena_env = sw.get("CATTUS_ARENA");  // 0: every buffer its own allocation (A/B runs)
    {
        // the 4-frequency kernel wherever it covers the layer shape (filters a multiple of 64), else the 16-frequency one (128)
        const char* wk = sw.get("CATTUS_WINO_KERNEL");
        if (wk && strcmp(wk, "k16") != 0 && strcmp(wk, "k4") != 0 && strcmp(wk, "k8") != 0) return fail(CATTUS_E_INVALID, "CATTUS_WINO_KERNEL is k16, k4 or k8");
        e->wino_k8 = wk && strcmp(wk, "k8") == 0;
        e->wino_k4 = wk ? strcmp(wk, "k4") == 0 || e->wino_k8 : wino4_supported(e->bpad, e->fpad, e->fpad, d.board);
    }
    {
        const char* wp = sw.get("CATTUS_WINO_PERSIST");
        e->wino_persist = !(wp && wp[0] == '0');
        const char* ws = sw.get("CATTUS_WINO_SPIN");
        if (ws) e->persist_spin = (uint32_t)std::max(1L, atol(ws));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, cfg->device));
        e->cus = (uint32_t)prop.multiProcessorCount;
    }
    if (cfg->tower_form == CATTUS_TOWER_WINOGRAD && !(e->tuned && e->act == Act::F16S && d.blocks > 0 && wino_shape_ok(e.get(), e->fpad, e->fpad)))
        return fail(CATTUS_E_UNSUPPORTED, "tower_form WINOGRAD needs dtype f16x2, an 8x8 board, at least one residual block and a multiple of 64 filters");
    if (e->tuned && e->act == Act::F16S && e->winograd && d.blocks > 0 && wino_shape_ok(e.get(), e->fpad, e->fpad) &&
        !(arena_env && arena_env[0] == '0')) {
        // the Winograd tower's hot set in one block: U of every layer, then the lanes' activation buffers (DevArena)
        auto page = [](size_t b) { return (b + 4095) & ~(size_t)4095; };
        const size_t FPz = e->fpad, u_bytes = page(((size_t)16 * FPz * FPz * 2 + (size_t)WINO_RING_STAGES * 1024) * 2);
        const size_t act_bytes_ = page((size_t)e->bpad * e->slots * FPz * 4);
        const size_t want = 2 * (size_t)d.blocks * u_bytes + (size_t)NLANES * (e->wino_inplace ? 2 : 3) * act_bytes_ + (1u << 20);
        void* base = nullptr;
        if (hipMalloc(&base, want) == hipSuccess) e->arena.base = (char*)base, e->arena.cap = want;
        else (void)hipGetLastError();  // no room for one block: separate allocations, as before
    }
    int rc = build(e.get(), reinterpret_cast<const float*>((const char*)weights + HEADER_BYTES));
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    {
        std::lock_guard<std::mutex> lk(e->srv_mu);
        new_batch(e.get());
    }
    for (auto& t : e->servers) t = std::thread(server_loop, e.get());
    *out = e.release();
    return CATTUS_OK;
}

}  // namespace

CATTUS_API int cattus_hip_create(const void* weights, size_t nbytes, const cattus_eval_config* cfg, cattus_eval** out) {
    return create_impl(weights, nbytes, cfg, nullptr, out);
}

CATTUS_API int cattus_hip_create_diag(const void* weights, size_t nbytes, const cattus_eval_config* cfg, const char* switches, cattus_eval** out) {
    return create_impl(weights, nbytes, cfg, switches, out);
}

CATTUS_API void cattus_hip_destroy(cattus_eval* e) { delete e; }

CATTUS_API int cattus_hip_desc(const cattus_eval* e, cattus_net_desc* out) {
    if (!e || !out) return fail(CATTUS_E_INVALID, "NULL argument");
    *out = e->d;
    return CATTUS_OK;
}

CATTUS_API int cattus_hip_eval(cattus_eval* e, const uint64_t* planes, uint32_t n, float* policy, float* value) {
    if (!e || !planes || !policy || !value) return fail(CATTUS_E_INVALID, "NULL argument");
    // planes_to_tensor asserts 1 <= n <= batch_size (engine/src/net/mod.rs:122-127)
    if (n < 1 || n > e->cfg.max_batch) return fail(CATTUS_E_INVALID, "invalid sample len %u, 1..=%u", n, e->cfg.max_batch);
    return eval_host(e, planes, n, policy, value);
}

CATTUS_API int cattus_hip_eval_legal(cattus_eval* e, const uint64_t* planes, uint32_t n, const uint16_t* legal_idx,
                                     const uint16_t* legal_count, uint32_t legal_stride, float* probs, float* value) {
    if (!e || !planes || !legal_idx || !legal_count || !probs || !value) return fail(CATTUS_E_INVALID, "NULL argument");
    if (n < 1 || n > e->cfg.max_batch) return fail(CATTUS_E_INVALID, "invalid sample len %u, 1..=%u", n, e->cfg.max_batch);
    if (legal_stride < 1 || legal_stride > 1024) return fail(CATTUS_E_INVALID, "legal_stride %u outside 1..=1024", legal_stride);
    for (uint32_t i = 0; i < n; i++) {
        if (legal_count[i] > legal_stride) return fail(CATTUS_E_INVALID, "leaf %u: %u legal moves > stride %u", i, legal_count[i], legal_stride);
        for (uint32_t k = 0; k < legal_count[i]; k++)
            if (legal_idx[(size_t)i * legal_stride + k] >= e->d.moves)
                return fail(CATTUS_E_INVALID, "leaf %u: policy index %u >= %u", i, legal_idx[(size_t)i * legal_stride + k], e->d.moves);
    }
    const LegalArgs lg{legal_idx, legal_count, legal_stride, probs};
    return eval_host(e, planes, n, nullptr, value, &lg);
}

CATTUS_API int cattus_hip_eval_device(cattus_eval* e, const uint64_t* d_planes, uint32_t n, float* d_policy, float* d_value,
                                      void* stream) {
    return cattus_hip_eval_device_lane(e, 0, d_planes, n, d_policy, d_value, stream);
}

CATTUS_API int cattus_hip_eval_device_lane(cattus_eval* e, uint32_t lane, const uint64_t* d_planes, uint32_t n, float* d_policy,
                                           float* d_value, void* stream) {
    if (!e || !d_planes || !d_policy || !d_value) return fail(CATTUS_E_INVALID, "NULL argument");
    if (n < 1 || n > e->cfg.max_batch) return fail(CATTUS_E_INVALID, "invalid sample len %u, 1..=%u", n, e->cfg.max_batch);
    if (lane >= NLANES) return fail(CATTUS_E_INVALID, "lane %u out of range (%d lanes)", lane, NLANES);
    Lane& L = e->lanes[lane];
    std::lock_guard<std::mutex> lk(L.mu);
    HIP_TRY(hipSetDevice(e->device));
    if (L.h_tower_err.p && *L.h_tower_err.as<volatile unsigned>() != 0) {
        // this entry point is asynchronous: what an earlier call's one-launch tower reported is seen here, at the next call
        *L.h_tower_err.as<volatile unsigned>() = 0;
        e->persist_ok.store(false, std::memory_order_relaxed);
        return fail(CATTUS_E_DEVICE, "an earlier batch on lane %u ran the tower as one launch and a hand-off wait in it gave up (the device was shared): that batch's "
                                     "outputs are invalid; this evaluator uses the per-layer launches from now on", lane);
    }
    hipStream_t st = (hipStream_t)stream;  // as HIP itself: NULL is the legacy default stream, not a private one
    int rc = enqueue_forward(e, L, d_planes, n, d_policy, d_value, st);
    if (rc) return rc;
    std::lock_guard<std::mutex> sl(e->stat_mu);
    e->stats.batches += 1;
    e->stats.positions += n;
    if (n == e->cfg.max_batch) e->stats.full_batches += 1;
    return CATTUS_OK;
}

CATTUS_API int cattus_hip_lane_stream(cattus_eval* e, uint32_t lane, void** stream) {
    if (!e || !stream) return fail(CATTUS_E_INVALID, "NULL argument");
    if (lane >= NLANES) return fail(CATTUS_E_INVALID, "lane %u out of range (%d lanes)", lane, NLANES);
    *stream = (void*)e->lanes[lane].stream;
    return CATTUS_OK;
}

CATTUS_API int cattus_hip_submit(cattus_eval* e, const uint64_t* planes_one, uint64_t* ticket) {
    if (!e || !planes_one || !ticket) return fail(CATTUS_E_INVALID, "NULL argument");
    const size_t words = (size_t)e->d.planes * e->cfg.plane_words;
    std::unique_lock<std::mutex> lk(e->srv_mu);
    if (e->stop) return fail(CATTUS_E_STATE, "evaluator is shutting down");
    // the deque can be empty: a deadline-sealed batch may already have been collected and erased
    ServerBatch* cur = e->batches.empty() ? nullptr : e->batches.back().get();
    if (!cur || cur->sealed) cur = new_batch(e);
    const uint32_t slot = cur->count++;
    if (slot == 0) cur->t0 = std::chrono::steady_clock::now();
    memcpy(cur->planes.data() + slot * words, planes_one, words * 8);
    *ticket = (cur->seq << 32) | slot;
    const bool full = cur->count == e->cfg.max_batch;
    if (full) {
        cur->sealed = true;
        new_batch(e);
    }
    lk.unlock();
    if (full || slot == 0) e->srv_cv.notify_all();
    return CATTUS_OK;
}

CATTUS_API int cattus_hip_wait(cattus_eval* e, uint64_t ticket, float* policy, float* value) {
    if (!e || !policy || !value) return fail(CATTUS_E_INVALID, "NULL argument");
    const uint64_t seq = ticket >> 32;
    const uint32_t slot = (uint32_t)ticket;
    std::unique_lock<std::mutex> lk(e->srv_mu);
    for (;;) {
        ServerBatch* b = nullptr;
        for (auto& x : e->batches)
            if (x->seq == seq) {
                b = x.get();
                break;
            }
        if (!b || slot >= b->count) return fail(CATTUS_E_STATE, "unknown or already collected ticket %llu", (unsigned long long)ticket);
        if (b->done) {
            // a ticket is single-use whether its batch succeeded or not: the last waiter through erases the batch
            if (slot >= b->taken.size() || b->taken[slot]) return fail(CATTUS_E_STATE, "ticket %llu was already collected", (unsigned long long)ticket);
            b->taken[slot] = 1;
            const int status = b->status;
            if (status != CATTUS_OK) {
                g_last_error = b->error;
            } else {
                memcpy(policy, b->policy.data() + (size_t)slot * e->d.moves, (size_t)e->d.moves * 4);
                *value = b->value[slot];
            }
            if (++b->collected == b->count) {
                for (auto it = e->batches.begin(); it != e->batches.end(); ++it)
                    if (it->get() == b) {
                        e->batches.erase(it);
                        break;
                    }
            }
            return status;
        }
        if (e->stop) return fail(CATTUS_E_STATE, "evaluator is shutting down");
        e->done_cv.wait(lk);
    }
}

CATTUS_API int cattus_hip_apply(cattus_eval* e, const uint64_t* planes, uint32_t n, float* policy, float* value) {
    if (!e || !planes || !policy || !value) return fail(CATTUS_E_INVALID, "NULL argument");
    const size_t words = (size_t)e->d.planes * e->cfg.plane_words;
    std::vector<uint64_t> tickets(n);
    int rc = CATTUS_OK;
    uint32_t submitted = 0;
    for (; submitted < n; submitted++)
        if ((rc = cattus_hip_submit(e, planes + submitted * words, &tickets[submitted]))) break;
    // every submitted ticket is collected, also after a failure, so that no batch stays behind
    for (uint32_t i = 0; i < submitted; i++) {
        const int wrc = cattus_hip_wait(e, tickets[i], policy + (size_t)i * e->d.moves, value + i);
        if (wrc && !rc) rc = wrc;
    }
    return rc;
}

CATTUS_API int cattus_hip_flush(cattus_eval* e) {
    if (!e) return fail(CATTUS_E_INVALID, "NULL argument");
    {
        std::lock_guard<std::mutex> lk(e->srv_mu);
        e->flush_req = true;
    }
    e->srv_cv.notify_all();
    return CATTUS_OK;
}

CATTUS_API void* cattus_hip_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) {
        fail(CATTUS_E_NOMEM, "hipHostMalloc(%zu) failed", bytes);
        return nullptr;
    }
    std::unique_lock<std::shared_mutex> lk(g_pin_mu);
    g_pinned[(const char*)p] = bytes ? bytes : 16;
    return p;
}

CATTUS_API void cattus_hip_host_free(void* p) {
    if (!p) return;
    {
        std::unique_lock<std::shared_mutex> lk(g_pin_mu);
        g_pinned.erase((const char*)p);
    }
    (void)hipHostFree(p);
}

CATTUS_API int cattus_hip_stats(cattus_eval* e, cattus_stats* out) {
    if (!e || !out) return fail(CATTUS_E_INVALID, "NULL argument");
    unsigned sat = 0;
    if (e->d_saturated.p) {  // a 4-byte read of the sticky device counter (batches still in flight may add to it later)
        HIP_TRY(hipSetDevice(e->device));
        HIP_TRY(hipMemcpy(&sat, e->d_saturated.p, sizeof sat, hipMemcpyDeviceToHost));
    }
    std::lock_guard<std::mutex> lk(e->stat_mu);
    e->stats.saturated = sat;
    *out = e->stats;
    return CATTUS_OK;
}

CATTUS_API int cattus_hip_time_tower(cattus_eval* e, uint32_t n, uint32_t reps, float* avg_launch_us, uint32_t* launches) {
    if (!e || !avg_launch_us || !launches) return fail(CATTUS_E_INVALID, "NULL argument");
    if (n < 1 || n > e->cfg.max_batch || reps < 1) return fail(CATTUS_E_INVALID, "bad n/reps");
    if (e->simple) return fail(CATTUS_E_UNSUPPORTED, "a SimpleTwoHeadedModel has no conv tower to time");
    Lane& L = e->lanes[0];
    std::lock_guard<std::mutex> lk(L.mu);
    HIP_TRY(hipSetDevice(e->device));
    // (the one-launch Winograd tower is reported per LAYER: its duration + the stem's over 1 + 2 blocks, so that a caller's
    // flops-per-launch arithmetic does not change with how the layers are launched)
    const uint32_t per_fwd = e->tower64 || e->tower64s ? 1 : 1 + 2 * e->d.blocks;
    TowerTimer tt;
    tt.ev.resize((size_t)2 * per_fwd);
    for (auto& ev : tt.ev) HIP_TRY(hipEventCreate(&ev));
    HIP_TRY(hipMemsetAsync(L.d_planes.p, 0x5a, (size_t)n * e->d.planes * e->cfg.plane_words * 8, L.stream));
    double total_ms = 0;
    int rc = CATTUS_OK;
    for (uint32_t rep = 0; rep < reps + 1 && rc == CATTUS_OK; rep++) {  // first pass is warm-up
        tt.used = 0;
        rc = enqueue_forward(e, L, L.d_planes.as<uint64_t>(), n, L.d_policy.as<float>(), L.d_value.as<float>(), L.stream, &tt);
        if (rc) break;
        hipError_t err = hipStreamSynchronize(L.stream);
        if (err != hipSuccess) {
            rc = fail(CATTUS_E_DEVICE, "hipStreamSynchronize: %s", hipGetErrorString(err));
            break;
        }
        if (rep == 0) continue;
        for (size_t i = 0; i + 1 < tt.used; i += 2) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, tt.ev[i], tt.ev[i + 1]);
            total_ms += ms;
        }
    }
    for (auto& ev : tt.ev) (void)hipEventDestroy(ev);
    if (rc) return rc;
    *launches = per_fwd;
    *avg_launch_us = (float)(total_ms * 1000.0 / ((double)reps * per_fwd));
    return CATTUS_OK;
}

CATTUS_API int cattus_hip_mfma_sustained(cattus_eval* e, double seconds, double* tflops) {
    if (!e || !tflops) return fail(CATTUS_E_INVALID, "NULL argument");
    if (!(seconds > 0) || seconds > 30) return fail(CATTUS_E_INVALID, "seconds out of range");
    Lane& L = e->lanes[0];
    std::lock_guard<std::mutex> lk(L.mu);
    HIP_TRY(hipSetDevice(e->device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, e->device));
    const int cus = prop.multiProcessorCount;
    DevBuf out;
    int rc = out.alloc((size_t)cus * 256 * 4);
    if (rc) return rc;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    // launches of ~50 us (the length of a conv layer), groups of 50, until `seconds` have passed: the last group's rate
    const int iters = e->act == Act::F32 ? 480 : 640, reps = 50;
    double rate = 0, elapsed = 0;
    while (elapsed < seconds) {
        double flop = 0;
        (void)hipEventRecord(e0, L.stream);
        for (int r = 0; r < reps; r++) flop += launch_mfma_sustain(e->act, cus, iters, out.as<float>(), L.stream);
        (void)hipEventRecord(e1, L.stream);
        const hipError_t err = hipEventSynchronize(e1);
        if (err != hipSuccess) {
            (void)hipEventDestroy(e0), (void)hipEventDestroy(e1);
            return fail(CATTUS_E_DEVICE, "hipEventSynchronize: %s", hipGetErrorString(err));
        }
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        elapsed += ms * 1e-3;
        rate = flop / (ms * 1e-3) / 1e12;
    }
    (void)hipEventDestroy(e0), (void)hipEventDestroy(e1);
    *tflops = rate;
    return CATTUS_OK;
}

CATTUS_API int cattus_hip_planes_to_tensor_device(const uint64_t* d_planes, uint32_t n, uint32_t C, uint32_t plane_words,
                                                  uint32_t S, uint32_t batch, float* d_out, void* stream) {
    if (!d_planes || !d_out) return fail(CATTUS_E_INVALID, "NULL argument");
    if (n < 1 || n > batch) return fail(CATTUS_E_INVALID, "invalid sample len %u, 1..=%u", n, batch);
    if (S < 1 || S > 11 || plane_words * 64 < S * S || !C) return fail(CATTUS_E_INVALID, "bad plane geometry");
    launch_planes_to_tensor_nchw(d_planes, n, C, plane_words, S, batch, d_out, (hipStream_t)stream);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(CATTUS_E_DEVICE, "kernel launch failed: %s", hipGetErrorString(err));
    return CATTUS_OK;
}

CATTUS_API int cattus_hip_planes_to_tensor(int device, const uint64_t* planes, uint32_t n, uint32_t C, uint32_t plane_words,
                                           uint32_t S, uint32_t batch, float* out) {
    if (!planes || !out) return fail(CATTUS_E_INVALID, "NULL argument");
    if (n < 1 || n > batch) return fail(CATTUS_E_INVALID, "invalid sample len %u, 1..=%u", n, batch);
    if (S < 1 || S > 11 || plane_words * 64 < S * S || !C) return fail(CATTUS_E_INVALID, "bad plane geometry");
    int ndev = 0;
    hipError_t herr = hipGetDeviceCount(&ndev);
    if (herr != hipSuccess || device < 0 || device >= ndev)
        return fail(CATTUS_E_DEVICE, "no usable HIP device %d (%s); this library has no CPU path", device, hipGetErrorString(herr));
    HIP_TRY(hipSetDevice(device));
    DevBuf dp, dout;
    int rc;
    const size_t pbytes = (size_t)n * C * plane_words * 8, obytes = (size_t)batch * C * S * S * 4;
    if ((rc = dp.upload(planes, pbytes))) return rc;
    if ((rc = dout.alloc(obytes))) return rc;
    rc = cattus_hip_planes_to_tensor_device(dp.as<uint64_t>(), n, C, plane_words, S, batch, dout.as<float>(), nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(out, dout.p, obytes, hipMemcpyDeviceToHost));
    return CATTUS_OK;
}
