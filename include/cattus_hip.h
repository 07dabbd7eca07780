/*
 * cattus_hip.h -- C ABI of the MI355X leaf evaluator for Cattus self-play.
 *
 * This library is the drop-in for the reference's leaf-evaluation seam: the closure body of
 * Batcher::apply inside NNetwork::evaluate_impl (engine/src/net/mod.rs:94-98), i.e.
 *     planes_to_tensor (engine/src/net/mod.rs:121-156)
 *  -> NNetwork::run_net (engine/src/net/mod.rs:41-72)
 *  -> Model::run        (engine/src/net/model.rs:146-218)
 * Input per leaf: the position's bitboard planes (position_to_planes,
 * engine/src/chess/net/mod.rs:19-60, hex/net.rs:14-24, ttt/net.rs:14-24) for the position
 * already flipped so Player1 is to move (net/mod.rs:79).  Output per leaf: the M raw policy
 * logits with non-finite values replaced by f32::MIN (net/mod.rs:56-61) and the tanh value.
 * Everything above the seam (flip, cache, legal-move softmax, MCTS) is unchanged.
 *
 * Conventions: every entry point returns 0 (CATTUS_OK) or a negative cattus_status; it never
 * throws, aborts or keeps a caller pointer.  cattus_hip_last_error() returns a thread-local
 * message for the last failing call on the calling thread.  All entry points are thread-safe.
 */
#ifndef CATTUS_HIP_H
#define CATTUS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cattus_eval cattus_eval;

typedef enum cattus_status {
    CATTUS_OK = 0,
    CATTUS_E_INVALID = -1,     /* bad argument (the reference would panic: model.rs:100-125) */
    CATTUS_E_UNSUPPORTED = -2, /* shape/dtype combination this build has no kernel for */
    CATTUS_E_DEVICE = -3,      /* HIP runtime error; message in last_error */
    CATTUS_E_NOMEM = -4,
    CATTUS_E_STATE = -5,       /* unknown / already-collected ticket, evaluator shut down */
} cattus_status;

typedef enum cattus_dtype {
    CATTUS_DTYPE_F32 = 0,  /* exact f32 MFMA, bit-identical to the CPU oracle's summation order */
    CATTUS_DTYPE_BF16 = 1, /* bf16 operands / f32 accumulate MFMA tower (throughput mode, 8 significant bits) */
    /* Split precision: every activation and weight of the conv tower is a pair of f16 values (hi + lo, 22 significant
     * bits), a product is three f16 MFMA terms accumulated in f32, the heads run in exact f32.  Error against a
     * float64 run of the network is that of an f32 runtime (the reference's cross-runtime tolerance,
     * training/tests/test_net_output.py:28-33), at about a third of the bf16 tower's rate. */
    CATTUS_DTYPE_F16X2 = 2,
    /* Single-term f16: f16 operands and activations (11 significant bits; weights pre-scaled per output channel by a power
     * of two), f32 accumulation, f32 heads.  A throughput mode like bf16 -- same MFMA count and kernel -- with three more
     * significand bits; outside the reference's cross-runtime tolerance (DESIGN.md section 4 for what it does to a search). */
    CATTUS_DTYPE_F16 = 3,
} cattus_dtype;

/* Form of the f16x2 conv tower on 8x8 boards with a multiple of 128 filters.  The two forms agree to < 2e-6 per logit but not
 * bit for bit, and a leaf's bits must not depend on the batch it came in: the form is fixed per evaluator, here. */
typedef enum cattus_tower_form {
    CATTUS_TOWER_AUTO = 0,     /* Winograd where the shape allows it and max_batch > 128, else direct */
    CATTUS_TOWER_DIRECT = 1,   /* direct 3x3 (conv3x3_splitw_kernel): the faster form up to 128 leaves per batch */
    CATTUS_TOWER_WINOGRAD = 2, /* Winograd F(2x2,3x3) whatever max_batch is; CATTUS_E_UNSUPPORTED where no such kernel exists */
} cattus_tower_form;

/* Replaces the reference's InferenceConfig + batch_size (engine/src/net/model.rs:17-25,
 * training/self-play/src/self_play_cmd.rs:41-44). */
typedef struct cattus_eval_config {
    uint32_t struct_size; /* sizeof(cattus_eval_config); 24 (the fields up to flush_us, tower_form = AUTO) is accepted too */
    int32_t device;       /* HIP device ordinal */
    uint32_t max_batch;   /* model.batch_size: largest n accepted by eval / batch the server fills */
    uint32_t plane_words; /* u64 words per bitboard plane: chess 1, ttt 1, hex 2 (u128 as lo,hi) */
    uint32_t dtype;       /* cattus_dtype */
    uint32_t flush_us;    /* partial-batch deadline of the leaf server (reference: 20 ms, net/mod.rs:96) */
    uint32_t tower_form;  /* cattus_tower_form; ignored by every dtype but f16x2 */
} cattus_eval_config;

typedef struct cattus_stats {
    uint64_t batches;       /* == reference metric model.activation_count (net/mod.rs:68) */
    uint64_t positions;     /* real (non-padded) leaves evaluated */
    uint64_t full_batches;  /* batches that ran with n == max_batch */
    double run_seconds_ema; /* == reference metric model.run_duration (EMA 0.99, util/metric.rs:16-19) */
    double run_seconds_total;
    /* f16x2 / f16 towers: activation values that exceeded the f16 range (65504; in the Winograd form of the f16x2 tower a
     * quarter of it, 16376: its transformed inputs are sums of four) and were clamped, since the evaluator was created (sticky).  0 for every network inside the range; > 0 means outputs of this evaluator are NOT the network's:
     * use dtype f32 for that network (a BatchNorm scale of 1e5 does it; one of 200 does not). */
    uint64_t saturated;
} cattus_stats;

/* Network shape as stored in the weight blob header (cattus_amd/weights.py).  filters == 0 (with blocks = vhc = phc = 0) is
 * the reference's other model type, SimpleTwoHeadedModel (training/cattus_train/net_utils.py:92-121: two dense layers and two
 * dense heads on the flattened planes); it is evaluated in f32 whatever dtype is configured. */
typedef struct cattus_net_desc {
    uint32_t planes, board, moves, blocks, filters, vhc, phc, fc_hidden;
} cattus_net_desc;

/* Model::new (model.rs:61): parse the weight blob (copied), fold BatchNorm, upload.  Everything that selects a code path is in
 * `cfg`; the library reads two operational environment variables and no other: CATTUS_HIP_WAIT=block (the host thread sleeps
 * on an event instead of spinning while a batch runs) and CATTUS_ROCTX=1 (ROCTx ranges around every batch).  The A/B switches
 * of the tests and timing scripts go through cattus_hip_create_diag (cattus_hip_diag.h), never through the environment. */
int cattus_hip_create(const void* weights, size_t nbytes, const cattus_eval_config* cfg, cattus_eval** out);
void cattus_hip_destroy(cattus_eval* e);
int cattus_hip_desc(const cattus_eval* e, cattus_net_desc* out);

/* planes_to_tensor + run_net for n leaves, blocking (1 <= n <= max_batch, net/mod.rs:122-127).
 * planes: [n][planes][plane_words] u64 host memory; policy: [n][moves]; value: [n]. */
int cattus_hip_eval(cattus_eval* e, const uint64_t* planes, uint32_t n, float* policy, float* value);

/* cattus_hip_eval followed by calc_moves_probs (engine/src/net/mod.rs:100-119) on the device: the
 * logits stay in HBM and each leaf's softmax over its legal moves comes back instead.
 * legal_idx: [n][legal_stride] policy indices (Move::to_nn_idx) of the leaf's legal moves, in the
 * order the caller wants the probabilities; legal_count: [n], each <= legal_stride <= 1024;
 * probs: [n][legal_stride], entries past legal_count[i] are 0.  max = fold(f32::MIN, max), the sum
 * runs in move order, exp is the evaluator's own, so a host restatement
 * reproduces the result bit for bit. */
int cattus_hip_eval_legal(cattus_eval* e, const uint64_t* planes, uint32_t n, const uint16_t* legal_idx,
                          const uint16_t* legal_count, uint32_t legal_stride, float* probs, float* value);

/* Same with every buffer already resident in device memory (HBM) and asynchronous on `stream`, a
 * hipStream_t taken as HIP takes it: NULL is the legacy default stream.  All work of the call is enqueued
 * on that stream and nowhere else, so the caller orders it (and reads d_policy / d_value) with the
 * stream's own means: later work on the same stream, an event, or hipStreamSynchronize.  The evaluator's
 * intermediate buffers belong to a lane (below): two calls on one lane must not overlap on the device.
 * Used by bench.py so that the timed region excludes PCIe. */
int cattus_hip_eval_device(cattus_eval* e, const uint64_t* d_planes, uint32_t n, float* d_policy,
                           float* d_value, void* stream);

/* The evaluator keeps CATTUS_HIP_LANES independent sets of activation buffers.  cattus_hip_eval and the
 * leaf server pick a free one per call, so that many host threads can each have a batch in flight;
 * cattus_hip_eval_device uses lane 0.  This variant names the lane: work enqueued on different lanes
 * and different streams may overlap on the device (the tail of one batch's kernels with the head of
 * the other's).  Two calls on the same lane must be ordered by the caller (same stream or events). */
#ifndef CATTUS_HIP_LANES
#define CATTUS_HIP_LANES 2
#endif
int cattus_hip_eval_device_lane(cattus_eval* e, uint32_t lane, const uint64_t* d_planes, uint32_t n,
                                float* d_policy, float* d_value, void* stream);
/* The lane's own non-blocking stream (the one cattus_hip_eval uses for that lane), for callers of the
 * device entry points that have no stream of their own. */
int cattus_hip_lane_stream(cattus_eval* e, uint32_t lane, void** stream);

/* Leaf-batching server, the replacement of Batcher::apply (engine/src/util/batch.rs:49-177):
 * submit copies one leaf's planes and returns a ticket; wait blocks until that leaf's batch ran
 * and copies its logits/value out.  A batch runs when max_batch leaves are queued, when the oldest
 * queued leaf is flush_us old, or on cattus_hip_flush. */
int cattus_hip_submit(cattus_eval* e, const uint64_t* planes_one, uint64_t* ticket);
int cattus_hip_wait(cattus_eval* e, uint64_t ticket, float* policy, float* value);
int cattus_hip_flush(cattus_eval* e);
/* Batcher::apply as the reference's worker threads call it (engine/src/util/batch.rs:49-177,
 * net/mod.rs:94-98): a blocking call with the signature of cattus_hip_eval that goes through the leaf
 * server, i.e. the n leaves (n = 1 from a search thread) share batches with whatever the other calling
 * threads submit meanwhile; a batch runs when max_batch leaves are queued or flush_us have passed. */
int cattus_hip_apply(cattus_eval* e, const uint64_t* planes, uint32_t n, float* policy, float* value);

/* Page-locked host memory.  cattus_hip_eval transfers directly from / into buffers allocated here
 * (no staging copy); any other host memory works too, through the evaluator's own staging buffers. */
void* cattus_hip_host_alloc(size_t bytes);
void cattus_hip_host_free(void* p);

int cattus_hip_stats(cattus_eval* e, cattus_stats* out);

/* Average device time in microseconds of one 3x3-conv tower launch over `reps` forwards of n
 * leaves.  Every launch carries its own start/stop HIP event pair stamped by the kernel dispatch
 * itself (hipExtLaunchKernelGGL) on the evaluator's stream.  Also returns the number of such
 * launches per forward. */
int cattus_hip_time_tower(cattus_eval* e, uint32_t n, uint32_t reps, float* avg_launch_us, uint32_t* launches);

/* Diagnostic: the rate (TFLOP/s) the matrix pipe of the evaluator's device sustains on nothing but back-to-back MFMAs of
 * the evaluator's tower kind (f16x2: v_mfma_f32_32x32x16_f16, bf16: ..._bf16, f32: v_mfma_f32_32x32x2_f32) -- one wave per
 * SIMD on every CU, operands in registers, launches of ~50 us repeated for `seconds` (0 < seconds <= 30), the last group's
 * rate.  The nominal peaks are 2.4 GHz figures; under MFMA load the device's power management sets the clock. */
int cattus_hip_mfma_sustained(cattus_eval* e, double seconds, double* tflops);

/* Stand-alone planes_to_tensor (engine/src/net/mod.rs:121-156): host planes [n][C][plane_words]
 * -> host f32 tensor [batch][C][S][S], rows n..batch zero. */
int cattus_hip_planes_to_tensor(int device, const uint64_t* planes, uint32_t n, uint32_t C, uint32_t plane_words,
                                uint32_t S, uint32_t batch, float* out);
/* Device-resident variant, asynchronous on `stream`. */
int cattus_hip_planes_to_tensor_device(const uint64_t* d_planes, uint32_t n, uint32_t C, uint32_t plane_words,
                                       uint32_t S, uint32_t batch, float* d_out, void* stream);

/* Name of the kernel that runs this evaluator's tower (the dominant kernel of a forward pass): "conv3x3_splitw_kernel",
 * "tower_wino4_kernel" (f16x2 on 8x8 boards with max_batch > 128: the Winograd tower, every layer behind the stem in one launch),
 * "tower64_split_kernel", "conv3x3_mfma_v2_kernel", ... */
const char* cattus_hip_tower_kernel(const cattus_eval* e);

const char* cattus_hip_last_error(void);
const char* cattus_hip_version(void);
/* What the last cattus_hip_create found for HIP_FORCE_DEV_KERNARG (the host process exports it before HIP initialises:
 * kernel arguments in device memory, -8 % per batch; the library never changes the environment itself). */
const char* cattus_hip_runtime_note(void);

#ifdef __cplusplus
}
#endif
#endif /* CATTUS_HIP_H */
