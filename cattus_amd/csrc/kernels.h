// Internal launch interface between evaluator.hip (host logic) and kernels.hip (gfx950 kernels).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cattus {

// Tower layout: a board owns 64 pixel slots (board edge <= 8) or 128 (edge 9..11); rows = board * slots + slot.
inline uint32_t tower_slots(uint32_t S) { return S <= 8 ? 64u : 128u; }
constexpr int ROWS_PER_WG = 256; // tower rows per workgroup of the conv kernel: 4 boards of 64 slots, 2 of 128
constexpr int COUT_PER_WG = 64;  // output channels per workgroup of the tower conv kernel

// Activation element of the tuned tower: 2-byte bf16, 4-byte f32, a pair of f16 values (hi, lo) of 2 + 2 bytes
// (the split-precision tower: 128 bytes of a row are [hi of 32 channels | lo of the same 32]), or a single 2-byte f16
// (the single-term f16 tower: the bf16 kernel's layout and MFMA count with 11 significant bits instead of 8).
enum class Act : int { F32 = 0, BF16 = 1, F16S = 2, F16 = 3 };

inline bool act_two_bytes(Act a) { return a == Act::BF16 || a == Act::F16; }
inline int act_bytes(Act a) { return act_two_bytes(a) ? 2 : 4; }  // bytes of one activation in HBM
// Input channels consumed per pipeline stage: one 128-byte LDS row.
inline int act_kc(Act a) { return act_two_bytes(a) ? 64 : 32; }
// Element type of the head kernels (K3-K5) for a tower of type `a`: the f16 towers' heads run in exact f32 (their last
// layer writes plain f32 rows).
inline Act head_act(Act a) { return a == Act::BF16 ? Act::BF16 : Act::F32; }
// The f16 towers: weights pre-scaled per output channel by a power of two, the bias buffer carries the inverse scales,
// activations saturate at 65504 (counted), the last layer writes f32 rows.
inline bool act_f16_family(Act a) { return a == Act::F16S || a == Act::F16; }

// Sets the > 64 KiB dynamic-LDS attribute of every tower kernel variant on the current device; called once per device
// from cattus_hip_create before the first launch.
hipError_t prepare_device();

// ---- K0: bitboard planes -> tensors -------------------------------------------------------
// NHWC tower input [bpad][slots][cpad] (rows >= n, slots >= S*S and channels >= C are zero).
void launch_pack_planes_nhwc(Act act, const uint64_t* planes, uint32_t n, uint32_t bpad, uint32_t C,
                             uint32_t w64, uint32_t S, uint32_t cpad, void* out, hipStream_t st);
// Reference layout: f32 NCHW [batch][C][S*S], rows >= n zero (engine/src/net/mod.rs:121-156).
void launch_planes_to_tensor_nchw(const uint64_t* planes, uint32_t n, uint32_t C, uint32_t w64, uint32_t S,
                                  uint32_t batch, float* out, hipStream_t st);

// ---- K1/K2: 3x3 conv + folded BN (+ residual) + ReLU, MFMA, NHWC -------------------------
// in [bpad][slots][cin], w [9][cout][cin], bias [cout] f32, res/out [bpad][slots][cout], slots = tower_slots(S).
// Requires bpad * slots % 256 == 0, cin % kc == 0, cout % 64 == 0, S <= 11.
// ev_start / ev_stop (optional): events stamped with the kernel's own begin / end time.
// stem (optional): the layer reads the leaves' bitboard planes instead of `in` (K0 fused into the stem conv: the
// loader waves expand the planes straight into LDS).  Requires C <= 32 planes and cin == one 128-byte row.
struct StemInput {
    const uint64_t* planes;  // [n][C][w64]
    uint32_t n, C, w64;
};
// Act::F16S (split precision): rows of in / res / out and of w are f16 pairs interleaved in groups of 32 channels,
// [hi of channels 32g .. 32g+31 | lo of the same 32] per 128 bytes; w is pre-scaled per output channel by a power of
// two, bias is [cout biases | cout inverse scales]; cin counts channels (pairs).
// Act::F16 (single-term f16): rows of f16 values in the bf16 tower's layout, w [9][cout][cin] f16 pre-scaled as above, bias
// [cout biases | cout inverse scales].
// flags & CONV_OUT_F32 (F16S / F16 only): out is written as plain f32 [row][cout] (last tower layer, read by the f32 head kernels).
// flags & CONV_W_FRAG (F16S only): w is in MFMA fragment order, [cout / 32][stage][hi | lo][lane][8 f16] with
// stage = ((chunk * 3 + dy) * 3 + dx) * 2 + k-half (split_frag_index below): the kernel that keeps the weights in a register
// ring instead of LDS.  Non-stem layers then need cin >= 64.
// flags & CONV_WINO_IN (with CONV_OUT_F32): the rows feed a Winograd tower (K1w): values are capped at WINO_ACT_MAX and counted in
// `sat` beyond it -- a transformed input B^T d B is a signed sum of four activations, so no |V| can leave the f16 range then and the
// Winograd kernel's transform needs neither a clamp nor a range check of its own.
constexpr int CONV_OUT_F32 = 1, CONV_W_FRAG = 2, CONV_WINO_IN = 4;
constexpr float WINO_ACT_MAX = 16376.0f;  // 65504 / 4, exactly
// Element index of weight (tap t, output channel co, input channel ci, part 0 = hi / 1 = lo) in fragment order.
inline size_t split_frag_index(uint32_t t, uint32_t co, uint32_t ci, uint32_t part, uint32_t cin_pad) {
    const uint32_t nst = cin_pad / 32 * 18, ch = ci >> 5, k = (ci >> 4) & 1, h = (ci >> 3) & 1, e = ci & 7;
    const uint32_t stage = ((ch * 3 + t / 3) * 3 + t % 3) * 2 + k, lane = h * 32 + (co & 31);
    return ((((size_t)(co >> 5) * nst + stage) * 2 + part) * 64 + lane) * 8 + e;
}
// Per-evaluator launch options (they used to be process-wide switches: a second evaluator must not re-tile the first).
struct ConvOpts {
    int cb = 0;   // forces the conv kernel's tile: 1 = 256 rows x 32 couts, 2 = 256 rows x 64 couts; 0 = chosen by grid size
    int pbw = 0;  // 2: never / 1: whenever the tile is 32 couts -- the 128-row workgroup; 0 = chosen by grid size
    // F16S / F16: device counter of activation values that were clamped to 65504 (atomic add on the clamp path only);
    // must be non-null for those towers
    unsigned* saturated = nullptr;
};
void launch_conv3x3_mfma(Act act, const void* in, const void* w, const float* bias, const void* res, void* out,
                         uint32_t bpad, uint32_t cin, uint32_t cout, uint32_t S, hipStream_t st,
                         hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, const StemInput* stem = nullptr,
                         int flags = 0, const ConvOpts& opts = ConvOpts());

// ---- K1w: the split-precision conv in Winograd F(2x2, 3x3) form (kernels_wino.hip) ----
// 8x8 boards only; in / res / out are PLAIN F32 rows [row][channel] (what CONV_OUT_F32 writes: a tower in this form keeps f32
// activations between its layers); wu = G g G^T as (hi, lo) f16 pairs pre-scaled per output channel, in fragment order
// [cout / 32][(cin / 16) k-steps x 16 frequencies][hi | lo][lane][8 f16] (wino_frag_index); bias = [cout biases | cout inverse
// scales] of THAT scaling; res may be out (a block's output over its skip rows); outputs are capped at WINO_ACT_MAX and sat counts
// the threads that wrote the cap.
// the kernel's weight ring reads WINO_RING_STAGES stages (of 2,048 B) past a cout block's end: wu is allocated with that much behind it
constexpr int WINO_RING_STAGES = 8;
bool wino_supported(uint32_t bpad, uint32_t cin, uint32_t cout, uint32_t S);
inline size_t wino_frag_index(uint32_t f, uint32_t co, uint32_t ci, uint32_t part, uint32_t cin_pad) {
    const uint32_t nst = cin_pad / 16 * 16, stage = (ci >> 4) * 16 + f, lane = ((ci >> 3) & 1) * 32 + (co & 31);
    return ((((size_t)(co >> 5) * nst + stage) * 2 + part) * 64 + lane) * 8 + (ci & 7);
}
void launch_conv3x3_wino(const float* in, const void* wu, const float* bias, const float* res, float* out, uint32_t bpad, uint32_t cin,
                         uint32_t cout, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop, unsigned* sat);
hipError_t prepare_wino();  // its dynamic-LDS opt-in; called by prepare_device()

// ---- K1w4: the same layer with 4 frequencies x (2 tile blocks x 2 cout blocks) per wave (kernels_wino4.hip) ----
// Same arguments, same U buffer, same bits as launch_conv3x3_wino; workgroup = 4 boards x 64 couts (bpad % 4 == 0, cout % 64 == 0):
// half of K1w's U bytes per CU, the transformed input made in registers.  The evaluator prefers it wherever it is supported.
bool wino4_supported(uint32_t bpad, uint32_t cin, uint32_t cout, uint32_t S);
void launch_conv3x3_wino4(const float* in, const void* wu, const float* bias, const float* res, float* out, uint32_t bpad, uint32_t cin,
                          uint32_t cout, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop, unsigned* sat);
hipError_t prepare_wino4();
// ---- K1w8: the same layer on EIGHT waves of 128 accumulator registers, two per SIMD (kernels_wino8.hip); same arguments, U and bits ----
bool wino8_supported(uint32_t bpad, uint32_t cin, uint32_t cout, uint32_t S);
void launch_conv3x3_wino8(const float* in, const void* wu, const float* bias, const float* res, float* out, uint32_t bpad, uint32_t cin,
                          uint32_t cout, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop, unsigned* sat);
hipError_t prepare_wino8();
// The whole Winograd tower in one launch (tower_wino4_kernel): a table of its layers in device memory (an even number: residual blocks,
// the first layer of a block without skip rows, the second with; res may equal out), `ready` = nlayers x (bpad / 4) zeroed counters, `err` a zeroed word the launch sets when a hand-off wait ran out of
// `spin_budget` polls (the outputs are then invalid: run the batch on the per-layer launches).  Only while wino4_tower_fits: every
// workgroup resident at once (one per CU), and never two such launches on one device at a time.
struct Wino4TowerLayer {
    const float* in;
    const void* wu;
    const float* bias;
    const float* res;
    float* out;
};
bool wino4_tower_fits(uint32_t bpad, uint32_t filters, uint32_t cus);
void launch_tower_wino4(const Wino4TowerLayer* d_layers, uint32_t nlayers, unsigned* ready, unsigned* err, unsigned* sat, uint32_t bpad, uint32_t filters,
                        uint32_t spin_budget, uint32_t cus, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop);

// Diagnostic: one launch of nothing but back-to-back MFMAs of the tower's kind (F16S: f16, BF16, F32: 32x32x2 f32), four
// waves on each of `cus` workgroups, iters x 4 MFMAs per wave; `out` holds cus * 256 floats.  Returns the launch's FLOPs.
double launch_mfma_sustain(Act act, int cus, int iters, float* out, hipStream_t st);

// ---- K1 resident: whole tower of a network with <= 64 (padded) filters in one launch, bf16 (see kernels.hip) ----
struct Tower64Layer {
    const void* w;      // [9][64][64] bf16 (stem: input channels padded to 64)
    const float* bias;  // [64]
    int res;            // 1: add the block input (second conv of a residual block)
    int pad_;
};
struct Tower64Args {
    const uint64_t* planes;      // [n][C][w64] bitboards
    const Tower64Layer* layers;  // device array [nlayers]: stem, then (conv1, conv2) per block
    void* out;                   // optional: [rows][64] bf16 tower output, rows = boards * tower_slots(S)
    uint32_t n, C, w64, S, nlayers;
    // optional, fused K3: the two 1x1 head convs (+ folded BN + ReLU) on the resident tower output, written in the
    // layout the head FC kernels read (HeadsMfma::hv)
    const void* head_w;   // [32][64] bf16, value rows first, rows >= ocn zero
    const float* head_b;  // [32]: entries >= ocn are not used
    void* hv;             // head activations in the fragment-packed layout of HeadsMfma::hv, bf16
    uint32_t hv_pol, kvp, kpp, vhc, ocn;  // hv_pol: element offset of the policy part
};
// rows % 256 == 0; ch = 2: 128 rows per workgroup, ch = 4: one 64-slot board per workgroup (small batches; 128-slot
// boards run as ch = 2).  layer_steps: with ch = 4 and boards of <= 63 pixels, one barrier per layer instead of three
// (two layers of weights in LDS); same results.
void launch_tower64(const Tower64Args& args, uint32_t rows, int ch, bool layer_steps, hipStream_t st,
                    hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);

// ---- K1rs resident, split precision: the whole tower of a <= 64-filter f16x2 network in one launch (see kernels.hip) ----
struct Tower64SplitLayer {
    const void* wf;     // fragment-ordered (hi, lo) weights (CONV_W_FRAG layout, cout padded to 64)
    const float* bias;  // [64 biases | 64 inverse scales]
    int res;            // 1: add the block input (second conv of a residual block)
    int nch;            // 32-channel chunks of the layer's input: 1 for the stem (planes), 2 otherwise
};
struct Tower64SplitArgs {
    const uint64_t* planes;            // [n][C][w64] bitboards, C <= 32
    const Tower64SplitLayer* layers;   // device array [nlayers]: stem, then (conv1, conv2) per block
    const float* bias_all;             // [nlayers][64 biases | 64 inverse scales]: every layer's `bias`, contiguous
    float* out;                        // optional: [rows][64] f32, the tower output, rows = boards * tower_slots(S)
    unsigned* sat;                     // the evaluator's saturation counter (ConvOpts::saturated)
    uint32_t n, C, w64, S, nlayers;
    // optional, fused K3: the two 1x1 head convs (+ folded BN + ReLU) in f32 on the resident output, written in the layout the
    // head FC kernels read (HeadsMfma::hv); head_w [32][64] f32 (value rows first, rows >= ocn zero), head_b [32]
    const float* head_w;
    const float* head_b;
    float* hv;
    uint32_t hv_pol, kvp, kpp, vhc, ocn;  // hv_pol: element offset of the policy part
};
constexpr int T64S_MAX_LAYERS = 81;  // the bias table of all layers lives in LDS (512 B per layer)
// rows % 256 == 0 (whole boards); one workgroup per board
// shape (64-slot boards): 1 = one board per workgroup (four one-tile waves), 2 = two boards per workgroup (a wave holds a
// board's 64 pixels x 32 couts), 9 = shape 1 with a 9-stage weight ring and two stages of pixel look-ahead (A/B); 0 = by grid size
void launch_tower64_split(const Tower64SplitArgs& args, uint32_t rows, int shape, hipStream_t st, hipEvent_t ev_start = nullptr,
                          hipEvent_t ev_stop = nullptr);
hipError_t prepare_tower64_split();  // its dynamic-LDS opt-in; called by prepare_device()

// Generic f32 NCHW direct conv for shapes the MFMA kernel does not cover (any S <= 11, any C).
// in [b][cin][hw], w [9][cout][cin], out [b][cout][hw]; same summation order as the MFMA f32 kernel.
void launch_conv3x3_generic(const float* in, const float* w, const float* bias, const float* res, float* out,
                            uint32_t b, uint32_t cin, uint32_t cout, uint32_t S, hipStream_t st,
                            hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);

// ---- K3-K5: heads on MFMA (tuned tower layout) ----------------------------------------------
// All matrices are in the tower element type `act` with K contiguous and zero-padded:
//   conv_w [32][F] (value rows, then policy rows, rest zero), conv_b [32] f32 (entries >= vhc+phc zero),
//   hv: value part [leaf][kvp] at element 0, policy part [leaf][kpp] at element hv_leaves * kvp, both stored in MFMA
//   fragment order (below; pad columns stay zero), w1 [128][kvp] and wp [round32(M)][kpp] in the same order,
//   b1 [128] f32, bp [M] f32, policy [nb][M] f32;
//   fragment order of a [rows][K] matrix (rows padded to 32): element (row, k) lives at
//     ((((row / 32) * (K / KSTEP) + k / KSTEP) * 2 + (k % KSTEP) / HALF) * 32 + row % 32) * HALF + k % HALF,
//   KSTEP = 32 bytes of k, HALF = 16 bytes: the 64 lanes' 16-byte operand pieces of one 32-row tile and one MFMA k-step
//   are one contiguous KiB, so a wave's fragment load is one fully coalesced instruction (no LDS staging, no barrier);
//   h1 [nb][128] f32 is scratch of the stand-alone FC1 (the fused FC launch keeps the hidden units in LDS).
//   kvp, kpp and F are multiples of 16 elements (the k walk reads whole 16-byte pieces).
struct HeadsMfma {
    const void* conv_w;
    const float* conv_b;
    void* hv;
    const void* w1;
    const float* b1;
    float* h1;
    const void* wp;
    const float* bp;
    float* policy;
    const float *w2, *b2;  // value FC2 [128], [1]
    float* value;          // [nb], tanh applied
    uint32_t hw, vhc, phc, kvp, kpp, M;
    uint32_t hv_leaves;    // leaf capacity of hv, a multiple of 32
    uint32_t slots;  // pixel slots per board of the tower (tower_slots)
};
void launch_heads_mfma(Act act, const void* tower, uint32_t nb, uint32_t F, const HeadsMfma& hd, hipStream_t st);

// ---- K3-K5: heads, SIMT f32 (generic tower layout), same term order as the MFMA path ------
// Tower output addressing: element (b, k, p) at  x[b*sb + k*sk + p*sp]  (elements of type `act`).
struct TowerView {
    const void* x;
    Act act;
    uint32_t sb, sk, sp;
};
// hv[b][(oc)*hw + p] = relu(sum_k w[oc][k]*x(b,k,p) + bias[oc]) for oc < ocn (value rows first, then policy)
void launch_head_conv1x1(TowerView x, const float* w, const float* bias, uint32_t b, uint32_t F, uint32_t ocn,
                         uint32_t hw, float* hv, hipStream_t st);
// h1[b][j] = relu(sum_k w1t[k][j]*hv[b*hv_stride + k] + b1[j]), j < 128
void launch_value_fc1(const float* hv, uint32_t hv_stride, const float* w1t, const float* b1, uint32_t b, uint32_t K,
                      float* h1, hipStream_t st);
// value[b] = tanh(sum_j w2[j]*h1[b][j] + b2)
void launch_value_fc2_tanh(const float* h1, const float* w2, const float* b2, uint32_t b, float* value,
                           hipStream_t st);
// policy[b][m] = scrub(sum_k wpt[k][m]*hv[b*hv_stride + off + k] + bp[m])
// Softmax over each leaf's legal moves (net/mod.rs:100-119): idx [b][L] policy indices, cnt [b] <= L, probs [b][L].
int launch_legal_softmax(const float* policy, uint32_t M, const uint16_t* idx, const uint16_t* cnt, uint32_t L, uint32_t b,
                         float* probs, hipStream_t st);
void launch_policy_fc(const float* hv, uint32_t hv_stride, uint32_t off, const float* wpt, const float* bp, uint32_t b,
                      uint32_t K, uint32_t M, float* policy, hipStream_t st);
// One dense layer of SimpleTwoHeadedModel (training/cattus_train/net_utils.py:92-121), f32, the term order of every
// other dot product here: y[b][n] = epi(sum_k wt[k][n] * x[b * x_stride + k] + bias[n]); epi 0: non-finite -> f32::MIN
// (policy logits), 1: ReLU, 2: tanh.  K * 32 bytes of LDS (K <= 2048).
void launch_dense(const float* x, uint32_t x_stride, const float* wt, const float* bias, uint32_t b, uint32_t K, uint32_t N,
                  float* y, int epi, hipStream_t st);

}  // namespace cattus
