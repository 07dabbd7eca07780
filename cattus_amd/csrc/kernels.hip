// gfx950 (MI355X / CDNA4) kernels of the Cattus leaf evaluator.
//
//   K0  planes_to_tensor_nchw[64] / pack_planes_nhwc    bitboards -> tensors (HBM-bound); inside the forward pass the
//                                                       planes are expanded straight into LDS by K1 / K1r instead
//   K1  conv3x3_mfma_v2<T, HAS_RES, BIG, STEM>          3x3 conv + folded BN (+skip) + ReLU, one launch per layer (MFMA-bound), bf16 / f32
//   K1s conv3x3_splitw<HAS_RES, BIG, STEM, CB>          the same layer of the split-precision tower (dtype f16x2, the default):
//                                                       operands as pairs of f16 values, three f16 MFMA terms per product;
//                                                       weights from L2 into a register ring, activations through LDS
//       conv3x3_split<HAS_RES, BIG, STEM, CB>           its first form, both operands through LDS (CATTUS_SPLIT_W=0; bit-identical)
//   K1r tower64_lds<CH, BIG>                            whole tower of a <= 64-filter bf16 network in one launch, activations in LDS
//   K1g conv3x3_generic                                 same arithmetic, any shape, SIMT f32 (checker; wide heads)
//   K3  head_conv, K4 + K5 head_fc_pair                 1x1 head convs; value FC1 + FC2 + tanh and policy FC (MFMA)
//       head_conv1x1, value_fc1, value_fc2_tanh, policy_fc   the same on the generic path (SIMT f32)
//
// Arithmetic the kernels reproduce: ConvNetV1.forward in eval mode
// (reference: training/cattus_train/net_utils.py:4-89) on the tensor planes_to_tensor builds
// (reference: engine/src/net/mod.rs:121-156).  In f32 mode every output is an in-order fmaf
// chain (v_mfma_f32_32x32x2_f32 chains k in issue order), in the order documented in DESIGN.md,
// so results are bit-identical across batch sizes, batch compositions and to the CPU oracle.  The split-precision and
// bf16 towers are bit-reproducible and batch-independent too, within the error bounds stated in DESIGN.md section 4.
#include "kernels.h"
#include "device_common.h"

#include <hip/hip_ext.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace cattus {


// ------------------------------------------------------------------------------------------
// K0: plane expansion
// ------------------------------------------------------------------------------------------

// One workgroup per board: the board's C*w64 plane words are staged in LDS, then every thread
// writes 16-byte channel vectors so that a wave stores 1 KiB contiguous.
template <typename T>
__global__ void __launch_bounds__(256) pack_planes_nhwc_kernel(const uint64_t* __restrict__ planes, uint32_t n,
                                                               uint32_t C, uint32_t w64, uint32_t hw, uint32_t cpad,
                                                               uint32_t slots, T* __restrict__ out) {
    constexpr uint32_t VEC = 16 / sizeof(T);
    __shared__ uint64_t pl[128];
    const uint32_t b = blockIdx.x;
    const uint32_t words = C * w64;
    for (uint32_t i = threadIdx.x; i < words; i += 256) pl[i] = b < n ? planes[(size_t)b * words + i] : 0ull;
    __syncthreads();
    const uint32_t groups = cpad / VEC;
    T* ob = out + (size_t)b * slots * cpad;
    for (uint32_t v = threadIdx.x; v < slots * groups; v += 256) {
        const uint32_t q = v / groups, c0 = (v % groups) * VEC;
        T vals[VEC];
#pragma unroll
        for (uint32_t i = 0; i < VEC; i++) {
            const uint32_t c = c0 + i;
            uint32_t bit = 0;
            if (c < C && q < hw) bit = (uint32_t)(pl[c * w64 + (q >> 6)] >> (q & 63)) & 1u;
            vals[i] = bit ? (T)1.0f : (T)0.0f;
        }
        *reinterpret_cast<f32x4*>(ob + (size_t)q * cpad + c0) = *reinterpret_cast<f32x4*>(vals);
    }
}

// Reference layout (f32 NCHW).  Each thread produces four consecutive floats of the flat output.
__global__ void __launch_bounds__(256) planes_to_tensor_nchw_kernel(const uint64_t* __restrict__ planes, uint32_t n,
                                                                    uint32_t C, uint32_t w64, uint32_t hw,
                                                                    uint64_t total, float* __restrict__ out) {
    const uint64_t per_board = (uint64_t)C * hw;
    for (uint64_t e0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4; e0 < total; e0 += (uint64_t)gridDim.x * 1024) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint64_t e = e0 + i;
            float x = 0.0f;
            if (e < total) {
                const uint64_t b = e / per_board;
                const uint32_t rem = (uint32_t)(e - b * per_board);
                const uint32_t c = rem / hw, p = rem - c * hw;
                if (b < n) x = (planes[(b * C + c) * w64 + (p >> 6)] >> (p & 63)) & 1ull ? 1.0f : 0.0f;
            }
            v[i] = x;
        }
        if (e0 + 4 <= total) {
            *reinterpret_cast<f32x4*>(out + e0) = *reinterpret_cast<f32x4*>(v);
        } else {
            for (int i = 0; i < 4 && e0 + i < total; i++) out[e0 + i] = v[i];
        }
    }
}

void launch_pack_planes_nhwc(Act act, const uint64_t* planes, uint32_t n, uint32_t bpad, uint32_t C, uint32_t w64,
                             uint32_t S, uint32_t cpad, void* out, hipStream_t st) {
    const uint32_t slots = tower_slots(S);
    if (act == Act::BF16)
        hipLaunchKernelGGL(pack_planes_nhwc_kernel<__bf16>, dim3(bpad), dim3(256), 0, st, planes, n, C, w64, S * S, cpad,
                           slots, (__bf16*)out);
    else
        hipLaunchKernelGGL(pack_planes_nhwc_kernel<float>, dim3(bpad), dim3(256), 0, st, planes, n, C, w64, S * S, cpad,
                           slots, (float*)out);
}

// 8x8 boards: one plane = 64 floats = 256 B.  A thread turns one nibble of the plane word into 4 floats
// (one 16-byte store); 16 consecutive threads cover one plane, a wave stores 1 KiB contiguous.
__global__ void __launch_bounds__(256) planes_to_tensor_nchw64_kernel(const uint64_t* __restrict__ planes, uint64_t n_planes,
                                                                      uint64_t total_planes, float* __restrict__ out) {
    // Grid-stride over 16-byte output pieces; 4 pieces per thread and trip, 8 blocks per CU: the store stream
    // is what bounds this kernel, and fewer, longer store streams measured 5.8 TB/s against 4.7 TB/s for
    // one piece per trip on 16 blocks per CU (contiguous per-block ranges and more unrolling were slower).
    constexpr int UNROLL = 4;
    const uint64_t stride = (uint64_t)gridDim.x * 256, total = total_planes * 16;
    for (uint64_t t0 = (uint64_t)blockIdx.x * 256 + threadIdx.x; t0 < total; t0 += stride * UNROLL) {
        uint32_t bits[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const uint64_t t = t0 + u * stride;
            const uint64_t pl = t >> 4;
            const uint32_t nib = (uint32_t)(t & 15);
            bits[u] = (t < total && pl < n_planes) ? (uint32_t)(planes[pl] >> (nib * 4)) & 0xfu : 0u;
        }
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const uint64_t t = t0 + u * stride;
            if (t >= total) break;
            f32x4 a;
#pragma unroll
            for (int i = 0; i < 4; i++) a[i] = (bits[u] >> i) & 1u ? 1.0f : 0.0f;
            __builtin_nontemporal_store(a, reinterpret_cast<f32x4*>(out + t * 4));
        }
    }
}

void launch_planes_to_tensor_nchw(const uint64_t* planes, uint32_t n, uint32_t C, uint32_t w64, uint32_t S,
                                  uint32_t batch, float* out, hipStream_t st) {
    if (S == 8 && w64 == 1) {
        const uint64_t total_planes = (uint64_t)batch * C;
        uint64_t blocks = (total_planes * 16 + 255) / 256;
        if (blocks > 256 * 8) blocks = 256 * 8;
        if (blocks == 0) return;
        hipLaunchKernelGGL(planes_to_tensor_nchw64_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, planes, (uint64_t)n * C,
                           total_planes, out);
        return;
    }
    const uint64_t total = (uint64_t)batch * C * S * S;
    uint64_t blocks = (total + 1023) / 1024;
    if (blocks > 2048) blocks = 2048;  // 256 CUs x 8 blocks, grid-stride beyond that
    if (blocks == 0) return;
    hipLaunchKernelGGL(planes_to_tensor_nchw_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, planes, n, C, w64, S * S,
                       total, out);
}

// ------------------------------------------------------------------------------------------
// K1: 3x3 conv as implicit GEMM on MFMA
// ------------------------------------------------------------------------------------------
//
// out^T[cout][pixel] = sum_{chunk, tap, k} W[tap][cout][chunk*KC + k] * in[pixel + tap][chunk*KC + k]
//
// Weights are the MFMA A operand (row = cout), activations the B operand (col = pixel), so each lane ends
// up with 4 consecutive couts of one pixel.  K is walked as (chunk of one 128-byte row = KC channels) x
// (9 taps); the activation chunk of a workgroup's 256 pixel rows is loaded once per chunk and re-read by
// all 9 taps with shifted pixel rows (out-of-board lanes read a zero row), so activations cross L2->LDS
// once, not nine times.  LDS rows are XOR-swizzled at 16-byte granularity (chunk ^ ((row>>1)&7)) on the
// global SOURCE side and on the ds_read side; the LDS-DMA destination stays lane-linear.
//
// Tower layout: a board owns 64 pixel slots (board edge <= 8) or 128 (edge 9..11); a workgroup always
// covers 256 consecutive tower rows = 4 boards of 64 slots or 2 boards of 128.


// ---- diagnostic build only (-DCATTUS_STAMPS): per-wave cycle stamps of the v2 tower kernel ------
#ifdef CATTUS_STAMPS
__device__ unsigned long long g_stamps[512 * 8 * 8];
#define STAMP_DECL unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define STAMP(i) st_[i] = __builtin_amdgcn_s_memtime()
#define STAMP_RT(i) st_[i] = __builtin_amdgcn_s_memrealtime()
#define STAMP_ACC_BEGIN unsigned long long acc_t0_ = __builtin_amdgcn_s_memtime()
#define STAMP_ACC_END(i) st_[i] += __builtin_amdgcn_s_memtime() - acc_t0_
#define STAMP_FLUSH(wave)                                                          \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 256)                               \
        for (int q_ = 0; q_ < 8; q_++) g_stamps[(blockIdx.x * 16 + (wave)) * 8 + q_] = st_[q_]
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_RT(i)
#define STAMP_ACC_BEGIN
#define STAMP_ACC_END(i)
#define STAMP_FLUSH(wave)
#endif

// Workgroup = 8 waves = 256 tower rows x 64 output channels.  Waves 0-3 are MFMA consumers: wave w owns
// rows w*64 .. w*64+63 (a whole 64-slot board, or half of a 128-slot board) = a 64(cout) x 64(pixel)
// tile = 2x2 MFMA 32x32 accumulators; waves 4-7 are loaders that do nothing but LDS-DMA.  One barrier per (chunk, kernel row): a step covers the
// three taps dx = -1,0,+1 of one dy, i.e. 48 MFMAs (bf16) per consumer wave between barriers, and
// a 24 KiB weight slab [3 taps][64 cout][128 B].  Loaders run two steps ahead (3-slab ring) and
// fetch the next activation chunk during the first two steps of the current one.  Consumers issue
// no global memory instruction and almost no address arithmetic inside the loop.
//
// The fmaf-chain order per output element is unchanged: chunk -> tap 0..8 -> k (0,4,1,5,2,6,3,7).
// LDS map: weight ring first so that (ring slot, tap) offsets fold into ds_read immediates.
constexpr int V2_SLAB = 3 * 8192;
constexpr int V2_LDS_W = 0;                           // 3 x 24 KiB
constexpr int V2_LDS_ZERO = 3 * V2_SLAB;              // 256 B of zeros (a whole bank row: device_common.h, ZAREA_SP)
constexpr int V2_LDS_ACT = V2_LDS_ZERO + 256;         // 2 x 32 KiB
constexpr int V2_LDS_TOTAL = V2_LDS_ACT + 2 * 32768;  // 139520 B
static_assert(V2_LDS_ZERO % 256 == 0 && V2_LDS_ACT % 256 == 0, "off-board reads keep their row's banks only if both are 256-B aligned");


#ifndef CATTUS_NLOAD
#define CATTUS_NLOAD 4  // loader waves per workgroup (the kernels assert 4)
#endif
constexpr int NLOAD = CATTUS_NLOAD;
constexpr int WPL = 24 / NLOAD;  // weight pieces per loader wave and step
constexpr int APL = 16 / NLOAD;  // activation pieces per loader wave and half-chunk

// Stem input of the STEM variant: the leaves' bitboard planes (K0 fused: the loader waves expand them into the
// activation chunk in LDS instead of fetching a packed tensor).  Empty for the other layers, so that their
// kernel arguments stay within the 16 preloaded dwords.
template <bool STEM>
struct StemPlanes {};
template <>
struct StemPlanes<true> {
    const uint64_t* planes;  // [n][C][w64]
    uint32_t n, C, w64;
};

// BIG: 128 pixel slots per board (two consumer waves per board).
// STEM: the layer's input is the bitboard planes (C <= 32 planes, one 128-byte chunk): `in` is not read.
// CB: 32-cout blocks per workgroup.  2 = the 256 rows x 64 couts tile described above.  1 = 256 rows x 32 couts
//     (consumer wave: 1 x 2 tiles): twice the workgroups for the same layer, used while the full-size grid would
//     leave more than half of the CUs empty (batches of <= 128 leaves on a 256-filter net).  Same MFMA shape and
//     k order per output element, so the result does not depend on which of the two ran.
// PBW: 32-pixel blocks per consumer wave.  2: the workgroup covers 256 tower rows; 1 (with CB = 1): 128 rows, 32 pixels per
//     wave -- twice the workgroups and half the MFMA chain per wave, for grids that would otherwise leave CUs empty.
template <typename T, bool HAS_RES, bool BIG, bool STEM = false, int CB = 2, int PBW = 2>
__global__ void __launch_bounds__(256 + 64 * NLOAD, (4 + NLOAD) / 4)
    conv3x3_mfma_v2_kernel(const T* __restrict__ in, const T* __restrict__ w, const float* __restrict__ bias,
                           const T* __restrict__ res, T* __restrict__ out, unsigned* __restrict__ sat, int cin, int cout, int S,
                           int flags, StemPlanes<STEM> sp) {
    static_assert(NLOAD == 4, "the stem expansion, the 32-cout tile and the piece counts below assume four loader waves");
    // single-term f16 tower: weights pre-scaled per output channel (`bias` = [cout biases | cout inverse scales]), activations
    // clamped at 65504 and counted in `sat`, flags & CONV_OUT_F32 = plain f32 output rows; `sat` / `flags` are not read otherwise
    constexpr bool F16T = std::is_same<T, _Float16>::value;
    constexpr int KC = 128 / (int)sizeof(T);
    constexpr int CPW = 32 * CB;          // output channels of this workgroup
    constexpr int WPLC = WPL * CB / 2;    // weight pieces per loader wave and step
    static_assert(PBW == 2 || (PBW == 1 && CB == 1), "the 128-row workgroup exists for the 32-cout tile only");
    constexpr int RW = 128 * PBW;         // tower rows of this workgroup
    constexpr int PXW = 32 * PBW;         // pixels (rows) per consumer wave
    constexpr int NHALF = PBW;            // an activation chunk is NHALF x 16 pieces of 1 KiB
    typedef typename Mfma<T>::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const bool is_loader = wave >= 4;  // waves 4 .. 4+NLOAD-1
    STAMP_DECL;
    STAMP(0);
    STAMP_RT(5);

    const int nblk = gridDim.x, ncb = cout / CPW;
    int logical = blockIdx.x;
    if ((nblk & 7) == 0) logical = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
    const int cout0 = (logical % ncb) * CPW;
    const int row0 = (logical / ncb) * RW;  // first tower row (board * slots + pixel slot) of this workgroup

    if (tid < 16) reinterpret_cast<f32x4*>(smem + V2_LDS_ZERO)[tid] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the zero row is written before the first barrier

    const int nch = cin / KC;
    const int T_total = nch * 3;
    const uint32_t row_bytes = (uint32_t)cin * sizeof(T);

    if (is_loader) {
        // ================================ loader waves ================================
        const int lw = wave - 4;
        const int prow = lane >> 3, pslot = lane & 7;
        uint32_t off_w[WPLC], off_a[NHALF][APL];
        int dst_w[WPLC], dst_a[NHALF][APL];
#pragma unroll
        for (int i = 0; i < WPLC; i++) {
            const int pid = lw * WPLC + i;              // 0..12*CB-1: tap_i = pid / (4*CB), 8 cout rows each
            const int tap_i = pid / (4 * CB), within = pid - tap_i * (4 * CB), row = within * 8 + prow;
            const int c = pslot ^ ((row >> 1) & 7);
            off_w[i] = ((uint32_t)(tap_i * cout + row)) * row_bytes + c * 16;
            dst_w[i] = tap_i * 8192 + within * 1024;    // a tap's rows start 8 KiB apart whatever CB is
        }
#pragma unroll
        for (int g = 0; g < NHALF; g++)
#pragma unroll
            for (int i = 0; i < APL; i++) {
                const int id = g * 16 + lw * APL + i;   // 0..31 (0..15 at 128 rows), 8 rows each
                const int row = id * 8 + prow;
                const int c = pslot ^ ((row >> 1) & 7);
                off_a[g][i] = (uint32_t)row * row_bytes + c * 16;
                dst_a[g][i] = id * 1024;
            }
        const char* wbase0 = reinterpret_cast<const char*>(w) + (size_t)cout0 * row_bytes;
        const char* abase0 = reinterpret_cast<const char*>(in) + (size_t)row0 * row_bytes;
        auto issue_w = [&](int t) {  // weight slab of step t -> ring slot t % 3
            const int ch = t / 3, g = t - ch * 3;
            const char* src = wbase0 + (size_t)(g * 3) * cout * row_bytes + (size_t)ch * 128;
            char* dst = smem + V2_LDS_W + (t % 3) * V2_SLAB;
#pragma unroll
            for (int i = 0; i < WPLC; i++) glds16(src + off_w[i], dst + dst_w[i]);
        };
        auto issue_a = [&](int ch, int g) {  // half g of activation chunk ch -> buffer ch & 1
            const char* src = abase0 + (size_t)ch * 128;
            char* dst = smem + V2_LDS_ACT + (ch & 1) * 32768;
#pragma unroll
            for (int i = 0; i < APL; i++) glds16(src + off_a[g][i], dst + dst_a[g][i]);
        };

        if constexpr (STEM) {
            // K0 fused: this thread expands pixel row (lw * 64 + lane) of the workgroup's 256 rows.  All plane words
            // of its board first (one round trip), the first weight slabs behind them, then the row: channel c of
            // the row is 1.0 where plane c has the pixel's bit; channel slot s lands in LDS slot s ^ swizzle(row),
            // where the LDS-DMA path would have put it.
            constexpr int VEC = 16 / (int)sizeof(T);   // channels per 16-byte slot
            constexpr int MAXC = 32;
            const int row = lw * 64 + lane;
            const uint32_t grow = (uint32_t)(row0 + row), slots = BIG ? 128u : 64u;
            const uint32_t board = grow / slots, px = grow % slots;
            const bool live = row < RW && board < sp.n && (int)px < S * S;
            typedef const __attribute__((address_space(1))) uint64_t* gu64p;
            const gu64p pl = (gu64p)(sp.planes + (size_t)(live ? board : 0) * sp.C * sp.w64 + (live ? (px >> 6) : 0));
            uint64_t words[MAXC];
#pragma unroll
            for (int c = 0; c < MAXC; c++) words[c] = (live && (uint32_t)c < sp.C) ? pl[(size_t)c * sp.w64] : 0ull;
            issue_w(0);
            issue_w(1);
            char* dstrow = smem + V2_LDS_ACT + row * 128;
            const int swz = (row >> 1) & 7;
#pragma unroll
            for (int sl = 0; sl < 8; sl++) {
                T vals[VEC];
#pragma unroll
                for (int i = 0; i < VEC; i++) {
                    const int c = sl * VEC + i;
                    vals[i] = (c < MAXC && ((words[c < MAXC ? c : 0] >> (px & 63)) & 1ull)) ? (T)1.0f : (T)0.0f;
                }
                if (row < RW) *reinterpret_cast<f32x4*>(dstrow + ((sl ^ swz) << 4)) = *reinterpret_cast<f32x4*>(vals);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the rows are written before the first barrier
        } else {
            issue_a(0, 0);
            if constexpr (NHALF == 2) issue_a(0, 1);
            issue_w(0);
            issue_w(1);
        }
        int pending = WPLC;  // loads issued after the data of the upcoming step
        for (int t = 0; t < T_total; t++) {
#ifdef CATTUS_STAMPS
            {  // diagnostic build: time the wait for the data (slot 4) and the wait at the barrier (slot 7) apart
                const unsigned long long t0_ = __builtin_amdgcn_s_memtime();
                if (pending == WPLC + APL) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPLC + APL) : "memory");
                else if (pending == WPLC) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WPLC) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned long long t1_ = __builtin_amdgcn_s_memtime();
                asm volatile("s_barrier" ::: "memory");
                st_[4] += t1_ - t0_;
                st_[7] += __builtin_amdgcn_s_memtime() - t1_;
            }
#else
            if (pending == WPLC + APL) wait_vm_barrier<WPLC + APL>();
            else if (pending == WPLC) wait_vm_barrier<WPLC>();
            else wait_vm_barrier<0>();
#endif
            if (t == 0) STAMP(1);
            pending = 0;
            const int ch = t / 3, g = t - ch * 3;
            if (t + 2 < T_total) {
                issue_w(t + 2);
                pending += WPLC;
            }
            if (g < NHALF && ch + 1 < nch) {
                issue_a(ch + 1, g);
                pending += APL;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(2);
        STAMP(3);
        STAMP_RT(6);
        STAMP_FLUSH(wave);
        return;
    }

    // ================================ consumer waves ================================
#ifdef CATTUS_CONSUMER_PRIO
    __builtin_amdgcn_s_setprio(1);  // MFMA waves win issue arbitration against the loader wave on their SIMD
#endif
    const int r = lane & 31, h = lane >> 5;
    // pixel slot of this lane's two output columns within its board, and the LDS offset of the board's rows
    constexpr int SLOTS = BIG ? 128 : 64, WPB = SLOTS / PXW;  // consumer waves per board
    const int pslot0 = (wave % WPB) * PXW;
    const int board_lds = (wave / WPB) * SLOTS * 128;
    int ph[PBW], pw[PBW];
    bool pvalid[PBW];
#pragma unroll
    for (int pb = 0; pb < PBW; pb++) {
        const int p = pslot0 + pb * 32 + r;
        ph[pb] = p / S;
        pw[pb] = p - ph[pb] * S;
        pvalid[pb] = p < S * S;
    }
    // weight fragment address per (k-slice, cout block): loop-invariant; ring slot and tap are immediates
    int aaddr[4][CB];
#pragma unroll
    for (int cb = 0; cb < CB; cb++) {
        const int row = cb * 32 + r;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) aaddr[ks][cb] = V2_LDS_W + row * 128 + (((ks * 2 + h) ^ ((row >> 1) & 7)) << 4);
    }

    f32x16 acc[CB][PBW];
#pragma unroll
    for (int i = 0; i < CB; i++)
#pragma unroll
        for (int j = 0; j < PBW; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

    // Epilogue operands that depend on nothing computed here are requested now, so their latency hides
    // under the main loop: the folded-BN bias of this lane's 8 cout quads, and (2-byte activations only,
    // for register budget) the skip-connection rows in the epilogue's (pixel row, 8 couts) layout.
    // epilogue store layout: a pixel row of the tile is CPW couts = LPR lanes x 8 couts; 64 / LPR rows per trip
    constexpr int LPR = 4 * CB, RPT = 64 / LPR, EIT = PXW / RPT;
    const int prow = lane / LPR, cg = lane % LPR;
    f32x4 biasv[CB][4];   // bf16 / f32: added in the accumulator layout, before the transpose
    f32x4 bias8[2], ds8[2];  // f16: bias and inverse weight scale of the 8 couts the lane owns AFTER the transpose
    if constexpr (!F16T) {
#pragma unroll
        for (int cb = 0; cb < CB; cb++)
#pragma unroll
            for (int g = 0; g < 4; g++) biasv[cb][g] = *reinterpret_cast<const f32x4*>(bias + cout0 + cb * 32 + g * 8 + h * 4);
    } else {
#pragma unroll
        for (int q = 0; q < 2; q++) {
            bias8[q] = *reinterpret_cast<const f32x4*>(bias + cout0 + cg * 8 + q * 4);
            ds8[q] = *reinterpret_cast<const f32x4*>(bias + cout + cout0 + cg * 8 + q * 4);
        }
    }
    constexpr bool RES_EARLY = HAS_RES && sizeof(T) == 2;
    T resv[EIT][8];
    if (RES_EARLY) {
#pragma unroll
        for (int i = 0; i < EIT; i++) {
            const size_t off = ((size_t)row0 + wave * PXW + i * RPT + prow) * (size_t)cout + cout0 + cg * 8;
            *reinterpret_cast<f32x4*>(resv[i]) = *reinterpret_cast<const f32x4*>(res + off);
        }
    }

    // A step is 12 fragment stages (3 taps x 4 k-slices); each stage = CB weight + 2 activation
    // fragments feeding 2 CB MFMAs.  Fragments are read AHEAD stages before the MFMAs that consume them
    // (register ring) so the LDS latency hides under the MFMAs in between.
    constexpr int AHEAD = CB == 2 ? 2 : 3, RING = AHEAD + 1;
    constexpr bool MEET = BIG || PBW == 1;  // more than one consumer wave per board
    int opaque = 0;
    for (int ch = 0; ch < nch; ch++) {
        const int abase = V2_LDS_ACT + (ch & 1) * 32768 + board_lds;
#pragma unroll
        for (int g = 0; g < 3; g++) {
            // all fragment reads of the previous step have returned before the loaders may reuse its slab
            {
                STAMP_ACC_BEGIN;
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                STAMP_ACC_END(4);
            }
            if (ch == 0 && g == 0) STAMP(1);
            // keep the per-step activation addresses from being hoisted out of the chunk loop (72 VGPRs)
            asm volatile("" : "+v"(opaque));
            constexpr int dummy = 0;
            (void)dummy;
            const int wslab = g * V2_SLAB;  // ring slot of step ch*3+g is g
            int baddr[3][PBW][4];
#pragma unroll
            for (int dxi = 0; dxi < 3; dxi++)
#pragma unroll
                for (int pb = 0; pb < PBW; pb++) {
                    const int hh = ph[pb] + (g - 1) + opaque, ww = pw[pb] + dxi - 1;
                    const bool ok = pvalid[pb] && (unsigned)hh < (unsigned)S && (unsigned)ww < (unsigned)S;
                    const int q = hh * S + ww;
                    // off the board: the zero area, at the row's own address mod 256 (q may be negative: only its low bits are used)
                    const int rowa = ok ? abase + q * 128 : V2_LDS_ZERO + (q & 1) * 128;
                    const int x0 = (h ^ ((q >> 1) & 7)) << 4;
#pragma unroll
                    for (int ks = 0; ks < 4; ks++) baddr[dxi][pb][ks] = rowa + (x0 ^ (ks << 5));
                }
            frag fa[RING][CB], fb[RING][PBW];
            auto load_stage = [&](int i, frag (&a)[CB], frag (&b)[PBW]) {
                const int dxi = i >> 2, ks = i & 3;
#pragma unroll
                for (int cb = 0; cb < CB; cb++)
                    a[cb] = *reinterpret_cast<const frag*>(smem + aaddr[ks][cb] + (wslab + dxi * 8192));
#pragma unroll
                for (int pb = 0; pb < PBW; pb++) b[pb] = *reinterpret_cast<const frag*>(smem + baddr[dxi][pb][ks]);
            };
#pragma unroll
            for (int i = 0; i < AHEAD; i++) load_stage(i, fa[i % RING], fb[i % RING]);
            __builtin_amdgcn_sched_group_barrier(0x100, (CB + PBW) * AHEAD, 0);
#pragma unroll
            for (int i = 0; i < 12; i++) {
                if (i + AHEAD < 12) load_stage(i + AHEAD, fa[(i + AHEAD) % RING], fb[(i + AHEAD) % RING]);
#pragma unroll
                for (int cb = 0; cb < CB; cb++)
#pragma unroll
                    for (int pb = 0; pb < PBW; pb++) Mfma<T>::mac(fa[i % RING][cb], fb[i % RING][pb], acc[cb][pb]);
                // pin the issue order: this stage's look-ahead reads, then its MFMAs
                if (i + AHEAD < 12) __builtin_amdgcn_sched_group_barrier(0x100, CB + PBW, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, CB * PBW * (sizeof(T) == 2 ? 1 : 4), 0);
            }
        }
    }

    // ---- epilogue ----
    // The accumulators hold, per lane, 4 consecutive couts of one pixel; written straight to NHWC
    // that is 64 8-byte stores per lane hitting 32 lines each.  Instead the wave transposes its
    // 64 px x 64 cout tile through its own (now idle) activation LDS region as f32 [px][64 cout]
    // (XOR-swizzled 16-byte slots), then every lane owns 8 consecutive couts of one pixel: one
    // 16/32-byte residual load and one 16/32-byte store per lane and pixel row, whole 128-byte lines.
    STAMP(2);
    if (MEET) {
        // the staging region below holds pixel rows the OTHER wave of this board may still be reading as
        // neighbours in its last step: meet first (the loader waves have left; a barrier counts live waves)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    {
        const size_t wrow0 = (size_t)row0 + wave * PXW;  // first tower row of this wave's tile
        const int tile0 = V2_LDS_ACT + wave * PXW * 128;  // px 0..31; px 32..63 live 32768 bytes further (CB = 2)
        // skip-connection rows in the final (pixel row, 8 couts) layout, requested before the transpose
        if (HAS_RES && !RES_EARLY) {
#pragma unroll
            for (int i = 0; i < EIT; i++) {
                const size_t off = (wrow0 + i * RPT + prow) * (size_t)cout + cout0 + cg * 8;
                if (sizeof(T) == 2) {
                    *reinterpret_cast<f32x4*>(resv[i]) = *reinterpret_cast<const f32x4*>(res + off);
                } else {
                    reinterpret_cast<f32x4*>(resv[i])[0] = reinterpret_cast<const f32x4*>(res + off)[0];
                    reinterpret_cast<f32x4*>(resv[i])[1] = reinterpret_cast<const f32x4*>(res + off)[1];
                }
            }
        }
        // staged tile: CB = 2: f32 [px][64 couts] = 256-byte rows, px 0..31 at tile0, px 32..63 32768 bytes further;
        //              CB = 1: f32 [px][32 couts] = 128-byte rows, all 64 at tile0
        auto stage_row = [&](int px) { return CB == 2 ? tile0 + (px >> 5) * 32768 + (px & 31) * 256 : tile0 + px * 128; };
#pragma unroll
        for (int cb = 0; cb < CB; cb++)
#pragma unroll
            for (int pb = 0; pb < PBW; pb++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    f32x4 v;
                    if constexpr (F16T) {
#pragma unroll
                        for (int i = 0; i < 4; i++) v[i] = acc[cb][pb][g * 4 + i];
                    } else {
                        const f32x4 bv = biasv[cb][g];
#pragma unroll
                        for (int i = 0; i < 4; i++) v[i] = acc[cb][pb][g * 4 + i] + bv[i];
                    }
                    const int slot = (cb * 8 + g * 2 + h) ^ (r & 7);
                    *reinterpret_cast<f32x4*>(smem + stage_row(pb * 32 + r) + slot * 16) = v;
                }
#pragma unroll
        for (int i = 0; i < EIT; i++) {
            const int px = i * RPT + prow;
            const char* rowp = smem + stage_row(px);
            const f32x4 lo = *reinterpret_cast<const f32x4*>(rowp + (((2 * cg) ^ (px & 7)) << 4));
            const f32x4 hi = *reinterpret_cast<const f32x4*>(rowp + (((2 * cg + 1) ^ (px & 7)) << 4));
            float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            const size_t off = (wrow0 + px) * (size_t)cout + cout0 + cg * 8;
            if constexpr (F16T) {  // the accumulator carries the weights' power-of-two scale: times its inverse is exact
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = __builtin_fmaf(v[j], ds8[j >> 2][j & 3], bias8[j >> 2][j & 3]);
            }
            if (HAS_RES) {
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = v[j] + (float)resv[i][j];
            }
            const bool valid = pslot0 + px < S * S;
            T ov[8];
            if constexpr (F16T) {
                float y[8];
#pragma unroll
                for (int j = 0; j < 8; j++) y[j] = valid && v[j] > 0.0f ? v[j] : 0.0f;
                if (flags & CONV_OUT_F32) {  // the tower's last layer: plain f32 rows for the head kernels
                    float* of = reinterpret_cast<float*>(out) + off;
                    if constexpr (CB == 1) {
                        reinterpret_cast<f32x4*>(of)[0] = *reinterpret_cast<f32x4*>(y);
                        reinterpret_cast<f32x4*>(of)[1] = *reinterpret_cast<f32x4*>(y + 4);
                    } else {
                        __builtin_nontemporal_store(*reinterpret_cast<f32x4*>(y), reinterpret_cast<f32x4*>(of));
                        __builtin_nontemporal_store(*reinterpret_cast<f32x4*>(y + 4), reinterpret_cast<f32x4*>(of) + 1);
                    }
                    continue;
                }
                note_saturation(y, valid, sat);
#pragma unroll
                for (int j = 0; j < 8; j++) ov[j] = (T)(y[j] < 65504.0f ? y[j] : 65504.0f);
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    float y = v[j] > 0.0f ? v[j] : 0.0f;
                    if (!valid) y = 0.0f;
                    ov[j] = (T)y;
                }
            }
            // non-temporal: the line leaves for memory early instead of at the kernel's end (plain stores: launch 18.35 ->
            // 18.7 us, one batch at a time unchanged, two batches in flight +3 %, self-play +0.8 %; profiles/r02_experiments.txt)
            // On the small-grid tile (CB = 1) plain stores: the output stays in the L2 of the XCD whose workgroups read it back as
            // the next layer's input (measured on the split conv: 18.5 against 19.9 us per launch at 64 leaves).
            if constexpr (CB == 1) {
                if (sizeof(T) == 2) {
                    *reinterpret_cast<f32x4*>(out + off) = *reinterpret_cast<f32x4*>(ov);
                } else {
                    reinterpret_cast<f32x4*>(out + off)[0] = reinterpret_cast<f32x4*>(ov)[0];
                    reinterpret_cast<f32x4*>(out + off)[1] = reinterpret_cast<f32x4*>(ov)[1];
                }
            } else if (sizeof(T) == 2) {
                __builtin_nontemporal_store(*reinterpret_cast<f32x4*>(ov), reinterpret_cast<f32x4*>(out + off));
            } else {
                __builtin_nontemporal_store(reinterpret_cast<f32x4*>(ov)[0], reinterpret_cast<f32x4*>(out + off));
                __builtin_nontemporal_store(reinterpret_cast<f32x4*>(ov)[1], reinterpret_cast<f32x4*>(out + off) + 1);
            }
        }
    }
    STAMP(3);
    STAMP_RT(6);
    STAMP_FLUSH(wave);
}

#ifdef CATTUS_STAMPS
extern "C" __attribute__((visibility("default"))) int cattus_hip_debug_stamps(unsigned long long* out, size_t n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), n * sizeof(unsigned long long));
}
#endif


// ------------------------------------------------------------------------------------------
// K1s: 3x3 conv of the split-precision tower (dtype f16x2)
// ------------------------------------------------------------------------------------------
//
// Every activation and every weight is a PAIR of f16 values, x = hi + lo with hi = f16(x), lo = f16(x - hi) (22
// significant bits), and a product is three MFMA terms, a_hi w_hi + a_lo w_hi + a_hi w_lo, accumulated in f32 (a_lo w_lo,
// 2^-22 of the product, is dropped).  The f16 MFMA takes f16 subnormals as they are and sums its 16 products more
// accurately than an f32 fmaf chain does (scripts/probes/mfma_f16_probe.hip -> profiles/r03_mfma_f16_probe.txt), so the
// tower's error against a float64 run is that of an f32 runtime.
//
// HBM layout: pairs interleaved in groups of 32 channels -- 128 bytes of a row are [hi of 32 channels | lo of the same
// 32], in `in` / `res` / `out` rows and in `w` rows alike -- so a chunk of the k walk carries 32 channels, as in the f32
// kernel.  Weights are pre-scaled per output channel by a power of two (their lo halves stay normal f16 numbers); `bias`
// is followed by the cout inverse scales, applied exactly in the epilogue.  The stem's planes are 0/1 (lo half zero).
//
// Same roles as conv3x3_mfma_v2_kernel (4 MFMA consumer waves with a 64 cout x 64 pixel tile each, 4 LDS-DMA loader
// waves, a 3-slab weight ring, two activation chunk buffers, one barrier per (chunk, kernel row)), with two differences:
//   * LDS rows have a pitch of 144 bytes (128 of data + 16 of padding) instead of an XOR swizzle: ds_read_b128 down 16
//     consecutive rows is conflict-free either way, but with a pitch every fragment address is `row base + constant`, so
//     tap, k-slice and hi / lo all fold into ds_read immediates; a step needs 6 vector adds of address arithmetic (the
//     swizzled form: ~100 VALU instructions per step, 400 cycles of a 2,300-cycle step).  The loader waves build the
//     padded image from dense global rows through their per-lane source addresses (the pad slot re-reads a data slot).
//   * a consumer stage is (tap, 16 channels): the hi and lo fragments of 2 weight blocks and 2 pixel blocks (8
//     ds_read_b128) feed 12 MFMAs, one stage (384 cycles of MFMA issue) of look-ahead.
// flags & CONV_OUT_F32: the output is written as plain f32 [row][cout] (the last tower layer, for the f32 head kernels).
constexpr int SP_TAP = 64 * SP;                         // the 64 cout rows of one tap
constexpr int SP_SLAB = 3 * SP_TAP;                     // 27,648 B = 27 LDS-DMA pieces of 1 KiB
constexpr int SP_ZERO = 256 * SP;                       // a buffer's zero area (off-board taps read it: device_common.h), behind its 256 rows
constexpr int SP_ABUF = SP_ZERO + ZAREA_SP;             // 37,216 B; the rows alone are 36 pieces
constexpr int SP_LDS_ACT = 3 * SP_SLAB;
constexpr int SP_LDS_TOTAL = SP_LDS_ACT + 2 * SP_ABUF;  // 157,376 B
static_assert(SP_ZERO % 256 == 0, "off-board reads keep their row's banks only if the zero area is 256-B aligned inside the buffer");
constexpr int SP_WPL = 7, SP_APL = 5;                   // pieces per loader wave: 28 >= 27 per slab, 20 >= 18 per half chunk

template <bool HAS_RES, bool BIG, bool STEM, int CB>
__global__ void __launch_bounds__(512, 2)
    conv3x3_split_kernel(const _Float16* __restrict__ in, const _Float16* __restrict__ w, const float* __restrict__ bias,
                         const _Float16* __restrict__ res, _Float16* __restrict__ out, unsigned* __restrict__ sat, int cin, int cout,
                         int S, int flags, StemPlanes<STEM> sp) {
    typedef _Float16 T;
    typedef Mfma<T>::frag frag;
    constexpr int KC = 32;        // channels (pairs) per 128-byte chunk
    constexpr int CPW = 32 * CB;  // output channels of this workgroup
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const bool is_loader = wave >= 4;
    STAMP_DECL;
    STAMP(0);
    STAMP_RT(5);

    const int nblk = gridDim.x, ncb = cout / CPW;
    int logical = blockIdx.x;
    if ((nblk & 7) == 0) logical = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
    const int cout0 = (logical % ncb) * CPW;
    const int row0 = (logical / ncb) * ROWS_PER_WG;

    if (tid < 2 * (ZAREA_SP / 16))
        reinterpret_cast<f32x4*>(smem + SP_LDS_ACT + (tid / (ZAREA_SP / 16)) * SP_ABUF + SP_ZERO)[tid % (ZAREA_SP / 16)] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the zero rows are written before the first barrier

    const int nch = cin / KC;
    const int T_total = nch * 3;
    const uint32_t row_bytes = (uint32_t)cin * 4;

    if (is_loader) {
        // ================================ loader waves ================================
        // Piece `pid` fills image bytes [pid * 1024, +1024): lane l writes 16-byte slot pid * 64 + l of the image, i.e. slot
        // c = that % 9 of image row that / 9; slot 8 is the padding (it re-reads slot 7).  A wave's surplus pieces repeat
        // the image's last piece, so every wave issues the same count and the vmcnt arithmetic below stays uniform.
        const int lw = wave - 4;
        uint32_t off_w[SP_WPL], off_a[2][SP_APL];
        int dst_w[SP_WPL], dst_a[2][SP_APL];
#pragma unroll
        for (int i = 0; i < SP_WPL; i++) {
            const int pid = min(lw * SP_WPL + i, 26);
            const int sidx = pid * 64 + lane, irow = sidx / 9, c = min(sidx - irow * 9, 7);
            const int tap_i = irow >> 6, row = CB == 2 ? (irow & 63) : (irow & 31);  // CB = 1: rows 32..63 of a tap are not read
            off_w[i] = ((uint32_t)(tap_i * cout + row)) * row_bytes + c * 16;
            dst_w[i] = pid * 1024;
        }
#pragma unroll
        for (int g = 0; g < 2; g++)
#pragma unroll
            for (int i = 0; i < SP_APL; i++) {
                const int id = g * 18 + min(lw * SP_APL + i, 17);
                const int sidx = id * 64 + lane, irow = sidx / 9, c = min(sidx - irow * 9, 7);
                off_a[g][i] = (uint32_t)irow * row_bytes + c * 16;
                dst_a[g][i] = id * 1024;
            }
        const char* wbase0 = reinterpret_cast<const char*>(w) + (size_t)cout0 * row_bytes;
        const char* abase0 = reinterpret_cast<const char*>(in) + (size_t)row0 * row_bytes;
        auto issue_w = [&](int t) {  // weight slab of step t -> ring slot t % 3
            const int ch = t / 3, g = t - ch * 3;
            const char* src = wbase0 + (size_t)(g * 3) * cout * row_bytes + (size_t)ch * 128;
            char* dst = smem + (t % 3) * SP_SLAB;
#pragma unroll
            for (int i = 0; i < SP_WPL; i++) glds16(src + off_w[i], dst + dst_w[i]);
        };
        auto issue_a = [&](int ch, int g) {  // half g of activation chunk ch -> buffer ch & 1
            const char* src = abase0 + (size_t)ch * 128;
            char* dst = smem + SP_LDS_ACT + (ch & 1) * SP_ABUF;
#pragma unroll
            for (int i = 0; i < SP_APL; i++) glds16(src + off_a[g][i], dst + dst_a[g][i]);
        };

        if constexpr (STEM) {
            // K0 fused: this thread expands pixel row (lw * 64 + lane) of the workgroup's 256 rows: channel c of the row is 1.0
            // where plane c has the pixel's bit (hi half, slots 0..3); the lo half (slots 4..7) is zero
            constexpr int MAXC = 32;
            const int row = lw * 64 + lane;
            const uint32_t grow = (uint32_t)(row0 + row), slots = BIG ? 128u : 64u;
            const uint32_t board = grow / slots, px = grow % slots;
            const bool live = board < sp.n && (int)px < S * S;
            typedef const __attribute__((address_space(1))) uint64_t* gu64p;
            const gu64p pl = (gu64p)(sp.planes + (size_t)(live ? board : 0) * sp.C * sp.w64 + (live ? (px >> 6) : 0));
            uint64_t words[MAXC];
#pragma unroll
            for (int c = 0; c < MAXC; c++) words[c] = (live && (uint32_t)c < sp.C) ? pl[(size_t)c * sp.w64] : 0ull;
            issue_w(0);
            issue_w(1);
            char* dstrow = smem + SP_LDS_ACT + row * SP;
#pragma unroll
            for (int sl = 0; sl < 8; sl++) {
                T vals[8];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int c = sl * 8 + i;
                    vals[i] = (c < MAXC && ((words[c < MAXC ? c : 0] >> (px & 63)) & 1ull)) ? (T)1.0f : (T)0.0f;
                }
                *reinterpret_cast<f32x4*>(dstrow + sl * 16) = *reinterpret_cast<f32x4*>(vals);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the rows are written before the first barrier
        } else {
            issue_a(0, 0);
            issue_a(0, 1);
            issue_w(0);
            issue_w(1);
        }
        int pending = SP_WPL;  // loads issued after the data of the upcoming step
        for (int t = 0; t < T_total; t++) {
            {
#ifdef CATTUS_STAMPS
                const unsigned long long t0_ = __builtin_amdgcn_s_memtime();
                if (pending == SP_WPL + SP_APL) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SP_WPL + SP_APL) : "memory");
                else if (pending == SP_WPL) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SP_WPL) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned long long t1_ = __builtin_amdgcn_s_memtime();
                asm volatile("s_barrier" ::: "memory");
                st_[4] += t1_ - t0_;
                st_[7] += __builtin_amdgcn_s_memtime() - t1_;
#else
                if (pending == SP_WPL + SP_APL) wait_vm_barrier<SP_WPL + SP_APL>();
                else if (pending == SP_WPL) wait_vm_barrier<SP_WPL>();
                else wait_vm_barrier<0>();
#endif
            }
            if (t == 0) STAMP(1);
            pending = 0;
            const int ch = t / 3, g = t - ch * 3;
            if (t + 2 < T_total) {
                issue_w(t + 2);
                pending += SP_WPL;
            }
            if (g < 2 && ch + 1 < nch) {
                issue_a(ch + 1, g);
                pending += SP_APL;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(2);
        STAMP(3);
        STAMP_RT(6);
        STAMP_FLUSH(wave);
        return;
    }

    // ================================ consumer waves ================================
    const int r = lane & 31, h = lane >> 5;
    const int pslot0 = BIG ? (wave & 1) * 64 : 0;
    const int board_row = BIG ? (wave >> 1) * 128 : wave * 64;  // first LDS row of this wave's board
    // Byte offset, inside an activation buffer, of the row each tap reads for this lane's two pixels (the buffer's zero
    // row for a tap off the board), plus the lane half's 16 bytes: loop-invariant; a step adds the buffer's base.
    int rowa[9][2];
#pragma unroll
    for (int pb = 0; pb < 2; pb++) {
        const int p = pslot0 + pb * 32 + r;
        const int ph = p / S, pw = p - ph * S;
        const bool pvalid = p < S * S;
#pragma unroll
        for (int t9 = 0; t9 < 9; t9++) {
            const int hh = ph + t9 / 3 - 1, ww = pw + t9 % 3 - 1;
            const bool ok = pvalid && (unsigned)hh < (unsigned)S && (unsigned)ww < (unsigned)S;
            const int at = (board_row + hh * S + ww) * SP + h * 16;  // off the board: only its low 8 bits are used
            rowa[t9][pb] = ok ? at : SP_ZERO + (at & 255);
        }
    }
    int aaddr[CB];
#pragma unroll
    for (int cb = 0; cb < CB; cb++) aaddr[cb] = (cb * 32 + r) * SP + h * 16;

    f32x16 acc[CB][2];
#pragma unroll
    for (int i = 0; i < CB; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

    // epilogue store layout: a pixel row of the tile is CPW couts = LPR lanes x 8 couts; 64 / LPR rows per trip.  The bias
    // and the inverse weight scale of the 8 couts the lane owns after the transpose, and the skip rows, are requested now.
    constexpr int LPR = 4 * CB, RPT = 64 / LPR, EIT = 64 / RPT;
    const int prow = lane / LPR, cg = lane % LPR;
    f32x4 bias8[2], ds8[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        bias8[q] = *reinterpret_cast<const f32x4*>(bias + cout0 + cg * 8 + q * 4);
        ds8[q] = *reinterpret_cast<const f32x4*>(bias + cout + cout0 + cg * 8 + q * 4);
    }
    const size_t orow = (size_t)cout * 2;  // elements of T per row of `res` / `out`
    const int ocol = ((cout0 + cg * 8) >> 5) * 64 + ((cout0 + cg * 8) & 31);  // hi values of the lane's 8 couts; lo 32 further
    // The skip rows (64 KiB per workgroup) are requested behind the first barrier, not here: the prologue is a burst of
    // ~90 KiB per CU that every workgroup of the chip issues at once, and these have the whole loop to arrive (measured
    // against requesting them here: no difference, 1.770 vs 1.773 ms per batch; profiles/r03_experiments.txt).
    T resv[EIT][16];
    auto request_skip_rows = [&]() {
#pragma unroll
        for (int i = 0; i < EIT; i++) {
            const size_t off = ((size_t)row0 + wave * 64 + i * RPT + prow) * orow + ocol;
            reinterpret_cast<f32x4*>(resv[i])[0] = *reinterpret_cast<const f32x4*>(res + off);
            reinterpret_cast<f32x4*>(resv[i])[1] = *reinterpret_cast<const f32x4*>(res + off + 32);
        }
    };

    int opaque = 0;
    for (int ch = 0; ch < nch; ch++) {
#pragma unroll
        for (int g = 0; g < 3; g++) {
            // all fragment reads of the previous step have returned before the loaders may reuse its slab
            {
                STAMP_ACC_BEGIN;
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                STAMP_ACC_END(4);
            }
            if (ch == 0 && g == 0) STAMP(1);
            if (HAS_RES && ch == 0 && g == 0) request_skip_rows();
            asm volatile("" : "+v"(opaque));  // keeps the six sums below inside the loop (both parities hoisted: 12 more VGPRs)
            const int bufbase = SP_LDS_ACT + (ch & 1) * SP_ABUF + opaque;
            const int wslab = g * SP_SLAB;  // ring slot of step ch * 3 + g is g
            int ba[3][2];
#pragma unroll
            for (int dxi = 0; dxi < 3; dxi++)
#pragma unroll
                for (int pb = 0; pb < 2; pb++) ba[dxi][pb] = rowa[g * 3 + dxi][pb] + bufbase;
            // stage i = (tap dxi, 16 channels k): slots k and 2 + k of the rows hold the hi and the lo halves
            frag sah[2][CB], sal[2][CB], sbh[2][2], sbl[2][2];
            auto load_stage = [&](int i, int s) {
                const int dxi = i >> 1, k = i & 1;
#pragma unroll
                for (int cb = 0; cb < CB; cb++) {
                    sah[s][cb] = *reinterpret_cast<const frag*>(smem + aaddr[cb] + (wslab + dxi * SP_TAP + k * 32));
                    sal[s][cb] = *reinterpret_cast<const frag*>(smem + aaddr[cb] + (wslab + dxi * SP_TAP + k * 32 + 64));
                }
#pragma unroll
                for (int pb = 0; pb < 2; pb++) {
                    sbh[s][pb] = *reinterpret_cast<const frag*>(smem + ba[dxi][pb] + k * 32);
                    sbl[s][pb] = *reinterpret_cast<const frag*>(smem + ba[dxi][pb] + k * 32 + 64);
                }
            };
            load_stage(0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * CB + 4, 0);
#pragma unroll
            for (int i = 0; i < 6; i++) {
                if (i + 1 < 6) load_stage(i + 1, (i + 1) & 1);
                const int s = i & 1;
#pragma unroll
                for (int cb = 0; cb < CB; cb++)
#pragma unroll
                    for (int pb = 0; pb < 2; pb++) {
                        Mfma<T>::mac(sal[s][cb], sbh[s][pb], acc[cb][pb]);
                        Mfma<T>::mac(sah[s][cb], sbl[s][pb], acc[cb][pb]);
                        Mfma<T>::mac(sah[s][cb], sbh[s][pb], acc[cb][pb]);
                    }
                // pin the issue order: this stage's look-ahead reads, then its MFMAs
                if (i + 1 < 6) __builtin_amdgcn_sched_group_barrier(0x100, 2 * CB + 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, CB * 6, 0);
            }
        }
    }

    // ---- epilogue: transpose through the wave's own (idle) activation rows as in conv3x3_mfma_v2_kernel, then per lane 8
    // consecutive couts of one pixel: * inverse scale + bias, + skip, ReLU, split into (hi, lo), two 16-byte stores ----
    STAMP(2);
    if (BIG) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the board's other wave may still read these rows
    {
        const size_t wrow0 = (size_t)row0 + wave * 64;
        const int tile0 = SP_LDS_ACT + wave * 64 * SP;  // the wave's 64 rows of buffer 0 (9,216 B); the same rows of buffer 1 behind
        // staged tile: CB = 2: f32 [px][64 couts] = 256-byte rows, px 0..31 in buffer 0, px 32..63 in buffer 1;
        //              CB = 1: f32 [px][32 couts] = 128-byte rows, all 64 in buffer 0
        auto stage_row = [&](int px) { return CB == 2 ? tile0 + (px >> 5) * SP_ABUF + (px & 31) * 256 : tile0 + px * 128; };
#pragma unroll
        for (int cb = 0; cb < CB; cb++)
#pragma unroll
            for (int pb = 0; pb < 2; pb++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    f32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] = acc[cb][pb][g * 4 + i];
                    const int slot = (cb * 8 + g * 2 + h) ^ (r & 7);
                    *reinterpret_cast<f32x4*>(smem + stage_row(pb * 32 + r) + slot * 16) = v;
                }
#pragma unroll
        for (int i = 0; i < EIT; i++) {
            const int px = i * RPT + prow;
            const char* rowp = smem + stage_row(px);
            const f32x4 lo4 = *reinterpret_cast<const f32x4*>(rowp + (((2 * cg) ^ (px & 7)) << 4));
            const f32x4 hi4 = *reinterpret_cast<const f32x4*>(rowp + (((2 * cg + 1) ^ (px & 7)) << 4));
            float v[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
            // the accumulator carries the weights' power-of-two scale: times its inverse is exact
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = v[j] * ds8[j >> 2][j & 3] + bias8[j >> 2][j & 3];
            if (HAS_RES) {
                // hi + lo is exact in f32 (lo is at most half an ulp of hi: 22 significant bits)
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = v[j] + ((float)resv[i][j] + (float)resv[i][8 + j]);
            }
            const bool valid = pslot0 + px < S * S;
            float y[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                y[j] = v[j] > 0.0f ? v[j] : 0.0f;
                if (!valid) y[j] = 0.0f;
            }
            if (flags & CONV_OUT_F32) {  // the tower's last layer: plain f32 rows for the head kernels (or a Winograd tower's stem)
                if (flags & CONV_WINO_IN) cap_wino_input(y, true, sat);
                float* of = reinterpret_cast<float*>(out) + (wrow0 + px) * (size_t)cout + cout0 + cg * 8;
                __builtin_nontemporal_store(*reinterpret_cast<f32x4*>(y), reinterpret_cast<f32x4*>(of));
                __builtin_nontemporal_store(*reinterpret_cast<f32x4*>(y + 4), reinterpret_cast<f32x4*>(of) + 1);
            } else {
                const size_t off = (wrow0 + px) * orow + ocol;
                T hi[8], lo[8];
                note_saturation(y, true, sat);  // rows past the board are zero already
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float yc = y[j] < 65504.0f ? y[j] : 65504.0f;  // saturate instead of overflowing to infinity
                    hi[j] = (T)yc;
                    lo[j] = (T)(yc - (float)hi[j]);
                }
                __builtin_nontemporal_store(*reinterpret_cast<f32x4*>(hi), reinterpret_cast<f32x4*>(out + off));
                __builtin_nontemporal_store(*reinterpret_cast<f32x4*>(lo), reinterpret_cast<f32x4*>(out + off + 32));
            }
        }
    }
    STAMP(3);
    STAMP_RT(6);
    STAMP_FLUSH(wave);
}

// ------------------------------------------------------------------------------------------
// K1s': the split-precision conv with the WEIGHT fragments taken straight from L2 into a register ring
// ------------------------------------------------------------------------------------------
//
// conv3x3_split_kernel moves every operand byte through LDS twice (LDS-DMA write, ds_read): at three MFMA terms per
// product its LDS port is ~90 % busy while the matrix pipe is ~60 % busy, and the 3-slab weight ring costs a barrier --
// with a drained fragment pipeline behind it -- per (chunk, kernel row).  Here only the ACTIVATIONS live in LDS.  The
// weights are uploaded a second time in MFMA fragment order ([32-cout block][stage][hi | lo][lane][8 f16], 2 KiB per
// block and stage, stage = ((chunk * 3 + dy) * 3 + dx) * 2 + k-half), so that a consumer wave's operand for one stage is
// CB x 2 fully coalesced global_load_dwordx4; it keeps SW_D stages in flight in registers (hand-placed loads and
// vmcnt waits: the compiler's own placement sinks a refill down to its use).  The four consumer waves of a workgroup
// fetch the same bytes within a few hundred cycles of each other (L1 / L2 hits: scripts/probes/bdirect_probe.hip ->
// profiles/r03_bdirect_probe.txt measured this loop at 61.5 k cycles against 70.1 k for the ring in LDS, MFMA floor 55.3 k).
//
// What is left for the loader waves: the activation chunks (two buffers) and the layer's skip rows, which now go to
// LDS as well (64 KiB at CB = 2) instead of sitting in 64 VGPRs of every consumer lane through the whole loop.
// Hand-off: ONE barrier per chunk, in front of the chunk's last stage.  A stage's pixel fragments are read one stage ahead,
// so a consumer that has waited for its LDS reads there has read everything it will ever read of the chunk -- the loaders
// may refill the buffer with chunk + 2, a whole chunk (6.9 k cycles of MFMA issue) before it is needed -- and the loaders
// arrive only after chunk + 1 has landed, so the last stage's look-ahead reads cross into the next chunk and the fragment
// pipeline never drains at a barrier.
// The MFMA sequence per accumulator is that of conv3x3_split_kernel, so the two kernels agree bit for bit.
//
// Small grids: with CB = 1 (32 couts per workgroup) the outputs are stored plainly (the next layer finds its input in the
// XCD's L2), and while even that grid would leave half of the CUs empty the workgroup covers 128 rows instead of 256
// (PBW = 1: 32 pixels per consumer wave, half the MFMA chain per wave, twice the workgroups): 14.0 us per launch at 64
// leaves of chess 20x256, 11.9 at 17 (256-row workgroups: 18.6 / 18.4; the LDS-ring kernel: 22.5 / 22.0).
constexpr int sw_lds_total(int cb, int pbw = 2) { return 2 * (128 * pbw * SP + ZAREA_SP) + 128 * pbw * 32 * cb * 4; }  // 139,968 B at CB = 2

// PBW: 32-pixel blocks per consumer wave.  2: the workgroup covers 256 tower rows (64 pixels per wave); 1 (with CB = 1, for
// grids that would otherwise leave CUs empty: <= 64 leaves of an 8x8 game at 256 filters): 128 rows, 32 pixels per wave -- twice
// the workgroups, half the MFMA chain per wave.
template <bool HAS_RES, bool BIG, bool STEM, int CB, int PBW = 2>
__global__ void __launch_bounds__(512, 2)
    conv3x3_splitw_kernel(const _Float16* __restrict__ in, const _Float16* __restrict__ wf, const float* __restrict__ bias,
                          const _Float16* __restrict__ res, _Float16* __restrict__ out, unsigned* __restrict__ sat, int cin, int cout,
                          int S, int flags, StemPlanes<STEM> sp) {
    typedef _Float16 T;
    typedef Mfma<T>::frag frag;
    constexpr int KC = 32;        // channels (pairs) per 128-byte chunk
    constexpr int CPW = 32 * CB;  // output channels of this workgroup
    constexpr int D = SW_D;
    static_assert(PBW == 2 || (PBW == 1 && CB == 1), "the 128-row workgroup exists for the 32-cout tile only");
    constexpr int RW = 128 * PBW;       // tower rows of this workgroup
    constexpr int PXW = 32 * PBW;       // pixels (rows) per consumer wave
    constexpr int ZERO = RW * SP;       // a buffer's zero area (device_common.h), behind its RW rows
    constexpr int ABUF = ZERO + ZAREA_SP;  // bytes of an activation buffer
    static_assert(ZERO % 256 == 0, "off-board reads keep their row's banks only if the zero area is 256-B aligned inside the buffer");
    constexpr int SKIP0 = 2 * ABUF;     // skip rows behind the two buffers
    constexpr int SKIP_ROW = CPW * 4;   // bytes of a skip row: [32 hi | 32 lo] per 32 couts
    constexpr int SKIP_PPW = 4 * CB * PBW;  // 1 KiB pieces of the skip rows per loader wave
    constexpr int NHALF = PBW;          // an activation chunk is NHALF x 18 pieces of 1 KiB
    static_assert(18 % D == 0, "a chunk's 18 stages must map onto whole turns of the ring");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const bool is_loader = wave >= 4;
    STAMP_DECL;
    STAMP(0);
    STAMP_RT(5);

    const int nblk = gridDim.x, ncb = cout / CPW;
    int logical = blockIdx.x;
    if ((nblk & 7) == 0) logical = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
    const int cout0 = (logical % ncb) * CPW;
    const int row0 = (logical / ncb) * RW;

    if (tid < 2 * (ZAREA_SP / 16)) reinterpret_cast<f32x4*>(smem + (tid / (ZAREA_SP / 16)) * ABUF + ZERO)[tid % (ZAREA_SP / 16)] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the zero rows are written before the first barrier

    const int nch = cin / KC;
    const int nst = nch * 18;
    const uint32_t row_bytes = (uint32_t)cin * 4;

    if (is_loader) {
        // ================================ loader waves ================================
        // their few LDS-DMA instructions share the vector-memory issue path with the consumers' weight loads (16 KiB per
        // stage and CU): let them go first
        __builtin_amdgcn_s_setprio(3);
        const int lw = wave - 4;
        uint32_t off_a[NHALF][SP_APL];
        int dst_a[NHALF][SP_APL];
#pragma unroll
        for (int g = 0; g < NHALF; g++)
#pragma unroll
            for (int i = 0; i < SP_APL; i++) {  // the padded image of a chunk, as in conv3x3_split_kernel
                const int id = g * 18 + min(lw * SP_APL + i, 17);
                const int sidx = id * 64 + lane, irow = sidx / 9, c = min(sidx - irow * 9, 7);
                off_a[g][i] = (uint32_t)irow * row_bytes + c * 16;
                dst_a[g][i] = id * 1024;
            }
        const char* abase0 = reinterpret_cast<const char*>(in) + (size_t)row0 * row_bytes;
        auto issue_chunk = [&](int ch) {  // activation chunk ch -> buffer ch & 1
            const char* src = abase0 + (size_t)ch * 128;
            char* dst = smem + (ch & 1) * ABUF;
#pragma unroll
            for (int g = 0; g < NHALF; g++)
#pragma unroll
                for (int i = 0; i < SP_APL; i++) glds16(src + off_a[g][i], dst + dst_a[g][i]);
        };
        if constexpr (STEM) {
            // K0 fused (one chunk, no DMA): this thread expands pixel row (lw * 64 + lane) of the workgroup's RW rows
            constexpr int MAXC = 32;
            const int row = lw * 64 + lane;
            const uint32_t grow = (uint32_t)(row0 + row), slots = BIG ? 128u : 64u;
            const uint32_t board = grow / slots, px = grow % slots;
            const bool live = row < RW && board < sp.n && (int)px < S * S;
            typedef const __attribute__((address_space(1))) uint64_t* gu64p;
            const gu64p pl = (gu64p)(sp.planes + (size_t)(live ? board : 0) * sp.C * sp.w64 + (live ? (px >> 6) : 0));
            uint64_t words[MAXC];
#pragma unroll
            for (int c = 0; c < MAXC; c++) words[c] = (live && (uint32_t)c < sp.C) ? pl[(size_t)c * sp.w64] : 0ull;
            char* dstrow = smem + row * SP;
#pragma unroll
            for (int sl = 0; sl < 8; sl++) {
                T vals[8];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int c = sl * 8 + i;
                    vals[i] = (c < MAXC && ((words[c < MAXC ? c : 0] >> (px & 63)) & 1ull)) ? (T)1.0f : (T)0.0f;
                }
                if (row < RW) *reinterpret_cast<f32x4*>(dstrow + sl * 16) = *reinterpret_cast<f32x4*>(vals);
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // P
        } else {
            issue_chunk(0);
            wait_vm_barrier<0>();  // P: chunk 0 has landed
            issue_chunk(1);        // the host guarantees at least two chunks (cin is a multiple of 64)
            if constexpr (HAS_RES) {
                // skip rows: byte q of the LDS image is byte q % SKIP_ROW of row q / SKIP_ROW of the workgroup's rows of `res`;
                // requested behind chunk 1, they have until the second barrier (nothing reads them before the epilogue)
                const char* rbase = reinterpret_cast<const char*>(res) + ((size_t)row0 * cout + cout0) * 4;
#pragma unroll
                for (int i = 0; i < SKIP_PPW; i++) {
                    const int pc = lw * SKIP_PPW + i;
                    const int q = pc * 1024 + lane * 16, row = q / SKIP_ROW, col = q % SKIP_ROW;
                    glds16(rbase + (size_t)row * cout * 4 + col, smem + SKIP0 + pc * 1024);
                }
            }
        }
        STAMP(1);
        for (int ch = 0; ch < nch; ch++) {
            {
                // chunk ch + 1 has landed (the younger skip rows may still be in flight at the first barrier)
#ifdef CATTUS_STAMPS
                const unsigned long long t0_ = __builtin_amdgcn_s_memtime();
                if (HAS_RES && ch == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SKIP_PPW) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned long long t1_ = __builtin_amdgcn_s_memtime();
                asm volatile("s_barrier" ::: "memory");
                st_[4] += t1_ - t0_;
                st_[7] += __builtin_amdgcn_s_memtime() - t1_;
#else
                if (HAS_RES && ch == 0) wait_vm_barrier<SKIP_PPW>();
                else wait_vm_barrier<0>();
#endif
            }
            if (ch + 2 < nch) issue_chunk(ch + 2);  // every consumer has read chunk ch: its buffer is free
        }
        STAMP(2);
        STAMP(3);
        STAMP_RT(6);
        STAMP_FLUSH(wave);
        return;
    }

    // ================================ consumer waves ================================
    // weight ring: slot d holds the CB x (hi, lo) fragments of stage s with s % D == d
    const char* wblk = reinterpret_cast<const char*>(wf) + (size_t)(cout0 >> 5) * nst * SW_STAGE;
    const uint32_t voff0 = lane * 16, voff1 = lane * 16 + (uint32_t)nst * SW_STAGE;
    u32x4 ring[D][CB * 2];
    auto load_stage = [&](u32x4(&slot)[CB * 2], const char* p) {
        if constexpr (CB == 2) {
            u32x4 l0, l1, l2, l3;
            asm volatile(
                "global_load_dwordx4 %0, %4, %6\n\tglobal_load_dwordx4 %1, %4, %6 offset:1024\n\t"
                "global_load_dwordx4 %2, %5, %6\n\tglobal_load_dwordx4 %3, %5, %6 offset:1024"
                : "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
                : "v"(voff0), "v"(voff1), "s"(p)
                : "memory");
            slot[0] = l0, slot[1] = l1, slot[2] = l2, slot[3] = l3;
        } else {
            u32x4 l0, l1;
            asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024"
                         : "=&v"(l0), "=&v"(l1)
                         : "v"(voff0), "s"(p)
                         : "memory");
            slot[0] = l0, slot[1] = l1;
        }
    };
    // all but the n youngest loads have returned; the asm owns the registers, so nothing reads them before the wait
    // (n is a constant once the stage loop is unrolled; the immediate has to be a literal)
    auto wait_stage = [&](u32x4(&slot)[CB * 2], int n) {
#define CATTUS_WAIT_CASE(N)                                                                                       \
    case N:                                                                                                       \
        if constexpr (CB == 2) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3)); \
        else asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(r0), "+v"(r1));                                        \
        break;
        u32x4 r0 = slot[0], r1 = slot[1], r2 = slot[CB == 2 ? 2 : 0], r3 = slot[CB == 2 ? 3 : 1];
        switch (n) {
            CATTUS_WAIT_CASE(0)
            CATTUS_WAIT_CASE(2)
            CATTUS_WAIT_CASE(4)
            CATTUS_WAIT_CASE(6)
            CATTUS_WAIT_CASE(8)
            CATTUS_WAIT_CASE(10)
            CATTUS_WAIT_CASE(12)
            CATTUS_WAIT_CASE(16)
            CATTUS_WAIT_CASE(20)
            default: __builtin_trap();
        }
#undef CATTUS_WAIT_CASE
        slot[0] = r0, slot[1] = r1;
        if constexpr (CB == 2) slot[2] = r2, slot[3] = r3;
    };

    const int r = lane & 31, h = lane >> 5;
    // epilogue operands first (ordinary loads, older than every ring load: the hand-counted waits below stay exact)
    constexpr int LPR = 4 * CB, RPT = 64 / LPR, EIT = PXW / RPT;
    const int prow = lane / LPR, cg = lane % LPR;
    f32x4 bias8[2], ds8[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        bias8[q] = *reinterpret_cast<const f32x4*>(bias + cout0 + cg * 8 + q * 4);
        ds8[q] = *reinterpret_cast<const f32x4*>(bias + cout + cout0 + cg * 8 + q * 4);
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int d = 0; d < D; d++) load_stage(ring[d], wblk + (size_t)d * SW_STAGE);

    constexpr int SLOTS = BIG ? 128 : 64, WPB = SLOTS / PXW;  // consumer waves per board
    const int pslot0 = (wave % WPB) * PXW;           // this wave's first pixel slot of its board
    const int board_row = (wave / WPB) * SLOTS;      // first LDS row of this wave's board
    int rowa[9][PBW];
#pragma unroll
    for (int pb = 0; pb < PBW; pb++) {
        const int p = pslot0 + pb * 32 + r;
        const int ph_ = p / S, pw = p - ph_ * S;
        const bool pvalid = p < S * S;
#pragma unroll
        for (int t9 = 0; t9 < 9; t9++) {
            const int hh = ph_ + t9 / 3 - 1, ww = pw + t9 % 3 - 1;
            const bool ok = pvalid && (unsigned)hh < (unsigned)S && (unsigned)ww < (unsigned)S;
            const int at = (board_row + hh * S + ww) * SP + h * 16;  // off the board: only its low 8 bits are used
            rowa[t9][pb] = ok ? at : ZERO + (at & 255);
        }
    }

    f32x16 acc[CB][PBW];
#pragma unroll
    for (int i = 0; i < CB; i++)
#pragma unroll
        for (int j = 0; j < PBW; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

    const size_t orow = (size_t)cout * 2;  // elements of T per row of `res` / `out`
    const int ocol = ((cout0 + cg * 8) >> 5) * 64 + ((cout0 + cg * 8) & 31);  // hi values of the lane's 8 couts; lo 32 further

    {
        STAMP_ACC_BEGIN;
        asm volatile("s_barrier" ::: "memory");  // P: chunk 0 is in buffer 0
        STAMP_ACC_END(4);
    }
    STAMP(1);
    // pixel fragments: [stage parity][pixel block]
    frag ph[2][PBW], pl[2][PBW];
#pragma unroll
    for (int pb = 0; pb < PBW; pb++) {
        ph[0][pb] = *reinterpret_cast<const frag*>(smem + rowa[0][pb]);
        pl[0][pb] = *reinterpret_cast<const frag*>(smem + rowa[0][pb] + 64);
    }
    int opaque = 0;
    // one chunk = 18 stages; LAST: the layer's final chunk (no look-ahead beyond it, the ring runs empty)
    auto chunk = [&](int ch, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        asm volatile("" : "+v"(opaque));  // keeps `rowa + buffer base` from being hoisted for both parities (36 VGPRs)
        const int bufbase = (ch & 1) * ABUF + opaque;
        const int nextbase = ((ch + 1) & 1) * ABUF + opaque;
        const int s0 = ch * 18;
#pragma unroll
        for (int j = 0; j < 18; j++) {
            if (j == 17) {
                // this wave has read all of chunk ch (stage 17's fragments came in during stage 16); behind the barrier chunk
                // ch + 1 is in the other buffer
                STAMP_ACC_BEGIN;
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                STAMP_ACC_END(4);
            }
            const int cur = j & 1, nxt = cur ^ 1;
            // one stage of look-ahead on the pixel fragments; a chunk's last stage reads the next chunk's first
            if (j + 1 < 18 || !LAST) {
                const int t = j + 1 < 18 ? (j + 1) >> 1 : 0, k = j + 1 < 18 ? (j + 1) & 1 : 0;
#pragma unroll
                for (int pb = 0; pb < PBW; pb++) {
                    const int a = rowa[t][pb] + (j + 1 < 18 ? bufbase : nextbase) + k * 32;
                    ph[nxt][pb] = *reinterpret_cast<const frag*>(smem + a);
                    pl[nxt][pb] = *reinterpret_cast<const frag*>(smem + a + 64);
                }
            }
            // the D - 1 younger stages stay in flight; the layer's last D stages are not refilled and count down instead
            const bool tail = LAST && j >= 18 - D;
            wait_stage(ring[j % D], tail ? CB * 2 * (17 - j) : CB * 2 * (D - 1));
#pragma unroll
            for (int cb = 0; cb < CB; cb++) {
                const frag wh = __builtin_bit_cast(frag, ring[j % D][cb * 2]);
                const frag wl = __builtin_bit_cast(frag, ring[j % D][cb * 2 + 1]);
#pragma unroll
                for (int pb = 0; pb < PBW; pb++) {
                    Mfma<T>::mac(wl, ph[cur][pb], acc[cb][pb]);
                    Mfma<T>::mac(wh, pl[cur][pb], acc[cb][pb]);
                    Mfma<T>::mac(wh, ph[cur][pb], acc[cb][pb]);
                }
            }
            if (!tail) load_stage(ring[j % D], wblk + (size_t)(s0 + j + D) * SW_STAGE);  // D stages ahead
        }
    };
    for (int ch = 0; ch + 1 < nch; ch++) chunk(ch, std::false_type{});
    chunk(nch - 1, std::true_type{});

    // ---- epilogue: as conv3x3_split_kernel (the last chunk's barrier has every wave out of the buffers); the skip values
    // come from LDS ----
    STAMP(2);
    {
        // all skip values of the lane at once (the ring's registers are free now): one LDS latency instead of one per trip
        T resv[EIT][16];
        if (HAS_RES) {
#pragma unroll
            for (int i = 0; i < EIT; i++) {
                const char* sk = smem + SKIP0 + (wave * PXW + i * RPT + prow) * SKIP_ROW + ((cg * 8) >> 5) * 128 + ((cg * 8) & 31) * 2;
                reinterpret_cast<f32x4*>(resv[i])[0] = *reinterpret_cast<const f32x4*>(sk);
                reinterpret_cast<f32x4*>(resv[i])[1] = *reinterpret_cast<const f32x4*>(sk + 64);
            }
        }
        const size_t wrow0 = (size_t)row0 + wave * PXW;
        const int tile0 = wave * PXW * SP;  // the wave's own rows of buffer 0 (PXW x 144 B); the same rows of buffer 1 behind
        auto stage_row = [&](int px) { return CB == 2 ? tile0 + (px >> 5) * ABUF + (px & 31) * 256 : tile0 + px * 128; };
#pragma unroll
        for (int cb = 0; cb < CB; cb++)
#pragma unroll
            for (int pb = 0; pb < PBW; pb++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    f32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; i++) v[i] = acc[cb][pb][g * 4 + i];
                    const int slot = (cb * 8 + g * 2 + h) ^ (r & 7);
                    *reinterpret_cast<f32x4*>(smem + stage_row(pb * 32 + r) + slot * 16) = v;
                }
#pragma unroll
        for (int i = 0; i < EIT; i++) {
            const int px = i * RPT + prow;
            const char* rowp = smem + stage_row(px);
            const f32x4 lo4 = *reinterpret_cast<const f32x4*>(rowp + (((2 * cg) ^ (px & 7)) << 4));
            const f32x4 hi4 = *reinterpret_cast<const f32x4*>(rowp + (((2 * cg + 1) ^ (px & 7)) << 4));
            float v[8] = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
            // the inverse scale is a power of two, so the product is exact and one fused multiply-add rounds exactly as the
            // multiply followed by the add of conv3x3_split_kernel does
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = __builtin_fmaf(v[j], ds8[j >> 2][j & 3], bias8[j >> 2][j & 3]);
            if (HAS_RES) {
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = v[j] + ((float)resv[i][j] + (float)resv[i][8 + j]);  // hi + lo is exact in f32
            }
            const bool valid = pslot0 + px < S * S;  // slots past the board stay zero: masked once, on the packed vectors
            float y[8];
#pragma unroll
            for (int j = 0; j < 8; j++) y[j] = v[j] > 0.0f ? v[j] : 0.0f;
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
            if (flags & CONV_OUT_F32) {
                if (flags & CONV_WINO_IN) cap_wino_input(y, valid, sat);
                float* of = reinterpret_cast<float*>(out) + (wrow0 + px) * (size_t)cout + cout0 + cg * 8;
                const f32x4 y0 = valid ? *reinterpret_cast<f32x4*>(y) : zero4, y1 = valid ? *reinterpret_cast<f32x4*>(y + 4) : zero4;
                if constexpr (CB == 2) {
                    __builtin_nontemporal_store(y0, reinterpret_cast<f32x4*>(of));
                    __builtin_nontemporal_store(y1, reinterpret_cast<f32x4*>(of) + 1);
                } else {  // small grid: the head kernels find the rows in L2 (see below)
                    reinterpret_cast<f32x4*>(of)[0] = y0;
                    reinterpret_cast<f32x4*>(of)[1] = y1;
                }
            } else {
                const size_t off = (wrow0 + px) * orow + ocol;
                T hi[8], lo[8];
                note_saturation(y, valid, sat);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float yc = y[j] < 65504.0f ? y[j] : 65504.0f;
                    hi[j] = (T)yc;
                    lo[j] = (T)(yc - (float)hi[j]);
                }
                const f32x4 hi4o = valid ? *reinterpret_cast<f32x4*>(hi) : zero4, lo4o = valid ? *reinterpret_cast<f32x4*>(lo) : zero4;
                if constexpr (CB == 2) {
                    // the full grid: a layer's 16.7 MB of output and its 2.4 MB of weights do not fit the XCDs' L2 together
                    __builtin_nontemporal_store(hi4o, reinterpret_cast<f32x4*>(out + off));
                    __builtin_nontemporal_store(lo4o, reinterpret_cast<f32x4*>(out + off + 32));
                } else {
                    // the small-grid tile (<= 128 leaves of an 8x8 game): the output stays in the L2 of the XCD whose workgroups
                    // read it back as the next layer's input (the block remap keeps a row block on one XCD): 18.5 us per
                    // launch against 19.9 with non-temporal stores at batch 64, no difference at 128
                    *reinterpret_cast<f32x4*>(out + off) = hi4o;
                    *reinterpret_cast<f32x4*>(out + off + 32) = lo4o;
                }
            }
        }
    }
    STAMP(3);
    STAMP_RT(6);
    STAMP_FLUSH(wave);
}

// Every conv variant that exists, with its opt-in for > 64 KiB of dynamic LDS (a per-device function attribute).
// Called once per device from cattus_hip_create (under its lock), so that no launch ever races the attribute call.
template <typename T>
static hipError_t conv_attrs_for() {
    hipError_t err = hipSuccess;
    auto set = [&](const void* fn) {
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, V2_LDS_TOTAL);
        if (e != hipSuccess && err == hipSuccess) err = e;
    };
#define CATTUS_ATTR_CB(CBV)                                                                           \
    set(reinterpret_cast<const void*>(&conv3x3_mfma_v2_kernel<T, false, false, false, CBV>));         \
    set(reinterpret_cast<const void*>(&conv3x3_mfma_v2_kernel<T, true, false, false, CBV>));          \
    set(reinterpret_cast<const void*>(&conv3x3_mfma_v2_kernel<T, false, true, false, CBV>));          \
    set(reinterpret_cast<const void*>(&conv3x3_mfma_v2_kernel<T, true, true, false, CBV>));           \
    set(reinterpret_cast<const void*>(&conv3x3_mfma_v2_kernel<T, false, false, true, CBV>));          \
    set(reinterpret_cast<const void*>(&conv3x3_mfma_v2_kernel<T, false, true, true, CBV>));
    CATTUS_ATTR_CB(1)
    CATTUS_ATTR_CB(2)
#undef CATTUS_ATTR_CB
    // the 128-row workgroups of the 32-cout tile
    set(reinterpret_cast<const void*>(&conv3x3_mfma_v2_kernel<T, false, false, false, 1, 1>));
    set(reinterpret_cast<const void*>(&conv3x3_mfma_v2_kernel<T, true, false, false, 1, 1>));
    set(reinterpret_cast<const void*>(&conv3x3_mfma_v2_kernel<T, false, true, false, 1, 1>));
    set(reinterpret_cast<const void*>(&conv3x3_mfma_v2_kernel<T, true, true, false, 1, 1>));
    set(reinterpret_cast<const void*>(&conv3x3_mfma_v2_kernel<T, false, false, true, 1, 1>));
    set(reinterpret_cast<const void*>(&conv3x3_mfma_v2_kernel<T, false, true, true, 1, 1>));
    return err;
}
static hipError_t split_attrs() {
    hipError_t err = hipSuccess;
    auto set = [&](const void* fn) {
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, SP_LDS_TOTAL);
        if (e != hipSuccess && err == hipSuccess) err = e;
    };
#define CATTUS_ATTR_CB(CBV)                                                                  \
    set(reinterpret_cast<const void*>(&conv3x3_split_kernel<false, false, false, CBV>));     \
    set(reinterpret_cast<const void*>(&conv3x3_split_kernel<true, false, false, CBV>));      \
    set(reinterpret_cast<const void*>(&conv3x3_split_kernel<false, true, false, CBV>));      \
    set(reinterpret_cast<const void*>(&conv3x3_split_kernel<true, true, false, CBV>));       \
    set(reinterpret_cast<const void*>(&conv3x3_split_kernel<false, false, true, CBV>));      \
    set(reinterpret_cast<const void*>(&conv3x3_split_kernel<false, true, true, CBV>));
    CATTUS_ATTR_CB(1)
    CATTUS_ATTR_CB(2)
#undef CATTUS_ATTR_CB
    auto setw = [&](const void* fn, int cb) {
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, sw_lds_total(cb));
        if (e != hipSuccess && err == hipSuccess) err = e;
    };
#define CATTUS_ATTR_CB(CBV)                                                                        \
    setw(reinterpret_cast<const void*>(&conv3x3_splitw_kernel<false, false, false, CBV>), CBV);    \
    setw(reinterpret_cast<const void*>(&conv3x3_splitw_kernel<true, false, false, CBV>), CBV);     \
    setw(reinterpret_cast<const void*>(&conv3x3_splitw_kernel<false, true, false, CBV>), CBV);     \
    setw(reinterpret_cast<const void*>(&conv3x3_splitw_kernel<true, true, false, CBV>), CBV);      \
    setw(reinterpret_cast<const void*>(&conv3x3_splitw_kernel<false, false, true, CBV>), CBV);     \
    setw(reinterpret_cast<const void*>(&conv3x3_splitw_kernel<false, true, true, CBV>), CBV);
    CATTUS_ATTR_CB(1)
    CATTUS_ATTR_CB(2)
#undef CATTUS_ATTR_CB
    // the 128-row workgroup needs no opt-in (53,536 B)
    static_assert(sw_lds_total(1, 1) <= 64 * 1024, "the small tile fits the default dynamic LDS limit");
    return err;
}

void launch_conv3x3_mfma(Act act, const void* in, const void* w, const float* bias, const void* res, void* out,
                         uint32_t bpad, uint32_t cin, uint32_t cout, uint32_t S, hipStream_t st, hipEvent_t ev_start,
                         hipEvent_t ev_stop, const StemInput* stem, int flags, const ConvOpts& opts) {
    const uint32_t slots = tower_slots(S);
    // 256 rows x 64 couts per workgroup; 256 rows x 32 couts while that grid would leave half of the CUs empty
    const uint32_t full_grid = (bpad * slots / ROWS_PER_WG) * (cout / COUT_PER_WG);
    int cb = full_grid <= 128 ? 1 : 2;
    if (opts.cb == 1 || opts.cb == 2) cb = opts.cb;
    unsigned* const sat = opts.saturated;
    if (act_f16_family(act) && !sat) {
        fprintf(stderr, "cattus: launch_conv3x3_mfma: the f16 towers need ConvOpts::saturated\n");
        abort();
    }
    const dim3 grid(full_grid * (cb == 1 ? 2 : 1));
    if (act == Act::F16S) {
        typedef _Float16 H;
        const bool wfrag = (flags & CONV_W_FRAG) != 0;  // `w` is in fragment order: the register-ring kernel
        if (wfrag && !stem && cin < 64) {  // its loaders request two chunks up front (kernels.h); the evaluator pads filters to 64
            fprintf(stderr, "cattus: launch_conv3x3_mfma: CONV_W_FRAG needs cin >= 64 on a non-stem layer (got %u)\n", cin);
            abort();
        }
        // 128 rows x 32 couts per workgroup while even the 32-cout grid would leave half of the CUs empty (register ring only)
        const int pbw_forced = opts.pbw;
        const bool half_rows = wfrag && cb == 1 && ((grid.x <= 128 && pbw_forced != 2) || pbw_forced == 1);
        const dim3 grid_half(grid.x * 2);
#define CATTUS_LAUNCH_SPLIT(R, BIG, STEMV, CBV, SPV)                                                                                   \
    do {                                                                                                                               \
        if (wfrag && half_rows && CBV == 1)                                                                                            \
            hipExtLaunchKernelGGL((conv3x3_splitw_kernel<R, BIG, STEMV, 1, 1>), grid_half, dim3(512), sw_lds_total(1, 1), st, ev_start, \
                                  ev_stop, 0, (const H*)in, (const H*)w, bias, (const H*)res, (H*)out, sat, (int)cin, (int)cout, (int)S,   \
                                  flags, SPV);                                                                                         \
        else if (wfrag)                                                                                                                \
            hipExtLaunchKernelGGL((conv3x3_splitw_kernel<R, BIG, STEMV, CBV>), grid, dim3(512), sw_lds_total(CBV), st, ev_start,       \
                                  ev_stop, 0, (const H*)in, (const H*)w, bias, (const H*)res, (H*)out, sat, (int)cin, (int)cout, (int)S,   \
                                  flags, SPV);                                                                                         \
        else                                                                                                                           \
            hipExtLaunchKernelGGL((conv3x3_split_kernel<R, BIG, STEMV, CBV>), grid, dim3(512), SP_LDS_TOTAL, st, ev_start, ev_stop, 0, \
                                  (const H*)in, (const H*)w, bias, (const H*)res, (H*)out, sat, (int)cin, (int)cout, (int)S, flags, SPV);   \
    } while (0)
#define CATTUS_LAUNCH_SPLIT_CB(CBV)                                                                       \
    do {                                                                                                  \
        if (stem) {                                                                                       \
            const StemPlanes<true> spv{stem->planes, stem->n, stem->C, stem->w64};                        \
            if (slots == 128) CATTUS_LAUNCH_SPLIT(false, true, true, CBV, spv);                           \
            else CATTUS_LAUNCH_SPLIT(false, false, true, CBV, spv);                                       \
        } else if (slots == 128) {                                                                        \
            if (res) CATTUS_LAUNCH_SPLIT(true, true, false, CBV, StemPlanes<false>{});                    \
            else CATTUS_LAUNCH_SPLIT(false, true, false, CBV, StemPlanes<false>{});                       \
        } else {                                                                                          \
            if (res) CATTUS_LAUNCH_SPLIT(true, false, false, CBV, StemPlanes<false>{});                   \
            else CATTUS_LAUNCH_SPLIT(false, false, false, CBV, StemPlanes<false>{});                      \
        }                                                                                                 \
    } while (0)
        if (cb == 1) CATTUS_LAUNCH_SPLIT_CB(1);
        else CATTUS_LAUNCH_SPLIT_CB(2);
#undef CATTUS_LAUNCH_SPLIT_CB
#undef CATTUS_LAUNCH_SPLIT
        return;
    }
    // 128 rows x 32 couts per workgroup while even the 32-cout grid would leave half of the CUs empty (as the split conv does)
    const int pbw_forced2 = opts.pbw;
    const bool half_rows2 = cb == 1 && ((grid.x <= 128 && pbw_forced2 != 2) || pbw_forced2 == 1);
    const dim3 grid_half2(grid.x * 2);
#define CATTUS_LAUNCH_CONV2(T, R, BIG, CBV)                                                               \
    do {                                                                                                  \
        if (half_rows2 && CBV == 1)                                                                       \
            hipExtLaunchKernelGGL((conv3x3_mfma_v2_kernel<T, R, BIG, false, 1, 1>), grid_half2, dim3(256 + 64 * NLOAD), V2_LDS_TOTAL, st, ev_start, ev_stop, 0, \
                                  (const T*)in, (const T*)w, bias, (const T*)res, (T*)out, sat, (int)cin, (int)cout, (int)S, flags, StemPlanes<false>{}); \
        else                                                                                              \
            hipExtLaunchKernelGGL((conv3x3_mfma_v2_kernel<T, R, BIG, false, CBV>), grid, dim3(256 + 64 * NLOAD), V2_LDS_TOTAL, st, ev_start, ev_stop, 0, \
                                  (const T*)in, (const T*)w, bias, (const T*)res, (T*)out, sat, (int)cin, (int)cout, (int)S, flags, StemPlanes<false>{}); \
    } while (0)
#define CATTUS_LAUNCH_STEM(T, BIG, CBV)                                                                   \
    do {                                                                                                  \
        if (half_rows2 && CBV == 1)                                                                       \
            hipExtLaunchKernelGGL((conv3x3_mfma_v2_kernel<T, false, BIG, true, 1, 1>), grid_half2, dim3(256 + 64 * NLOAD), V2_LDS_TOTAL, st, ev_start, ev_stop, 0, \
                                  (const T*)nullptr, (const T*)w, bias, (const T*)nullptr, (T*)out, sat, (int)cin, (int)cout, (int)S, flags, \
                                  StemPlanes<true>{stem->planes, stem->n, stem->C, stem->w64});           \
        else                                                                                              \
            hipExtLaunchKernelGGL((conv3x3_mfma_v2_kernel<T, false, BIG, true, CBV>), grid, dim3(256 + 64 * NLOAD), V2_LDS_TOTAL, st, ev_start, ev_stop, 0, \
                                  (const T*)nullptr, (const T*)w, bias, (const T*)nullptr, (T*)out, sat, (int)cin, (int)cout, (int)S, flags, \
                                  StemPlanes<true>{stem->planes, stem->n, stem->C, stem->w64});           \
    } while (0)
#define CATTUS_LAUNCH_CONV2_CB(T, CBV)                            \
    do {                                                          \
        if (stem) {                                               \
            if (slots == 128) CATTUS_LAUNCH_STEM(T, true, CBV);   \
            else CATTUS_LAUNCH_STEM(T, false, CBV);               \
        } else if (slots == 128) {                                \
            if (res) CATTUS_LAUNCH_CONV2(T, true, true, CBV);     \
            else CATTUS_LAUNCH_CONV2(T, false, true, CBV);        \
        } else {                                                  \
            if (res) CATTUS_LAUNCH_CONV2(T, true, false, CBV);    \
            else CATTUS_LAUNCH_CONV2(T, false, false, CBV);       \
        }                                                         \
    } while (0)
#define CATTUS_LAUNCH_CONV2_T(T)                  \
    do {                                          \
        if (cb == 1) CATTUS_LAUNCH_CONV2_CB(T, 1); \
        else CATTUS_LAUNCH_CONV2_CB(T, 2);        \
    } while (0)
    if (act == Act::BF16) CATTUS_LAUNCH_CONV2_T(__bf16);
    else if (act == Act::F16) CATTUS_LAUNCH_CONV2_T(_Float16);
    else CATTUS_LAUNCH_CONV2_T(float);
#undef CATTUS_LAUNCH_CONV2_T
#undef CATTUS_LAUNCH_CONV2_CB
#undef CATTUS_LAUNCH_STEM
#undef CATTUS_LAUNCH_CONV2
}

// ------------------------------------------------------------------------------------------
// K1 resident: the whole tower of a <= 64-filter network in ONE launch, activations stay in LDS
// ------------------------------------------------------------------------------------------
//
// With 64 (padded) filters there is a single output-channel slab, so the input of a workgroup's next layer
// is its own output: nothing has to be handed to another workgroup, and nothing has to go through HBM.
// The workgroup expands its boards' bitboard planes into LDS (the plane pack, fused), then runs stem +
// 2 convs per residual block with the block input and the intermediate kept in two LDS buffers of
// [rows][64 ch] bf16 (XOR-swizzled 128-byte rows, the conv kernel's layout), streaming only the weights:
// waves 4-7 run the same LDS-DMA ring as in conv3x3_mfma_v2_kernel (3 slabs of [3 taps][64 cout][128 B]),
// two steps ahead across layer boundaries; waves 0-3 are MFMA consumers.  One s_barrier per (layer, kernel
// row) orders the ring and, at a layer boundary, one wave's LDS writes against its neighbours' reads.
// Arithmetic per output element is that of the per-layer kernel, operation for operation (bf16 operands,
// f32 accumulation in the same order, + bias, + skip, ReLU, one rounding to bf16), so results are bit-identical.
//
// CH = 2: 128 rows per workgroup, wave w owns row block w>>1 and the 32 couts of half w&1 (1x2 MFMA tiles).
// CH = 4: 64 rows = one 64-slot board per workgroup (<= 256 boards), wave w owns the 32 pixels of half w>>1 and the
//         32 couts of half w&1 (one tile).
// (A 256-row workgroup with 2x2 tiles per wave was within 1.5 % of CH = 2 from 1024 boards up, at 256 VGPRs and 93
//  spilled ones: dropped.)
// 2-byte activations only: f32 rows (2 chunks) would need 128 KiB besides the weight ring.
constexpr int R_LDS_W = 0;                    // 3 x 24 KiB
// behind the ring: 2 x (ROWS x 128 B + a 128-byte zero row)
// LS (CH = 4, boards of <= 63 pixels): a layer is ONE step.  The ring holds two whole layers (6 slabs = 144 KiB), the
// loaders run one layer ahead and there is one barrier per layer instead of three; the consumer's fragment pipeline
// runs through all 36 stages of the layer.  The two 8 KiB activation buffers fill the rest of the 160 KiB exactly, so
// there is no room for zero rows: pixel slot 63, which such a board does not use, is kept zero and serves as one.
constexpr int tower64_lds_bytes(int ch, bool ls) { return ls ? 6 * V2_SLAB + 2 * 64 * 128 : 3 * V2_SLAB + 2 * ((256 / ch) * 128 + 256); }

template <int CH, bool BIG, bool LS = false>
__global__ void __launch_bounds__(512, 2) tower64_lds_kernel(Tower64Args A) {
    static_assert(NLOAD == 4, "512 threads: four consumer and four loader waves");
    static_assert(!LS || CH == 4, "layer steps: one board per workgroup only");
    constexpr int R_LDS_ACT = (LS ? 6 : 3) * V2_SLAB;
    typedef __bf16 T;
    typedef Mfma<T>::frag frag;
    constexpr int ROWS = 256 / CH;      // tower rows of this workgroup
    static_assert(CH == 2 || CH == 4, "128 or 64 rows per workgroup");
    constexpr int CB = 1;                // 32-cout blocks per consumer wave
    constexpr int NPB = CH == 4 ? 1 : 2; // 32-pixel blocks per consumer wave
    static_assert(!(BIG && CH == 4), "a 128-slot board needs 128 rows in one workgroup");
    constexpr int ZERO_OFF = LS ? 63 * 128 : ROWS * 128;    // the zero row of a buffer (padding pixels read it): behind its rows / slot 63
    constexpr int ACT_BYTES = LS ? ROWS * 128 : ZERO_OFF + 256;  // buffer stride (not LS: a 256-byte zero area, device_common.h)
    constexpr int SLOTS_PER_BOARD = BIG ? 128 : 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const bool is_loader = wave >= 4;
    const int S = (int)A.S;
    const int row0 = blockIdx.x * ROWS;
    const int nlayers = (int)A.nlayers;
    STAMP_DECL;
    STAMP(0);

    if (tid < (LS ? 16 : 32)) reinterpret_cast<f32x4*>(smem + R_LDS_ACT + (tid / (LS ? 8 : 16)) * ACT_BYTES + ZERO_OFF)[tid % (LS ? 8 : 16)] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int T_total = nlayers * 3;
    // the loader waves put the first two weight slabs on their way before they help with the plane expansion
    const int lw = wave - 4;
    uint32_t off_w[WPL];
    int dst_w[WPL];
    auto issue_w = [&](int t) {  // weight slab of step t (layer t/3, kernel row t%3) -> ring slot t % 3
        const int layer = t / 3, g = t - layer * 3;
        const char* src = reinterpret_cast<const char*>(A.layers[layer].w) + (size_t)(g * 3) * 64 * 128;
        char* dst = smem + R_LDS_W + g * V2_SLAB;
#pragma unroll
        for (int i = 0; i < WPL; i++) glds16(src + off_w[i], dst + dst_w[i]);
    };
    auto issue_layer = [&](int layer) {  // LS: the three slabs of a layer -> ring half layer & 1
        const char* src = reinterpret_cast<const char*>(A.layers[layer].w);
        char* dst = smem + R_LDS_W + (layer & 1) * 3 * V2_SLAB;
#pragma unroll
        for (int g = 0; g < 3; g++)
#pragma unroll
            for (int i = 0; i < WPL; i++) glds16(src + (size_t)(g * 3) * 64 * 128 + off_w[i], dst + g * V2_SLAB + dst_w[i]);
    };
    if (is_loader) {
        const int prow = lane >> 3, pslot = lane & 7;
#pragma unroll
        for (int i = 0; i < WPL; i++) {
            const int pid = lw * WPL + i;  // 0..23: tap_i = pid/8, 8 rows each
            const int tap_i = pid >> 3, row = (pid & 7) * 8 + prow;
            const int c = pslot ^ ((row >> 1) & 7);
            off_w[i] = ((uint32_t)(tap_i * 64 + row)) * 128 + c * 16;
            dst_w[i] = pid * 1024;
        }
        if constexpr (LS) {
            issue_layer(0);
            if (nlayers > 1) issue_layer(1);
        } else {
            issue_w(0);
            if (T_total > 1) issue_w(1);
        }
    }

    // ---- stem input: bitboard planes -> buffer 1, [row][64 ch], 1.0 where the plane has the pixel's bit ----
    {
        const int hw = S * S;
        const uint32_t cslots = (A.C + 7) / 8;  // 16-byte channel groups that hold a real plane
        constexpr int ITEMS = ROWS * 8 / 512;   // (row, channel group) pairs per thread
        typedef const __attribute__((address_space(1))) uint64_t* gu64p;
        // all plane words first (one round trip to memory for the whole expansion), then the rows
        uint64_t words[ITEMS][8];
        bool live[ITEMS];
#pragma unroll
        for (int it = 0; it < ITEMS; it++) {
            const int v = tid + it * 512;
            const int row = v >> 3, sl = v & 7;
            const uint32_t grow = (uint32_t)(row0 + row);
            const uint32_t board = grow / SLOTS_PER_BOARD, px = grow % SLOTS_PER_BOARD;
            live[it] = (uint32_t)sl < cslots && board < A.n && (int)px < hw;
            const gu64p pl = (gu64p)(A.planes + (size_t)(live[it] ? board : 0) * A.C * A.w64 + (live[it] ? (px >> 6) : 0));
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t c = sl * 8 + i;
                words[it][i] = (live[it] && c < A.C) ? pl[(size_t)c * A.w64] : 0ull;
            }
        }
#pragma unroll
        for (int it = 0; it < ITEMS; it++) {
            const int v = tid + it * 512;
            const int row = v >> 3, sl = v & 7;
            const uint32_t px = (uint32_t)(row0 + row) % SLOTS_PER_BOARD;
            bf16x8 vals;
#pragma unroll
            for (int i = 0; i < 8; i++) vals[i] = ((words[it][i] >> (px & 63)) & 1ull) ? (T)1.0f : (T)0.0f;
            *reinterpret_cast<bf16x8*>(smem + R_LDS_ACT + ACT_BYTES + row * 128 + ((sl ^ ((row >> 1) & 7)) << 4)) = vals;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // written before this wave's first barrier
    }

    if (is_loader) {
        // ================================ loader waves: the weight stream ================================
        if constexpr (LS) {
            for (int layer = 0; layer < nlayers; layer++) {
                // layer 1 went out with layer 0; from then on layer l+1 is issued when the consumers have passed barrier l
                // (they are done with layer l-1, whose half it overwrites) and is waited for at barrier l+1
                if (layer == 0 && nlayers > 1) wait_vm_barrier<3 * WPL>();
                else wait_vm_barrier<0>();
                if (layer >= 1 && layer + 1 < nlayers) issue_layer(layer + 1);
            }
            return;
        }
        for (int t = 0; t < T_total; t++) {
            // step t's slab was issued two steps ago; only the slab of step t+1 may still be in flight
            if (t + 1 < T_total) wait_vm_barrier<WPL>();
            else wait_vm_barrier<0>();
            if (t + 2 < T_total) issue_w(t + 2);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ================================ consumer waves ================================
    const int r = lane & 31, h = lane >> 5;
    const int rb = CH == 2 ? wave >> 1 : 0;                   // 64-row block of this wave
    const int pb0 = CH == 4 ? wave >> 1 : 0;                  // first 32-pixel block of this wave inside its row block
    const int coutb = (wave & 1) * 32;                        // first output channel of this wave
    const int pslot0 = BIG ? (rb & 1) * 64 : 0;
    const int board_lds = BIG ? (rb >> 1) * 16384 : rb * 8192;
    // Activation fragment addresses, relative to the input buffer: per (kernel row, tap column, pixel block) the
    // byte offset of the shifted pixel's row (the buffer's zero row for pixels off the board) and the row's
    // swizzle term.  They depend on nothing but the lane, so they are made once; a step adds the buffer base.
    bool pvalid[NPB];
    int rel[3][3][NPB], swz[3][3][NPB];
#pragma unroll
    for (int pb = 0; pb < NPB; pb++) {
        const int p = pslot0 + (pb0 + pb) * 32 + r;
        const int ph = p / S, pw = p - ph * S;
        pvalid[pb] = p < S * S;
#pragma unroll
        for (int g = 0; g < 3; g++)
#pragma unroll
            for (int dxi = 0; dxi < 3; dxi++) {
                const int hh = ph + g - 1, ww = pw + dxi - 1;
                const bool ok = pvalid[pb] && (unsigned)hh < (unsigned)S && (unsigned)ww < (unsigned)S;
                const int q = hh * S + ww;
                // off the board: the zero area at the row's own address mod 256 (LS has one zero row only: slot 63)
                rel[g][dxi][pb] = ok ? board_lds + q * 128 : ZERO_OFF + (LS ? 0 : (q & 1) * 128);
                swz[g][dxi][pb] = ok || !LS ? ((h ^ ((q >> 1) & 7)) << 4) : 0;
            }
    }
    int aaddr[4][CB];
#pragma unroll
    for (int cb = 0; cb < CB; cb++) {
        const int row = coutb + cb * 32 + r;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) aaddr[ks][cb] = R_LDS_W + row * 128 + (((ks * 2 + h) ^ ((row >> 1) & 7)) << 4);
    }
    // LDS address of this lane's 4-cout group (cb, g4) in pixel row (pb) of a buffer: 8 bytes at
    // row*128 + swizzled 16-byte slot of the cout + h*8
    int eaddr[NPB];  // per pb: byte offset of the pixel row inside a buffer
    int eswz[NPB];
#pragma unroll
    for (int pb = 0; pb < NPB; pb++) {
        const int row = rb * 64 + (pb0 + pb) * 32 + r;
        eaddr[pb] = row * 128 + h * 8;
        eswz[pb] = (row >> 1) & 7;
    }

    // fragments are read AHEAD stages before the MFMAs that use them; a stage is CB * NPB MFMAs
    constexpr int AHEAD = CH == 2 ? 3 : 5, RING = AHEAD + 1;
    int opaque = 0;
    frag wa[4];       // head conv weights of this lane's head channel row
    f32x4 hbias[4];   // head conv bias of the 16 head channels this lane's accumulator holds
    for (int layer = 0; layer < nlayers; layer++) {
        const Tower64Layer L = A.layers[layer];
        // buffers: the stem reads 1 and writes 0; a block's first conv reads 0 and writes 1, its second reads 1,
        // adds 0 (the block input) and writes 0
        const int in_buf = (layer & 1) ? 0 : 1;
        const int ibase = R_LDS_ACT + in_buf * ACT_BYTES;
        const int obase = R_LDS_ACT + (1 - in_buf) * ACT_BYTES;
        // L.bias comes out of a table in memory, so the compiler cannot tell its address space: loaded through a
        // generic pointer these would be FLAT loads, which count in lgkmcnt as well and return out of order -- while
        // they are in flight (until the epilogue) every wait for an LDS fragment would have to be lgkmcnt(0).
        typedef const __attribute__((address_space(1))) f32x4* gf32x4p;
        const gf32x4p gbias = (gf32x4p)(L.bias + coutb + h * 4);
        f32x4 biasv[CB][4];
#pragma unroll
        for (int cb = 0; cb < CB; cb++)
#pragma unroll
            for (int g = 0; g < 4; g++) biasv[cb][g] = gbias[cb * 8 + g * 2];
        if (layer == nlayers - 1 && A.head_w) {  // operands of the fused head convs: on their way under the last layer
            const T* hw_ = reinterpret_cast<const T*>(A.head_w);
#pragma unroll
            for (int ks = 0; ks < 4; ks++) wa[ks] = *reinterpret_cast<const frag*>(hw_ + r * 64 + ks * 16 + h * 8);
#pragma unroll
            for (int q = 0; q < 4; q++) hbias[q] = *reinterpret_cast<const f32x4*>(A.head_b + 8 * q + 4 * h);
        }
        f32x16 acc[CB][NPB];
#pragma unroll
        for (int i = 0; i < CB; i++)
#pragma unroll
            for (int j = 0; j < NPB; j++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

        if constexpr (LS) {
            {
                STAMP_ACC_BEGIN;
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                STAMP_ACC_END(4);
            }
            if (layer == 0) STAMP(1);
            asm volatile("" : "+v"(opaque));
            const int wbase = (layer & 1) * 3 * V2_SLAB;
            int baddr[9][NPB][4];
#pragma unroll
            for (int t9 = 0; t9 < 9; t9++)
#pragma unroll
                for (int pb = 0; pb < NPB; pb++) {
                    const int rowa = ibase + rel[t9 / 3][t9 % 3][pb] + opaque;
#pragma unroll
                    for (int ks = 0; ks < 4; ks++) baddr[t9][pb][ks] = rowa + (swz[t9 / 3][t9 % 3][pb] ^ (ks << 5));
                }
            frag fa[RING][CB], fb[RING][NPB];
            auto load_stage = [&](int i, frag (&a)[CB], frag (&b)[NPB]) {  // i = tap * 4 + ks, tap = g * 3 + dxi: slab g, tap row dxi
                const int t9 = i >> 2, ks = i & 3;
#pragma unroll
                for (int cb = 0; cb < CB; cb++)
                    a[cb] = *reinterpret_cast<const frag*>(smem + aaddr[ks][cb] + (wbase + (t9 / 3) * V2_SLAB + (t9 % 3) * 8192));
#pragma unroll
                for (int pb = 0; pb < NPB; pb++) b[pb] = *reinterpret_cast<const frag*>(smem + baddr[t9][pb][ks]);
            };
#pragma unroll
            for (int i = 0; i < AHEAD; i++) load_stage(i, fa[i % RING], fb[i % RING]);
            __builtin_amdgcn_sched_group_barrier(0x100, (CB + NPB) * AHEAD, 0);
#pragma unroll
            for (int i = 0; i < 36; i++) {
                if (i + AHEAD < 36) load_stage(i + AHEAD, fa[(i + AHEAD) % RING], fb[(i + AHEAD) % RING]);
#pragma unroll
                for (int cb = 0; cb < CB; cb++)
#pragma unroll
                    for (int pb = 0; pb < NPB; pb++) Mfma<T>::mac(fa[i % RING][cb], fb[i % RING][pb], acc[cb][pb]);
                if (i + AHEAD < 36) __builtin_amdgcn_sched_group_barrier(0x100, CB + NPB, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, CB * NPB, 0);
            }
        } else {
#pragma unroll
        for (int g = 0; g < 3; g++) {
            // fragment reads of the previous step and (at a layer boundary) the previous layer's output writes
            // are done before anybody passes: the loaders may reuse the slab, the neighbours may read the rows
            {
                STAMP_ACC_BEGIN;
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                STAMP_ACC_END(4);
            }
            if (layer == 0 && g == 0) STAMP(1);
            asm volatile("" : "+v"(opaque));  // keeps the per-step addresses from being hoisted out of the layer loop
            const int wslab = g * V2_SLAB;
            int baddr[3][NPB][4];
#pragma unroll
            for (int dxi = 0; dxi < 3; dxi++)
#pragma unroll
                for (int pb = 0; pb < NPB; pb++) {
                    const int rowa = ibase + rel[g][dxi][pb] + opaque;
#pragma unroll
                    for (int ks = 0; ks < 4; ks++) baddr[dxi][pb][ks] = rowa + (swz[g][dxi][pb] ^ (ks << 5));
                }
            frag fa[RING][CB], fb[RING][NPB];
            auto load_stage = [&](int i, frag (&a)[CB], frag (&b)[NPB]) {
                const int dxi = i >> 2, ks = i & 3;
#pragma unroll
                for (int cb = 0; cb < CB; cb++)
                    a[cb] = *reinterpret_cast<const frag*>(smem + aaddr[ks][cb] + (wslab + dxi * 8192));
#pragma unroll
                for (int pb = 0; pb < NPB; pb++) b[pb] = *reinterpret_cast<const frag*>(smem + baddr[dxi][pb][ks]);
            };
#pragma unroll
            for (int i = 0; i < AHEAD; i++) load_stage(i, fa[i % RING], fb[i % RING]);
            __builtin_amdgcn_sched_group_barrier(0x100, (CB + NPB) * AHEAD, 0);
#pragma unroll
            for (int i = 0; i < 12; i++) {
                if (i + AHEAD < 12) load_stage(i + AHEAD, fa[(i + AHEAD) % RING], fb[(i + AHEAD) % RING]);
#pragma unroll
                for (int cb = 0; cb < CB; cb++)
#pragma unroll
                    for (int pb = 0; pb < NPB; pb++) Mfma<T>::mac(fa[i % RING][cb], fb[i % RING][pb], acc[cb][pb]);
                if (i + AHEAD < 12) __builtin_amdgcn_sched_group_barrier(0x100, CB + NPB, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, CB * NPB, 0);
            }
        }
        }

        // ---- layer epilogue: + bias (+ block input), ReLU, bf16, into the other buffer (same layout) ----
        // Pixel slots >= S*S are stored as they come: no conv ever reads them as a neighbour (the address table
        // sends off-board taps to the zero row) and the heads skip them.
        STAMP_ACC_BEGIN;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        int eoff[CB][NPB][4];
#pragma unroll
        for (int cb = 0; cb < CB; cb++)
#pragma unroll
            for (int pb = 0; pb < NPB; pb++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int slot16 = (coutb >> 3) + cb * 4 + g;  // 16-byte slot of couts coutb + cb*32 + g*8 .. +7
                    eoff[cb][pb][g] = obase + eaddr[pb] + ((slot16 ^ eswz[pb]) << 4);
                }
        // two code paths (uniform branch), packed f32 arithmetic: the MFMA pipes idle during an epilogue, so its
        // instruction count is what it costs
        typedef __attribute__((ext_vector_type(2))) float f32x2;
        auto finish = [&](const f32x16& a, int g, const f32x4& bv, const f32x2& r01, const f32x2& r23) {
            f32x2 v01 = f32x2{a[g * 4 + 0], a[g * 4 + 1]} + f32x2{bv[0], bv[1]};
            f32x2 v23 = f32x2{a[g * 4 + 2], a[g * 4 + 3]} + f32x2{bv[2], bv[3]};
            v01 = v01 + r01;
            v23 = v23 + r23;
            v01 = __builtin_elementwise_max(v01, f32x2{0.0f, 0.0f});
            v23 = __builtin_elementwise_max(v23, f32x2{0.0f, 0.0f});
            bf16x4 ov;
            ov[0] = (T)v01[0], ov[1] = (T)v01[1], ov[2] = (T)v23[0], ov[3] = (T)v23[1];
            return ov;
        };
        // LS: pixel slots the board does not have stay zero (slot 63 is the zero row of the next layer's taps)
        auto keep = [&](bf16x4 ov, int pb) {
            if constexpr (LS) {
                if (!pvalid[pb]) ov = bf16x4{(T)0.0f, (T)0.0f, (T)0.0f, (T)0.0f};
            }
            return ov;
        };
        if (L.res) {  // the block input lives in the buffer being overwritten: same addresses, all read first
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
            u32x2 rv[CB][NPB][4];
#pragma unroll
            for (int cb = 0; cb < CB; cb++)
#pragma unroll
                for (int pb = 0; pb < NPB; pb++)
#pragma unroll
                    for (int g = 0; g < 4; g++) rv[cb][pb][g] = *reinterpret_cast<const u32x2*>(smem + eoff[cb][pb][g]);
#pragma unroll
            for (int cb = 0; cb < CB; cb++)
#pragma unroll
                for (int pb = 0; pb < NPB; pb++)
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        const u32x2 w = rv[cb][pb][g];  // 4 bf16: a bf16 is the high half of the f32 with the same value
                        const f32x2 r01{__uint_as_float(w[0] << 16), __uint_as_float(w[0] & 0xffff0000u)};
                        const f32x2 r23{__uint_as_float(w[1] << 16), __uint_as_float(w[1] & 0xffff0000u)};
                        *reinterpret_cast<bf16x4*>(smem + eoff[cb][pb][g]) = keep(finish(acc[cb][pb], g, biasv[cb][g], r01, r23), pb);
                    }
        } else {
#pragma unroll
            for (int cb = 0; cb < CB; cb++)
#pragma unroll
                for (int pb = 0; pb < NPB; pb++)
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        // adding +0.0 leaves every value as it is (x + 0 == x, also for -0 + +0 = +0 before the ReLU)
                        *reinterpret_cast<bf16x4*>(smem + eoff[cb][pb][g]) =
                            keep(finish(acc[cb][pb], g, biasv[cb][g], f32x2{0.0f, 0.0f}, f32x2{0.0f, 0.0f}), pb);
                    }
        }
        STAMP_ACC_END(5);
    }
    STAMP(2);

    // ---- the two 1x1 head convs on the resident tower output (K3 fused; same MFMA order as head_conv_tile) ----
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the loader waves have left: live waves only
    const int fin = (nlayers & 1) ? 0 : 1;  // the stem and every second conv of a block write buffer 0
    if (A.head_w) {
        // D[i][px] = sum_k W[i][k] * X[px][k]: A = head weights [32][64] (rows >= ocn are zero), B = tower rows.
        // CH = 2: the two waves of a row block take one pixel block each; CH = 4: the first wave of each pixel block takes it.
        const int hwp = S * S;
        if (!(CH == 4 && (wave & 1))) {
            const int pb = CH == 2 ? (wave & 1) : pb0;
            const int row = rb * 64 + pb * 32 + r;
            f32x16 hacc;
#pragma unroll
            for (int e = 0; e < 16; e++) hacc[e] = 0.0f;
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {
                const frag xb = *reinterpret_cast<const frag*>(smem + R_LDS_ACT + fin * ACT_BYTES + row * 128 + (((ks * 2 + h) ^ ((row >> 1) & 7)) << 4));
                Mfma<T>::mac(wa[ks], xb, hacc);
            }
            const uint32_t grow = (uint32_t)(row0 + row);
            const uint32_t bb = grow / SLOTS_PER_BOARD, p = grow % SLOTS_PER_BOARD;
            if ((int)p < hwp) {
                T* hvp = reinterpret_cast<T*>(A.hv);
#pragma unroll
                for (int e = 0; e < 16; e++) {
                    const uint32_t i = (e & 3) + 8 * (e >> 2) + 4 * h;
                    if (i >= A.ocn) continue;
                    const float y = hacc[e] + hbias[e >> 2][e & 3];
                    const size_t at = i < A.vhc ? frag_packed_index<T>(bb, i * hwp + p, A.kvp)
                                                : A.hv_pol + frag_packed_index<T>(bb, (i - A.vhc) * hwp + p, A.kpp);
                    hvp[at] = (T)(y > 0.0f ? y : 0.0f);
                }
            }
        }
    }
    // ---- optional: the tower output itself -> HBM [rows][64], whole 128-byte rows ----
    if (A.out) {
        const int prow = lane >> 3, sl = lane & 7;
        constexpr int RPW = 64 / CH;  // rows this wave stores
        const int lrow0 = CH == 4 ? wave * 16 : rb * 64 + (CH == 2 ? (wave & 1) * 32 : 0);
        T* out = reinterpret_cast<T*>(A.out);
#pragma unroll
        for (int i = 0; i < RPW / 8; i++) {
            const int row = lrow0 + i * 8 + prow;
            const f32x4 v = *reinterpret_cast<const f32x4*>(smem + R_LDS_ACT + fin * ACT_BYTES + row * 128 + ((sl ^ ((row >> 1) & 7)) << 4));
            *reinterpret_cast<f32x4*>(out + ((size_t)row0 + row) * 64 + sl * 8) = v;
        }
    }
    STAMP(3);
    STAMP_FLUSH(wave);
}

void launch_tower64(const Tower64Args& args, uint32_t rows, int ch, bool layer_steps, hipStream_t st, hipEvent_t ev_start,
                    hipEvent_t ev_stop) {
    const bool big = tower_slots(args.S) == 128;
#define CATTUS_LAUNCH_T64_LS(CH, BIG, LS)                                                                       \
    hipExtLaunchKernelGGL((tower64_lds_kernel<CH, BIG, LS>), dim3(rows / (256 / CH)), dim3(512), tower64_lds_bytes(CH, LS), st, \
                          ev_start, ev_stop, 0, args)
#define CATTUS_LAUNCH_T64(CH, BIG) CATTUS_LAUNCH_T64_LS(CH, BIG, false)
    if (ch == 4 && !big) {
        if (layer_steps && args.S * args.S <= 63) CATTUS_LAUNCH_T64_LS(4, false, true);
        else CATTUS_LAUNCH_T64(4, false);
    } else {
        if (big) CATTUS_LAUNCH_T64(2, true);
        else CATTUS_LAUNCH_T64(2, false);
    }
#undef CATTUS_LAUNCH_T64
#undef CATTUS_LAUNCH_T64_LS
}

// The opt-in for > 64 KiB of dynamic LDS is a per-device function attribute.  Every variant gets it here, once per
// device, before anything is launched on that device: cattus_hip_create calls this under a lock, so two evaluation
// threads can never meet a variant whose attribute call is still on its way.
hipError_t prepare_device() {
    hipError_t err = conv_attrs_for<__bf16>();
    hipError_t e2 = conv_attrs_for<float>();
    if (err == hipSuccess) err = e2;
    e2 = conv_attrs_for<_Float16>();
    if (err == hipSuccess) err = e2;
    e2 = split_attrs();
    if (err == hipSuccess) err = e2;
    auto set = [&](const void* fn, int bytes) {
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess && err == hipSuccess) err = e;
    };
    set(reinterpret_cast<const void*>(&tower64_lds_kernel<4, false, true>), tower64_lds_bytes(4, true));
    set(reinterpret_cast<const void*>(&tower64_lds_kernel<4, false, false>), tower64_lds_bytes(4, false));
    set(reinterpret_cast<const void*>(&tower64_lds_kernel<2, true, false>), tower64_lds_bytes(2, false));
    set(reinterpret_cast<const void*>(&tower64_lds_kernel<2, false, false>), tower64_lds_bytes(2, false));
    const hipError_t e3 = prepare_tower64_split();
    if (err == hipSuccess) err = e3;
    const hipError_t e4 = prepare_wino();
    if (err == hipSuccess) err = e4;
    const hipError_t e5 = prepare_wino4();
    if (err == hipSuccess) err = e5;
    const hipError_t e6 = prepare_wino8();
    if (err == hipSuccess) err = e6;
    return err;
}

// ------------------------------------------------------------------------------------------
// K1g: generic direct conv, f32 NCHW, one thread per output element, canonical chain order
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) conv3x3_generic_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ res, float* __restrict__ out,
                                                              uint32_t total, int cin, int cout, int S) {
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int hw = S * S;
    const int p = e % hw, co = (e / hw) % cout, b = e / (hw * cout);
    const int ph = p / S, pw = p - ph * S;
    const float* xb = in + (size_t)b * cin * hw;
    float acc = 0.0f;
    const int nch = (cin + 31) / 32;
    for (int ch = 0; ch < nch; ch++)
        for (int tap = 0; tap < 9; tap++) {
            const int hh = ph + tap / 3 - 1, ww = pw + tap % 3 - 1;
            const bool ok = (unsigned)hh < (unsigned)S && (unsigned)ww < (unsigned)S;
            const int q = hh * S + ww;
            const float* wr = w + ((size_t)tap * cout + co) * cin;
            for (int kk = 0; kk < 32; kk++) {
                const int k = ch * 32 + (kk & ~7) + ((kk & 1) << 2) + ((kk & 7) >> 1);  // 0,4,1,5,2,6,3,7
                if (k >= cin) continue;
                const float x = ok ? xb[(size_t)k * hw + q] : 0.0f;
                acc = __builtin_fmaf(wr[k], x, acc);
            }
        }
    float y = acc + bias[co];
    if (res) y = y + res[e];
    out[e] = y > 0.0f ? y : 0.0f;
}

void launch_conv3x3_generic(const float* in, const float* w, const float* bias, const float* res, float* out,
                            uint32_t b, uint32_t cin, uint32_t cout, uint32_t S, hipStream_t st, hipEvent_t ev_start,
                            hipEvent_t ev_stop) {
    const uint32_t total = b * cout * S * S;
    if (!total) return;
    hipExtLaunchKernelGGL(conv3x3_generic_kernel, dim3((total + 255) / 256), dim3(256), 0, st, ev_start, ev_stop, 0, in, w,
                          bias, res, out, total, (int)cin, (int)cout, (int)S);
}

// ------------------------------------------------------------------------------------------
// K3-K5: heads
// ------------------------------------------------------------------------------------------
// Every dot product of the heads runs over k in 8-groups ascending and 0,4,1,5,2,6,3,7 inside a
// group -- the order an MFMA lane pair (k = 4h + j) produces -- in the MFMA kernels, in the SIMT
// kernels of the generic path, and in the CPU oracle alike.
__host__ __device__ constexpr uint32_t kperm(uint32_t kk) { return (kk & ~7u) + ((kk & 1u) << 2) + ((kk & 7u) >> 1); }

// ---- MFMA path: D[i][j] = sum_k P[i][k] * Q[j][k], K contiguous in both operands ------------
enum { EPI_FC1 = 1, EPI_POLICY = 2 };

struct HeadEpi {
    const float* bias;
    void* out;
    uint32_t hw, vhc, ocn, hv_pol, kvp, kpp, M, slots;
};

// ---- K3: the two 1x1 head convs as one GEMM, P = head weights [32][F], Q = tower rows [rows][F] (row-major) ----
// A block = 8 waves = a 32 (i) x 128 (j) strip: wave w < 4 owns j tile tj0 + w, waves 4-7 only stage operands.  The
// tower rows are what the conv kernel wrote, so the operands go through LDS in chunks of 256 bytes per row, fetched
// with whole-line loads (16 lanes per row; a fragment read straight from a row-major matrix touches 32 lines per
// instruction and uses a quarter of each: 18.9 us against 6.8 us for this launch, profiles/r02_experiments.txt),
// HG_DEPTH chunks ahead in registers.  The epilogue writes hv in fragment order for the FC launch.
constexpr int HG_ROWB = 256;                    // operand bytes per row and chunk: 128 bf16 / 64 f32 = 8 MFMA stages
constexpr int HG_PITCH = HG_ROWB + 16;          // LDS row pitch: ds_read_b128 down a column of rows is conflict-free
constexpr int HG_LDS = (32 + 128) * HG_PITCH;   // P rows 0..31, Q rows 32..159
constexpr int HG_THREADS = 512;                 // waves 0-3 own the four j tiles; all eight stage the operands
constexpr int HG_PIECES = (32 + 128) * (HG_ROWB / 16) / HG_THREADS;  // 16-byte pieces per thread and chunk (5)
constexpr int HG_DEPTH = 3;                     // chunks in flight per thread (registers)
constexpr int HG_GROUP = 8;                     // chunks per straight-line group (1024 bf16 / 512 f32 k)

template <typename T>
__device__ __forceinline__ void head_conv_tile(const T* __restrict__ P, uint32_t ldp, uint32_t I, const T* __restrict__ Q,
                                               uint32_t ldq, uint32_t J, uint32_t K, const HeadEpi& ep, uint32_t bx,
                                               uint32_t by, char* lds) {
    typedef typename Mfma<T>::frag frag;
    constexpr uint32_t KSTEP = 32 / sizeof(T);       // k per MFMA stage: 16 (bf16) or 8 (f32)
    constexpr uint32_t KCH = HG_ROWB / sizeof(T);    // k per chunk
    constexpr uint32_t NST = KCH / KSTEP;            // MFMA stages per chunk (8)
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t r = lane & 31, h = lane >> 5;
    const uint32_t i0 = by * 32, jb0 = bx * 128, j0 = jb0 + wave * 32;
    const bool active = wave < 4 && j0 < J;  // a wave without a tile still helps with the staging
    // this thread's pieces: piece q = tid + HG_THREADS t -> LDS row q >> 4 (P rows first), 16-byte column q & 15
    const char* src[HG_PIECES];
    uint32_t dst[HG_PIECES], colb[HG_PIECES];
#pragma unroll
    for (int t = 0; t < HG_PIECES; t++) {
        const uint32_t q = tid + HG_THREADS * t, row = q >> 4;
        colb[t] = (q & 15) * 16;
        src[t] = row < 32 ? reinterpret_cast<const char*>(P + (size_t)min(i0 + row, I - 1) * ldp)
                          : reinterpret_cast<const char*>(Q + (size_t)min(jb0 + row - 32, J - 1) * ldq);
        dst[t] = row * HG_PITCH + colb[t];
    }
    // K is a multiple of 16 elements on every caller (filters and head widths are padded by the evaluator), so a row
    // has at least two 16-byte pieces.  HG_DEPTH chunks are in flight in registers: the heads are a chain of L2
    // round trips (few blocks, operands cold), so the chunks a block will need are requested up front and refilled
    // as they are consumed.  The loads are unconditional (address clamped to the row's last piece; a piece beyond K
    // is zeroed when it is written to LDS): a load under a branch makes the compiler drain every outstanding load at
    // the loop head, which serialises the look-ahead.
    const uint32_t kbytes = K * sizeof(T), lastb = kbytes - 16;
    f32x4 pre[HG_DEPTH][HG_PIECES];
    auto fetch = [&](int s, uint32_t c) {
#pragma unroll
        for (int t = 0; t < HG_PIECES; t++)
            pre[s][t] = *reinterpret_cast<const f32x4*>(src[t] + min(c * HG_ROWB + colb[t], lastb));
    };
    auto put = [&](int s, uint32_t c) {
#pragma unroll
        for (int t = 0; t < HG_PIECES; t++) {
            f32x4 v = pre[s][t];
            if (c * HG_ROWB + colb[t] >= kbytes) v = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(lds + dst[t]) = v;
        }
    };
    // epilogue operands requested now, so that their round trip is not paid after the last MFMA
    const uint32_t j = j0 + r;
    f32x4 bias_i[4];
#pragma unroll
    for (int g = 0; g < 4; g++) bias_i[g] = *reinterpret_cast<const f32x4*>(ep.bias + 8 * g + 4 * h);  // 32 entries (padded)
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.0f;
    const uint32_t nchunks = (K + KCH - 1) / KCH;
    const char* arow = lds + r * HG_PITCH + h * 16;
    const char* brow = lds + (32 + wave * 32 + r) * HG_PITCH + h * 16;
    // Groups of HG_GROUP chunks as straight-line code (the look-ahead does not cross a group boundary: with loads in
    // flight around a loop's back edge the compiler drains all of them at the loop head)
    for (uint32_t g0 = 0; g0 < nchunks; g0 += HG_GROUP) {
#pragma unroll
        for (int s = 0; s < HG_DEPTH; s++) fetch(s, g0 + s);
#pragma unroll
        for (int q = 0; q < HG_GROUP; q++) {
            const uint32_t c = g0 + q;
            if (c >= nchunks) break;
            put(q % HG_DEPTH, c);
            __syncthreads();
            if (q + HG_DEPTH < HG_GROUP) fetch(q % HG_DEPTH, c + HG_DEPTH);
            if (active) {
                const uint32_t left = K - c * KCH;
                if (left >= KCH) {  // whole chunk: every fragment read is issued before the first MFMA needs one
                    frag fa[NST], fb[NST];
#pragma unroll
                    for (uint32_t u = 0; u < NST; u++) {
                        fa[u] = *reinterpret_cast<const frag*>(arow + u * 32);
                        fb[u] = *reinterpret_cast<const frag*>(brow + u * 32);
                    }
#pragma unroll
                    for (uint32_t u = 0; u < NST; u++) Mfma<T>::mac(fa[u], fb[u], acc);
                } else {
#pragma unroll
                    for (uint32_t u = 0; u < NST; u++)
                        if (u * KSTEP < left)
                            Mfma<T>::mac(*reinterpret_cast<const frag*>(arow + u * 32), *reinterpret_cast<const frag*>(brow + u * 32), acc);
                }
            }
            __syncthreads();
        }
    }
    if (!active) return;
    if (j >= J) return;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const uint32_t i = i0 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (i >= I) continue;
        // i = head channel (value rows first), j = tower row b*slots + p
        const uint32_t bb = j / ep.slots, p = j % ep.slots;
        if (i >= ep.ocn || p >= ep.hw) continue;
        const float y = acc[e] + bias_i[e >> 2][e & 3];
        const size_t at = i < ep.vhc ? frag_packed_index<T>(bb, i * ep.hw + p, ep.kvp)
                                     : ep.hv_pol + frag_packed_index<T>(bb, (i - ep.vhc) * ep.hw + p, ep.kpp);
        reinterpret_cast<T*>(ep.out)[at] = (T)(y > 0.0f ? y : 0.0f);
    }
}

template <typename T>
__global__ void __launch_bounds__(HG_THREADS) head_conv_kernel(const T* __restrict__ P, uint32_t ldp, uint32_t I,
                                                               const T* __restrict__ Q, uint32_t ldq, uint32_t J, uint32_t K,
                                                               HeadEpi ep) {
    __shared__ __attribute__((aligned(16))) char hg_lds[HG_LDS];
    head_conv_tile<T>(P, ldp, I, Q, ldq, J, K, ep, blockIdx.x, blockIdx.y, hg_lds);
}

// ---- K4 + K5: the two FC layers on fragment-packed operands ------------------------------------
// D[i][j] = sum_k P[i][k] * Q[j][k] with P (head activations, leaves) and Q (weights) both stored in MFMA fragment
// order (kernels.h): a wave's operand for k-step u is one contiguous KiB, so it is read straight into registers with
// one coalesced load per operand -- no LDS staging, no barrier -- HF_AHEAD steps ahead in a register ring.  The walk is
// unrolled in straight-line groups of HF_GROUP steps with unconditional loads (a step beyond K re-reads the last one
// and its MFMA is skipped): with loads in flight around a loop's back edge, or under a branch, the compiler waits for
// all of them; here it counts.  One wave per 32x32 tile, 4 waves = 128 j per block; MFMA order over k as everywhere.
constexpr int HF_AHEAD = 8;    // k-steps in flight per operand (2 x 8 x 4 VGPRs; 16 measured no better)
constexpr int HF_GROUP = 32;   // k-steps per straight-line group (512 bf16 / 256 f32 k); loads of a group's unused tail are issued all the same
template <typename T, int EPI>
__device__ __forceinline__ void head_fc_tile_packed(const T* __restrict__ P, uint32_t I, const T* __restrict__ Q, uint32_t J,
                                                    uint32_t K, const HeadEpi& ep, uint32_t bx, uint32_t by, float* stage) {
    typedef typename Mfma<T>::frag frag;
    constexpr uint32_t KSTEP = 32 / sizeof(T);  // k per MFMA stage: 16 (bf16) or 8 (f32)
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t r = lane & 31, h = lane >> 5;
    const uint32_t i0 = by * 32, jt = bx * 4 + wave, j0 = jt * 32;
    if (j0 >= J) return;
    const uint32_t nsteps = K / KSTEP;  // K is a multiple of 16 elements
    const char* pa = reinterpret_cast<const char*>(P) + ((size_t)by * nsteps * 64 + lane) * 16;
    const char* qb = reinterpret_cast<const char*>(Q) + ((size_t)jt * nsteps * 64 + lane) * 16;
    const uint32_t j = j0 + r;
    const float bias_j = ep.bias[min(j, J - 1)];  // requested now: its round trip is not paid after the last MFMA
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; e++) acc[e] = 0.0f;
    frag fa[HF_AHEAD], fb[HF_AHEAD];
    auto fetch = [&](int slot, uint32_t u) {
        const size_t off = (size_t)min(u, nsteps - 1) * 1024;
        fa[slot] = *reinterpret_cast<const frag*>(pa + off);
        fb[slot] = *reinterpret_cast<const frag*>(qb + off);
    };
    for (uint32_t u0 = 0; u0 < nsteps; u0 += HF_GROUP) {
        const uint32_t left = nsteps - u0;  // > 0
#pragma unroll
        for (int s = 0; s < HF_AHEAD; s++) fetch(s, u0 + s);
#pragma unroll
        for (int q = 0; q < HF_GROUP; q++) {
            if ((uint32_t)q < left) Mfma<T>::mac(fa[q % HF_AHEAD], fb[q % HF_AHEAD], acc);
            if (q + HF_AHEAD < HF_GROUP) fetch(q % HF_AHEAD, u0 + q + HF_AHEAD);
        }
    }
    if (j >= J) return;
#pragma unroll
    for (int e = 0; e < 16; e++) {
        const uint32_t i = i0 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (i >= I) continue;
        if constexpr (EPI == EPI_FC1) {
            const float y = acc[e] + bias_j;
            stage[(i - i0) * 129 + j] = y > 0.0f ? y : 0.0f;  // [32 leaves][128 hidden], padded rows
        } else {
            float y = acc[e] + bias_j;
            // non-finite logits -> f32::MIN (reference: engine/src/net/mod.rs:56-61)
            if (!(__builtin_fabsf(y) <= 3.40282347e+38f)) y = -3.40282347e+38f;
            reinterpret_cast<float*>(ep.out)[(size_t)i * ep.M + j] = y;
        }
    }
}

// Value FC1 and policy FC read the same head activations and do not depend on each other: one launch, block column 0
// does FC1 (128 hidden units = one strip), the others the policy FC (same tile code, same arithmetic).  What the first
// loads need travels as plain kernel arguments (preloaded into SGPRs with the wave); the rest, in the struct, is
// fetched by a scalar load that nothing waits on until the epilogue.
struct HeadFcTail {
    HeadEpi ep1, ep2;
    const float *w2, *b2;
    float* value;
};
__device__ __forceinline__ float tanh_exact(float x);
template <typename T>
__global__ void __launch_bounds__(256) head_fc_pair_kernel(const T* __restrict__ p1, const T* __restrict__ q1,
                                                           const T* __restrict__ p2, const T* __restrict__ q2, uint32_t I,
                                                           uint32_t K1, uint32_t J2, uint32_t K2, HeadFcTail a) {
    if (blockIdx.x == 0) {
        // The block's four waves hold all 128 hidden units of 32 leaves: stage them in LDS and finish the
        // value head here (FC2 + tanh, the fmaf chain of value_fc2_tanh_kernel), saving a launch.
        __shared__ float h1s[32 * 129];
        head_fc_tile_packed<T, EPI_FC1>(p1, I, q1, 128, K1, a.ep1, 0, blockIdx.y, h1s);
        __syncthreads();
        const uint32_t leaf = blockIdx.y * 32 + threadIdx.x;
        if (threadIdx.x < 32 && leaf < I) {
            float hrow[128];
#pragma unroll
            for (uint32_t k = 0; k < 128; k++) hrow[k] = h1s[threadIdx.x * 129 + k];
            float acc = 0.0f;
#pragma unroll
            for (uint32_t kk = 0; kk < 128; kk++) {
                const uint32_t k = kperm(kk);
                acc = __builtin_fmaf(a.w2[k], hrow[k], acc);
            }
            a.value[leaf] = tanh_exact(acc + a.b2[0]);
        }
    } else {
        head_fc_tile_packed<T, EPI_POLICY>(p2, I, q2, J2, K2, a.ep2, blockIdx.x - 1, blockIdx.y, nullptr);
    }
}

static void launch_head_conv(Act act, const void* P, uint32_t ldp, uint32_t I, const void* Q, uint32_t ldq, uint32_t J,
                             uint32_t K, const HeadEpi& ep, hipStream_t st) {
    if (!I || !J) return;
    const dim3 grid(((J + 31) / 32 + 3) / 4, (I + 31) / 32), block(HG_THREADS);
    if (act == Act::BF16)
        hipLaunchKernelGGL(head_conv_kernel<__bf16>, grid, block, 0, st, (const __bf16*)P, ldp, I, (const __bf16*)Q, ldq, J, K, ep);
    else
        hipLaunchKernelGGL(head_conv_kernel<float>, grid, block, 0, st, (const float*)P, ldp, I, (const float*)Q, ldq, J, K, ep);
}

void launch_heads_mfma(Act act, const void* tower, uint32_t nb, uint32_t F, const HeadsMfma& hd, hipStream_t st) {
    HeadEpi ep{};
    ep.hw = hd.hw, ep.vhc = hd.vhc, ep.ocn = hd.vhc + hd.phc, ep.kvp = hd.kvp, ep.kpp = hd.kpp, ep.M = hd.M;
    ep.hv_pol = hd.hv_leaves * hd.kvp;
    ep.slots = hd.slots;
    // K3: both 1x1 convs in one GEMM: i = head channel, j = tower row (tower == nullptr: hv was already written
    // by the resident tower kernel, which fuses this step); writes hv in fragment order
    ep.bias = hd.conv_b, ep.out = hd.hv;
    if (tower) launch_head_conv(act, hd.conv_w, F, 32, tower, F, nb * hd.slots, F, ep, st);
    // K4 + K5 in one launch: value FC1 (+ReLU): i = leaf, j = hidden unit; policy FC: i = leaf, j = move
    const size_t esz = act == Act::BF16 ? 2 : 4;
    if (!nb) return;
    HeadFcTail a{};
    a.ep1 = ep, a.ep1.bias = hd.b1, a.ep1.out = hd.h1;
    a.ep2 = ep, a.ep2.bias = hd.bp, a.ep2.out = hd.policy;
    a.w2 = hd.w2, a.b2 = hd.b2, a.value = hd.value;
    const void* p2 = (const char*)hd.hv + (size_t)ep.hv_pol * esz;
    const dim3 grid(1 + ((hd.M + 31) / 32 + 3) / 4, (nb + 31) / 32), block(256);
    if (act == Act::BF16)
        hipLaunchKernelGGL(head_fc_pair_kernel<__bf16>, grid, block, 0, st, (const __bf16*)hd.hv, (const __bf16*)hd.w1,
                           (const __bf16*)p2, (const __bf16*)hd.wp, nb, hd.kvp, hd.M, hd.kpp, a);
    else
        hipLaunchKernelGGL(head_fc_pair_kernel<float>, grid, block, 0, st, (const float*)hd.hv, (const float*)hd.w1,
                           (const float*)p2, (const float*)hd.wp, nb, hd.kvp, hd.M, hd.kpp, a);
}

// ---- SIMT path (generic tower layout, any shape), same term order ---------------------------
template <typename T>
__global__ void __launch_bounds__(256) head_conv1x1_kernel(const T* __restrict__ x, uint32_t sb, uint32_t sk, uint32_t sp,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           uint32_t total, uint32_t F, uint32_t ocn, uint32_t hw,
                                                           float* __restrict__ hv) {
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const uint32_t p = e % hw, oc = (e / hw) % ocn, b = e / (hw * ocn);
    const T* xp = x + (size_t)b * sb + (size_t)p * sp;
    const float* wr = w + (size_t)oc * F;
    float acc = 0.0f;
    const uint32_t F8 = (F + 7) & ~7u;
    for (uint32_t kk = 0; kk < F8; kk++) {
        const uint32_t k = kperm(kk);
        if (k < F) acc = __builtin_fmaf(wr[k], (float)xp[(size_t)k * sk], acc);
    }
    const float y = acc + bias[oc];
    hv[e] = y > 0.0f ? y : 0.0f;
}

void launch_head_conv1x1(TowerView x, const float* w, const float* bias, uint32_t b, uint32_t F, uint32_t ocn,
                         uint32_t hw, float* hv, hipStream_t st) {
    const uint32_t total = b * ocn * hw;
    if (!total) return;
    const dim3 grid((total + 255) / 256), block(256);
    if (x.act == Act::BF16)
        hipLaunchKernelGGL(head_conv1x1_kernel<__bf16>, grid, block, 0, st, (const __bf16*)x.x, x.sb, x.sk, x.sp, w, bias,
                           total, F, ocn, hw, hv);
    else
        hipLaunchKernelGGL(head_conv1x1_kernel<float>, grid, block, 0, st, (const float*)x.x, x.sb, x.sk, x.sp, w, bias,
                           total, F, ocn, hw, hv);
}

__global__ void __launch_bounds__(128) value_fc1_kernel(const float* __restrict__ hv, uint32_t hv_stride,
                                                        const float* __restrict__ w1t, const float* __restrict__ b1,
                                                        uint32_t K, float* __restrict__ h1) {
    const uint32_t b = blockIdx.x, j = threadIdx.x;
    const float* x = hv + (size_t)b * hv_stride;
    float acc = 0.0f;
    const uint32_t K8 = (K + 7) & ~7u;
    for (uint32_t kk = 0; kk < K8; kk++) {
        const uint32_t k = kperm(kk);
        if (k < K) acc = __builtin_fmaf(w1t[(size_t)k * 128 + j], x[k], acc);
    }
    const float y = acc + b1[j];
    h1[(size_t)b * 128 + j] = y > 0.0f ? y : 0.0f;
}

void launch_value_fc1(const float* hv, uint32_t hv_stride, const float* w1t, const float* b1, uint32_t b, uint32_t K,
                      float* h1, hipStream_t st) {
    if (!b) return;
    hipLaunchKernelGGL(value_fc1_kernel, dim3(b), dim3(128), 0, st, hv, hv_stride, w1t, b1, K, h1);
}

// tanh from + - * / fmaf and exponent-bit edits only, so the CPU oracle reproduces it bit for bit.
__device__ __forceinline__ float exp_pos(float x) {
    const float log2e = 1.44269504088896341f;
    const float ln2_hi = 0.693145751953125f;
    const float ln2_lo = 1.42860682030941723e-06f;
    const float nf = __builtin_rintf(x * log2e);
    float rr = __builtin_fmaf(nf, -ln2_hi, x);
    rr = __builtin_fmaf(nf, -ln2_lo, rr);
    float p = 1.0f / 720.0f;
    p = __builtin_fmaf(p, rr, 1.0f / 120.0f);
    p = __builtin_fmaf(p, rr, 1.0f / 24.0f);
    p = __builtin_fmaf(p, rr, 1.0f / 6.0f);
    p = __builtin_fmaf(p, rr, 0.5f);
    p = __builtin_fmaf(p, rr, 1.0f);
    p = __builtin_fmaf(p, rr, 1.0f);
    return __uint_as_float(__float_as_uint(p) + (((uint32_t)(int32_t)nf) << 23));
}

__device__ __forceinline__ float tanh_exact(float x) {
    const float ax = __builtin_fabsf(x);
    float t;
    if (!(ax == ax)) return x;
    if (ax < 0.5f) {
        const float s = ax * ax;
        float p = 21844.0f / 6081075.0f;
        p = __builtin_fmaf(p, s, -1382.0f / 155925.0f);
        p = __builtin_fmaf(p, s, 62.0f / 2835.0f);
        p = __builtin_fmaf(p, s, -17.0f / 315.0f);
        p = __builtin_fmaf(p, s, 2.0f / 15.0f);
        p = __builtin_fmaf(p, s, -1.0f / 3.0f);
        t = __builtin_fmaf(ax * s, p, ax);
    } else if (ax < 10.0f) {
        const float e = exp_pos(2.0f * ax);
        t = 1.0f - 2.0f / (e + 1.0f);
    } else {
        t = 1.0f;
    }
    return x < 0.0f ? -t : t;
}

__global__ void __launch_bounds__(256) value_fc2_tanh_kernel(const float* __restrict__ h1, const float* __restrict__ w2,
                                                             const float* __restrict__ b2, uint32_t nb,
                                                             float* __restrict__ value) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nb) return;
    const float* x = h1 + (size_t)b * 128;
    float acc = 0.0f;
    for (uint32_t kk = 0; kk < 128; kk++) {
        const uint32_t j = kperm(kk);
        acc = __builtin_fmaf(w2[j], x[j], acc);
    }
    value[b] = tanh_exact(acc + b2[0]);
}

void launch_value_fc2_tanh(const float* h1, const float* w2, const float* b2, uint32_t b, float* value, hipStream_t st) {
    if (!b) return;
    hipLaunchKernelGGL(value_fc2_tanh_kernel, dim3((b + 255) / 256), dim3(256), 0, st, h1, w2, b2, b, value);
}

// exp for x <= 0 from the same operations as exp_pos (arguments below -80 give 0), so that the
// legal-move softmax below is reproducible bit for bit on the host.
__device__ __forceinline__ float exp_nonpos(float x) { return x < -80.0f ? 0.0f : exp_pos(x); }

// calc_moves_probs (engine/src/net/mod.rs:100-119) for one leaf per wave: gather the logits of the
// leaf's legal moves, subtract their maximum, exponentiate, and divide by the sum taken in move
// order (the order of the reference's iterator sum).
__global__ void __launch_bounds__(64) legal_softmax_kernel(const float* __restrict__ policy, uint32_t M,
                                                           const uint16_t* __restrict__ idx,
                                                           const uint16_t* __restrict__ cnt, uint32_t L,
                                                           float* __restrict__ probs) {
    __shared__ float es[1024];
    __shared__ float total;
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    const uint32_t n = cnt[b] < L ? cnt[b] : L;
    const float* row = policy + (size_t)b * M;
    const uint16_t* ix = idx + (size_t)b * L;
    float mx = -3.40282347e+38f;  // fold(f32::MIN, f32::max)
    for (uint32_t i = lane; i < n; i += 64) {
        const float v = row[ix[i]];
        es[i] = v;
        mx = v > mx ? v : mx;
    }
    for (int off = 32; off; off >>= 1) {
        const float o = __shfl_xor(mx, off);
        mx = o > mx ? o : mx;
    }
    for (uint32_t i = lane; i < n; i += 64) es[i] = exp_nonpos(es[i] - mx);
    __syncthreads();
    if (lane == 0) {
        float s = 0.0f;
        for (uint32_t i = 0; i < n; i++) s += es[i];
        total = s;
    }
    __syncthreads();
    const float s = total;
    float* out = probs + (size_t)b * L;
    for (uint32_t i = lane; i < L; i += 64) out[i] = i < n ? es[i] / s : 0.0f;
}

int launch_legal_softmax(const float* policy, uint32_t M, const uint16_t* idx, const uint16_t* cnt, uint32_t L, uint32_t b,
                         float* probs, hipStream_t st) {
    if (L > 1024) return -1;
    if (!b) return 0;
    hipLaunchKernelGGL(legal_softmax_kernel, dim3(b), dim3(64), 0, st, policy, M, idx, cnt, L, probs);
    return 0;
}

// Block = 256 logits x 8 leaves; the 8 input rows are staged in LDS, every weight is loaded once
// (coalesced along m) and used for 8 fmaf.
constexpr int PFC_ROWS = 8;
enum { DENSE_SCRUB = 0, DENSE_RELU = 1, DENSE_TANH = 2 };  // what follows `+ bias`
template <int EPI>
__global__ void __launch_bounds__(256) policy_fc_kernel(const float* __restrict__ hv, uint32_t hv_stride, uint32_t off,
                                                        const float* __restrict__ wpt, const float* __restrict__ bp,
                                                        uint32_t nb, uint32_t K, uint32_t M, float* __restrict__ policy) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* xs = reinterpret_cast<float*>(smem);  // [PFC_ROWS][K]
    const uint32_t b0 = blockIdx.y * PFC_ROWS;
    for (uint32_t i = threadIdx.x; i < PFC_ROWS * K; i += 256) {
        const uint32_t rr = i / K, k = i - rr * K;
        xs[i] = (b0 + rr < nb) ? hv[(size_t)(b0 + rr) * hv_stride + off + k] : 0.0f;
    }
    __syncthreads();
    const uint32_t m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    float acc[PFC_ROWS];
#pragma unroll
    for (int rr = 0; rr < PFC_ROWS; rr++) acc[rr] = 0.0f;
    const uint32_t K8 = (K + 7) & ~7u;
    for (uint32_t kk = 0; kk < K8; kk++) {
        const uint32_t k = kperm(kk);
        if (k >= K) continue;
        const float wv = wpt[(size_t)k * M + m];
#pragma unroll
        for (int rr = 0; rr < PFC_ROWS; rr++) acc[rr] = __builtin_fmaf(wv, xs[rr * K + k], acc[rr]);
    }
    const float bias = bp[m];
#pragma unroll
    for (int rr = 0; rr < PFC_ROWS; rr++) {
        if (b0 + rr >= nb) break;
        float y = acc[rr] + bias;
        if constexpr (EPI == DENSE_SCRUB) {
            // non-finite logits -> f32::MIN (reference: engine/src/net/mod.rs:56-61)
            if (!(__builtin_fabsf(y) <= 3.40282347e+38f)) y = -3.40282347e+38f;
        } else if constexpr (EPI == DENSE_RELU) {
            y = y > 0.0f ? y : 0.0f;
        } else {
            y = tanh_exact(y);
        }
        policy[(size_t)(b0 + rr) * M + m] = y;
    }
}

void launch_policy_fc(const float* hv, uint32_t hv_stride, uint32_t off, const float* wpt, const float* bp, uint32_t b,
                      uint32_t K, uint32_t M, float* policy, hipStream_t st) {
    if (!b) return;
    const dim3 grid((M + 255) / 256, (b + PFC_ROWS - 1) / PFC_ROWS), block(256);
    hipLaunchKernelGGL(policy_fc_kernel<DENSE_SCRUB>, grid, block, PFC_ROWS * K * sizeof(float), st, hv, hv_stride, off, wpt, bp, b, K,
                       M, policy);
}

// A dense layer of SimpleTwoHeadedModel (net_utils.py:92-121) on the same kernel: y[b][n] = epi(sum_k wt[k][n] x[b][k] + bias[n])
void launch_dense(const float* x, uint32_t x_stride, const float* wt, const float* bias, uint32_t b, uint32_t K, uint32_t N,
                  float* y, int epi, hipStream_t st) {
    if (!b) return;
    const dim3 grid((N + 255) / 256, (b + PFC_ROWS - 1) / PFC_ROWS), block(256);
    const size_t lds = PFC_ROWS * K * sizeof(float);
    if (epi == DENSE_RELU) hipLaunchKernelGGL(policy_fc_kernel<DENSE_RELU>, grid, block, lds, st, x, x_stride, 0u, wt, bias, b, K, N, y);
    else if (epi == DENSE_TANH) hipLaunchKernelGGL(policy_fc_kernel<DENSE_TANH>, grid, block, lds, st, x, x_stride, 0u, wt, bias, b, K, N, y);
    else hipLaunchKernelGGL(policy_fc_kernel<DENSE_SCRUB>, grid, block, lds, st, x, x_stride, 0u, wt, bias, b, K, N, y);
}

// ------------------------------------------------------------------------------------------
// Diagnostic: what the matrix pipe sustains (cattus_hip_mfma_sustained; bench.py's roofline.sustained)
// ------------------------------------------------------------------------------------------
// Nothing but back-to-back MFMAs of the tower's kind -- four independent accumulators per wave, operands in registers, no
// memory traffic -- one wave per SIMD on every CU.  The nominal peaks (2.5 PFLOP/s f16 / bf16, 157 TFLOP/s f32) are 2.4 GHz
// figures; under MFMA load the part's power management sets the clock, and this is the rate that leaves
// (scripts/probes/mfma_peak_probe.hip is the stand-alone form; profiles/r03_mfma_peak_probe.txt).
template <int KIND>  // 0: v_mfma_f32_32x32x16_f16, 1: ..._bf16, 2: v_mfma_f32_32x32x2_f32
__global__ void __launch_bounds__(256) mfma_sustain_kernel(int iters, float* __restrict__ out) {
    // four different operand pairs of pseudo-random values in [-1, 1): consecutive MFMAs see different inputs, as a conv's do
    // (the same pair over and over toggles nothing between instructions and flatters the power)
    uint32_t x = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    auto rnd = [&]() {
        x = x * 1664525u + 1013904223u;
        return (float)(int)(x >> 8) * (1.0f / 8388608.0f) - 1.0f;
    };
    f16x8 a[4], b[4];
    bf16x8 ab[4], bb[4];
    float af[4], bf[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float u = rnd(), v = rnd();
            a[q][i] = (_Float16)u, b[q][i] = (_Float16)v, ab[q][i] = (__bf16)u, bb[q][i] = (__bf16)v;
        }
        af[q] = rnd(), bf[q] = rnd();
    }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int it = 0; it < iters; it++) {
        if constexpr (KIND == 0) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[1], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2], b[2], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[3], b[3], c3, 0, 0, 0);
        } else if constexpr (KIND == 1) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[0], bb[0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[1], bb[1], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[2], bb[2], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[3], bb[3], c3, 0, 0, 0);
        } else {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[0], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[1], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[2], bf[2], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[3], bf[3], c3, 0, 0, 0);
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) sum += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

// One launch: `cus` workgroups of four waves, iters x 4 MFMAs per wave; returns the FLOPs of the launch.
double launch_mfma_sustain(Act act, int cus, int iters, float* out, hipStream_t st) {
    const dim3 grid(cus), block(256);
    if (act_f16_family(act)) hipLaunchKernelGGL(mfma_sustain_kernel<0>, grid, block, 0, st, iters, out);
    else if (act == Act::BF16) hipLaunchKernelGGL(mfma_sustain_kernel<1>, grid, block, 0, st, iters, out);
    else hipLaunchKernelGGL(mfma_sustain_kernel<2>, grid, block, 0, st, iters, out);
    const double flop_per_mfma = act == Act::F32 ? 2.0 * 32 * 32 * 2 : 2.0 * 32 * 32 * 16;
    return (double)cus * 4 * iters * 4 * flop_per_mfma;
}

}  // namespace cattus
