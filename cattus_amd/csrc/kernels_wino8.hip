// gfx950 kernels of the Cattus leaf evaluator, part 5: the Winograd F(2x2, 3x3) split conv with TWO waves per SIMD (K1w8).
//
// K1w4 (kernels_wino4.hip) holds 256 accumulator registers per wave, so a SIMD runs one wave, and a single wave issues in order: the
// ~350 instructions a k-step carries besides its 48 MFMAs (transform, U ring, activation chunk) do not hide in the MFMAs' shadow --
// its loop takes 41 k cycles where the MFMAs need 24.6 k, and without its MFMAs it still takes 32.6 k (scripts/probes/w4_variant.py).
// Here the same workgroup tile (4 boards x 64 couts x 16 frequencies) is cut over EIGHT waves of 128 accumulator registers:
//   wave w = (q = w & 3: frequency row, lh = w >> 2: frequency columns 2 lh, 2 lh + 1), 2 tile blocks x 2 cout blocks each;
//   waves w and w + 4 share a SIMD: while one makes its V (VALU, LDS reads) the other multiplies (MFMA), a wave parked at a wait
//   leaves the SIMD to its partner.  No slices between MFMAs, no run-ahead, V single-buffered: a k-step of a wave is
//   [make V of this k-step: 24 ds_read_b128, 128 VALU] [24 MFMAs in four stages, the U stages' refills behind their last use].
//   U: the wave's two frequencies of both cout blocks, K1w's fragment order, a ring of one k-step (2 stages x 4 loads = 32 registers);
//   each byte of the workgroup's 1 MB of U is loaded by exactly one wave, as in K1w4.
//   V in registers in MFMA operand layout as in K1w4; a wave needs three of the patch's four columns (lh = 0: 0, 1, 2; lh = 1: 1, 2, 3),
//   so the row combination of a column pair is made by both waves of a pair: 128 VALU per wave and k-step against K1w4's 224 per 48 MFMAs.
//   d: K1w4's chunk images (LDS-DMA, 5 pieces per wave and chunk), a chunk is read in the k-steps it belongs to: one barrier per chunk.
//   epilogue, per tile block: the lh = 1 waves hand their M to their lh = 0 partners through LDS, those make Z = (row q of M) A in K1w's
//   order and put it where K1w4's exchange has it, all eight waves make Y = A^T Z, + bias, + skip, ReLU, cap, store.
// Per accumulator the MFMA sequence is K1w's, V, Z and Y are combined in K1w's order: the same bits as K1w and K1w4
// (tests/test_hip_parity.py::test_winograd_kernels_agree_bit_for_bit).
//
// Built, measured, NOT the default (round 5; diagnostic switch CATTUS_WINO_KERNEL=k8, scripts/probes/stamps_w8.py,
// profiles/r05_w4_anatomy.txt): 124 + 128 registers, no scratch, the same bits on the first run.  Cycles per wave and layer at chess
// 256 -> 256, batch 256: prologue 4.9 k, loop 37.8-38.9 k (K1w4: 41-43 k), epilogue 9.8-11.3 k (two exchanges per tile block: K1w4
// 7.5 k) = 53.6-54.0 k against K1w4's 56-58 k -- and 30.5-31.5 us per layer against K1w4's 28.8-29.2 on the same boxes: the part
// answers the denser kernel with a lower clock (1.74 GHz in-kernel against 1.9-1.95).  Per SIMD a k-step still takes 2,430 cycles =
// the 48 MFMAs' 1,536 + ~3.5 cycles for each of the pair's 256 VALU instructions: with the halves skewed (below) or in lockstep
// (30.5 us), the transform's VALU work does not disappear under the partner's MFMAs.  The tower is power-limited before it is
// issue-limited; what would help is fewer instructions per product, not more waves.
#include "kernels.h"
#include "device_common.h"

#include <hip/hip_ext.h>

namespace cattus {

constexpr int W8_RP = 8 * SP + 64;            // K1w4's chunk image: board-row pitch,
constexpr int W8_IMG = 16 * W8_RP;            // one tile block's image (2 boards),
constexpr int W8_ZAREA = 4608;                // its zero area,
constexpr int W8_IMGZ = W8_IMG + W8_ZAREA;
constexpr int W8_DBUF = 2 * W8_IMGZ;          // a chunk buffer (both tile blocks): 48,128 B
constexpr int W8_NBUF = 3;                    // chunk buffers (three: the halves of a workgroup meet the chunk barrier half an iteration apart)
constexpr int W8_LDS_LOOP = W8_NBUF * W8_DBUF;  // 144,384 B
constexpr int W8_LDS_M = 4 * 64 * 256;        // epilogue region A: the lh = 1 waves' M of one tile block, [q][x4 index 0..15][lane][16 B] = 65,536 B
constexpr int W8_LDS_Z = 4 * 2 * 32 * 256;    // region B: Z of one tile block, [q][c'][tile 0..31][64 couts f32] = 65,536 B
constexpr int W8_LDS_TOTAL = (W8_LDS_M + W8_LDS_Z) > W8_LDS_LOOP ? (W8_LDS_M + W8_LDS_Z) : W8_LDS_LOOP;  // 144,384 B
constexpr int W8_P = 5;                       // LDS-DMA pieces per wave and chunk: 8 x 5 = 40 >= 2 images x 19 KiB pieces
static_assert(W8_IMG % 256 == 0 && W8_IMGZ % 256 == 0 && W8_DBUF % 256 == 0, "the zero area keeps a read's banks only if everything is 256-B aligned");

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"  // "clobber list contains reserved registers: m0": intended (kernels_wino4.hip)
__device__ __forceinline__ void w8_glds16(const char* gsrc, uint32_t lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_dst) : "memory", "m0");
}
#pragma clang diagnostic pop

typedef __attribute__((ext_vector_type(2))) _Float16 w8_f16x2;
#ifdef CATTUS_STAMPS
__device__ unsigned long long g_stamps_w8[1024 * 8 * 4];
#define W8_STAMP(i) st_[i] = __builtin_amdgcn_s_memtime()
#else
#define W8_STAMP(i)
#endif
template <int V>
using w8_int = std::integral_constant<int, V>;

template <bool HAS_RES>
__global__ void __launch_bounds__(512, 2)
    conv3x3_wino8_kernel(const float* __restrict__ in, const _Float16* __restrict__ wu, const float* __restrict__ bias,
                         const float* res, float* out, unsigned* __restrict__ sat, int cin, int cout) {
    typedef _Float16 T;
    typedef Mfma<T>::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];

#ifdef CATTUS_STAMPS
    unsigned long long st_[4] = {0, 0, 0, 0};
#endif
    W8_STAMP(0);
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = w & 3, lh = w >> 2;  // frequency row; frequency columns 2 lh, 2 lh + 1
    const int lane = tid & 63;

    const int nblk = gridDim.x, ncg = cout >> 6;
    int logical = blockIdx.x;
    if ((nblk & 7) == 0) logical = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);  // one XCD: all cout groups of a range of boards
    const int cout0 = (logical % ncg) * 64;
    const int row0 = (logical / ncg) * 256;
    const int nch = cin >> 5, nks = cin >> 4;
    const uint32_t row_bytes = (uint32_t)cin * 4;

    // ---- the U ring: slot j = the U stage (k-step, frequency 4 q + 2 lh + j) = [cb0 hi, cb0 lo, cb1 hi, cb1 lo] ----
    const char* wb0 = reinterpret_cast<const char*>(wu) + ((size_t)(cout0 >> 5) * nks * 16 + 4 * q + 2 * lh) * SW_STAGE;
    const size_t wcb = (size_t)nks * 16 * SW_STAGE;
    const uint32_t voff0 = lane * 16;
    u32x4 ring[2][4];
#define W8_ULOAD(OFF)                                                                                                                      \
    asm volatile("global_load_dwordx4 %0, %4, %5 offset:" #OFF "\n\tglobal_load_dwordx4 %1, %4, %5 offset:" #OFF "+1024\n\t"             \
                 "global_load_dwordx4 %2, %4, %6 offset:" #OFF "\n\tglobal_load_dwordx4 %3, %4, %6 offset:" #OFF "+1024"                  \
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)                                                                                  \
                 : "v"(voff0), "s"(p0), "s"(p1)                                                                                            \
                 : "memory")
    auto load_ustage = [&](u32x4(&slot)[4], const char* p0, int j) __attribute__((always_inline)) {
        const char* p1 = p0 + wcb;
        u32x4 a, b, c, d;
        if (j) W8_ULOAD(2048);
        else W8_ULOAD(0);
        slot[0] = a, slot[1] = b, slot[2] = c, slot[3] = d;
    };
#undef W8_ULOAD

    // ---- LDS-DMA of a chunk: two images (K1w4's layout) = 38 pieces of 1 KiB, 5 per wave (the last wave's run stops at 37) ----
    const char* abase0 = reinterpret_cast<const char*>(in) + (size_t)row0 * row_bytes;
    uint32_t off_a[W8_P];
#pragma unroll
    for (int i = 0; i < W8_P; i++) {
        const int id = min(w * W8_P + i, 37), img = id >= 19 ? 1 : 0, pid = id - 19 * img;
        const int sidx = pid * 64 + lane, brow = sidx / 76, ww = sidx - brow * 76, wsh = max(ww - (brow >> 3), 0);
        const int x = min(wsh / 9, 7), c = min(wsh - (wsh / 9) * 9, 7);
        off_a[i] = (uint32_t)(img * 128 + brow * 8 + x) * row_bytes + c * 16;
    }
    auto issue_chunk = [&](int ch, uint32_t dbuf) __attribute__((always_inline)) {
        const char* src = abase0 + (size_t)ch * 128;
#pragma unroll
        for (int i = 0; i < W8_P; i++) {
            const int id = min(w * W8_P + i, 37), img = id >= 19 ? 1 : 0, pid = id - 19 * img;
            w8_glds16(src + off_a[i], dbuf + (uint32_t)(img * W8_IMGZ + pid * 1024));
        }
    };
    for (int i = tid; i < 2 * W8_NBUF * (W8_ZAREA / 16); i += 512)  // the six zero areas
        reinterpret_cast<f32x4*>(smem + (i / (W8_ZAREA / 16)) * W8_IMGZ + W8_IMG)[i % (W8_ZAREA / 16)] = f32x4{0.f, 0.f, 0.f, 0.f};
    issue_chunk(0, 0);
    issue_chunk(1, W8_DBUF);  // cin >= 128: at least four chunks
    const char* wks = wb0;  // U of the k-step being multiplied
    load_ustage(ring[0], wks, 0);
    load_ustage(ring[1], wks, 1);

    // ---- the transform's geometry (K1w4's): lane = (tile n of the tile block, k-half hh) ----
    // the wave combines patch rows (ra, rb) as t = fma(d[rb], sg, d[ra]) and needs three columns u0, u1, u2 of them with
    //   frequency 2 lh     = u0 - u2            (lh = 0: t0 - t2;  lh = 1: t2 - t1)
    //   frequency 2 lh + 1 = fma(u1, sb, u2)    (lh = 0: t1 + t2;  lh = 1: t1 - t3 as -t3 + t1)
    // i.e. columns (0, 1, 2) with sb = +1 for lh = 0 and (2, 3, 1) with sb = -1 for lh = 1 (sums are commutative: K1w's bits)
    const int n = lane & 31, hh = lane >> 5;
    const int b2 = n >> 4, ty = (n >> 2) & 3, tx = n & 3;
    const int ra = q == 0 ? 0 : q == 2 ? 2 : 1, rb = q == 3 ? 3 : q == 2 ? 1 : 2;
    const float sg = q == 1 ? 1.0f : -1.0f, sb = lh ? -1.0f : 1.0f;
    uint32_t cur[2][3];
    {
        const int tbase = ((b2 * 8 + 2 * ty - 1) * W8_RP) + (2 * tx - 1) * SP + b2 * 16 + hh * 32;
        const int tdelta = tbase - W8_IMG;
        const int mra = (ra == 0 && ty == 0) ? 255 : -1, mrb = (rb == 3 && ty == 3) ? 255 : -1;
#pragma unroll
        for (int X = 0; X < 2; X++)
#pragma unroll
            for (int u = 0; u < 3; u++) {
                const int col = lh ? (u == 0 ? 2 : u == 1 ? 3 : 1) : u;
                const int mc = (col == 0 && tx == 0) || (col == 3 && tx == 3) ? 255 : -1;
                const int mr = X ? mrb : mra, rr = X ? rb : ra;
                cur[X][u] = (uint32_t)(W8_IMG + (tdelta & mr & mc) + rr * W8_RP + col * SP);  // chunk 0's buffer
            }
    }

    u32x4 vh[2][2], vl[2][2];  // [tb][j]: V of frequency 2 lh + j, hi and lo halves (MFMA B operands)
    f32x16 acc[2][2][2];       // [j][tb][cb]
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
        for (int t2 = 0; t2 < 2; t2++)
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[j][t2][c][e] = 0.0f;

    // V of one k-step: both tile blocks, both 4-channel groups
    auto make_v = [&](int kp) __attribute__((always_inline)) {
#pragma unroll
        for (int tbv = 0; tbv < 2; tbv++)
#pragma unroll
            for (int g = 0; g < 2; g++) {
                const int imm = tbv * W8_IMGZ + kp * 64 + g * 16;
                f32x4 pa[3], pb[3], tt[3];
#pragma unroll
                for (int u = 0; u < 3; u++) {
                    pa[u] = *reinterpret_cast<const f32x4*>(smem + cur[0][u] + imm);
                    pb[u] = *reinterpret_cast<const f32x4*>(smem + cur[1][u] + imm);
                }
#pragma unroll
                for (int u = 0; u < 3; u++)
#pragma unroll
                    for (int e = 0; e < 4; e++) asm("v_fma_f32 %0, %1, %2, %3" : "=v"(tt[u][e]) : "v"(pb[u][e]), "s"(sg), "v"(pa[u][e]));
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    f32x4 xx;
                    if (j == 0) xx = tt[0] - tt[2];
                    else {
#pragma unroll
                        for (int e = 0; e < 4; e++) asm("v_fma_f32 %0, %1, %2, %3" : "=v"(xx[e]) : "v"(tt[1][e]), "s"(sb), "v"(tt[2][e]));
                    }
                    const w8_f16x2 h0 = __builtin_convertvector(__builtin_shufflevector(xx, xx, 0, 1), w8_f16x2);
                    const w8_f16x2 h1 = __builtin_convertvector(__builtin_shufflevector(xx, xx, 2, 3), w8_f16x2);
                    const uint32_t hu0 = __builtin_bit_cast(uint32_t, h0), hu1 = __builtin_bit_cast(uint32_t, h1);
                    uint32_t lo0, lo1;  // lo = f16(x - hi) as v_fma_mix_f32 + v_cvt_pk_f16_f32: kernels_wino4.hip (the bits of v_fma_mixlo/hi_f16, a third cheaper to issue)
                    float r0, r1, r2, r3;
                    asm("v_fma_mix_f32 %2, -%6, 1.0, %8 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %3, -%6, 1.0, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                        "v_fma_mix_f32 %4, -%7, 1.0, %10 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %5, -%7, 1.0, %11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                        "v_cvt_pk_f16_f32 %0, %2, %3\n\tv_cvt_pk_f16_f32 %1, %4, %5"
                        : "=&v"(lo0), "=&v"(lo1), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
                        : "v"(hu0), "v"(hu1), "v"(xx[0]), "v"(xx[1]), "v"(xx[2]), "v"(xx[3]));
                    vh[tbv][j][2 * g] = hu0, vh[tbv][j][2 * g + 1] = hu1;
                    vl[tbv][j][2 * g] = lo0, vl[tbv][j][2 * g + 1] = lo1;
                }
            }
    };

    // An iteration = a k-step of a wave: [make V: the transform] [multiply: 24 MFMAs in four stages, the refills of the U slots behind
    // their last use].  The same program in both halves of a workgroup meets at the chunk barrier and stays in lockstep -- both waves of
    // a SIMD want the VALU and then the matrix pipe at the same time (the first build: 30.5 us per layer against K1w4's 29.2).  So the
    // halves meet the chunk's barrier at DIFFERENT places of the iteration: the lh = 0 waves in front of its transform, the lh = 1 waves
    // behind it -- from then on one multiplies while its SIMD partner transforms.  What makes that legal is a third chunk buffer: at the
    // barrier of chunk c (c = iteration / 2) chunk c + 1 is published -- every wave has seen its pieces land -- and chunk c - 1 is dead
    // for everybody, its buffer takes chunk c + 2; a wave reads chunk c, published one barrier earlier, on either side of it.
    // SP_ = parity of the iteration: in the even one the chunk's W8_P pieces go out (in front of its first counted wait in both halves).
    // Behind a U stage in the queue: stage 0 (slot 0): slot 1 of the same k-step [+ the pieces]; stage 2 (slot 1): [the pieces] + slot 0
    // of the next k-step
    auto chunk_barrier = [&]() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto multiply = [&](const char* wnext, auto sp_tag) __attribute__((always_inline)) {
        constexpr int DMA = decltype(sp_tag)::value == 0 ? W8_P : 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            // stage i: (tb, j) in the order (0,0) (1,0) (0,1) (1,1): a U stage is used in two consecutive stages
            const int tbv = i & 1, j = i >> 1;
            if (tbv == 0) {
                u32x4 r0 = ring[j][0], r1 = ring[j][1], r2 = ring[j][2], r3 = ring[j][3];
                asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "n"(4 + DMA));
                ring[j][0] = r0, ring[j][1] = r1, ring[j][2] = r2, ring[j][3] = r3;
            }
            const frag uh0 = __builtin_bit_cast(frag, ring[j][0]), ul0 = __builtin_bit_cast(frag, ring[j][1]);
            const frag uh1 = __builtin_bit_cast(frag, ring[j][2]), ul1 = __builtin_bit_cast(frag, ring[j][3]);
            const frag bh = __builtin_bit_cast(frag, vh[tbv][j]), bl = __builtin_bit_cast(frag, vl[tbv][j]);
            Mfma<T>::mac(ul0, bh, acc[j][tbv][0]);
            Mfma<T>::mac(ul1, bh, acc[j][tbv][1]);
            Mfma<T>::mac(uh0, bl, acc[j][tbv][0]);
            Mfma<T>::mac(uh1, bl, acc[j][tbv][1]);
            Mfma<T>::mac(uh0, bh, acc[j][tbv][0]);
            Mfma<T>::mac(uh1, bh, acc[j][tbv][1]);
            __builtin_amdgcn_sched_barrier(0);
            if (tbv == 1) load_ustage(ring[j], wnext, j);  // the slot is free: the next k-step's stage
        }
    };
    // chunks 0 and 1 have landed (everything but the ring's 8 loads); the first barrier publishes chunk 0 (and the zero areas)
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const size_t kstride = (size_t)16 * SW_STAGE;  // U of the next k-step
    W8_STAMP(1);
    int c = 0, bi = 0;  // chunk, its buffer index (c mod 3)
    do {
        // chunk c's barrier: publishes chunk c + 1, frees the buffer of chunk c - 1 for chunk c + 2 (clamped to the last chunk: nobody reads it)
        const uint32_t into = (uint32_t)((bi == 0 ? 2 : bi - 1) * W8_DBUF);
        const int fetch = min(c + 2, nch - 1);
        if (lh == 0) chunk_barrier();
        __builtin_amdgcn_sched_barrier(0);
        make_v(0);
        __builtin_amdgcn_sched_barrier(0);
        if (lh == 1) chunk_barrier();
        issue_chunk(fetch, into);  // behind the barrier in both halves, at ONE place of the program: the counted waits are the same
        __builtin_amdgcn_sched_barrier(0);
        multiply(wks + kstride, w8_int<0>{});  // k-step 2c; refills: k-step 2c + 1
        __builtin_amdgcn_sched_barrier(0);
        make_v(1);
        __builtin_amdgcn_sched_barrier(0);
        multiply(wks + (c + 1 < nch ? 2 : 1) * kstride, w8_int<1>{});  // k-step 2c + 1; refills: k-step 2c + 2, or the last one again
        wks += 2 * kstride;
        // on to chunk c + 1's buffer
        const int step = bi == 2 ? -2 * W8_DBUF : W8_DBUF;
        bi = bi == 2 ? 0 : bi + 1;
#pragma unroll
        for (int X = 0; X < 2; X++)
#pragma unroll
            for (int u = 0; u < 3; u++) cur[X][u] += step;
    } while (++c < nch);
    W8_STAMP(2);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 2; j++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ring[j][0]), "+v"(ring[j][1]), "+v"(ring[j][2]), "+v"(ring[j][3])::"memory");
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue, one tile block at a time ----
    int elane = lane;  // (an opaque copy: kernels_wino.hip)
    asm volatile("" : "+v"(elane));
    const int en = elane & 31, eh = elane >> 5;
    const int erow = elane >> 4, pc = elane & 15;  // final layout: lane = (row erow of the instruction's four, couts 4 pc .. 4 pc + 3)
    const f32x4 bias4 = *reinterpret_cast<const f32x4*>(bias + cout0 + pc * 4);
    const f32x4 ds4 = *reinterpret_cast<const f32x4*>(bias + cout + cout0 + pc * 4);
    float vmax = 0.0f;
    char* regM = smem;             // region A
    char* regZ = smem + W8_LDS_M;  // region B
    asm volatile("s_barrier" ::: "memory");  // every wave has left the chunk buffers
#pragma unroll
    for (int tbv = 0; tbv < 2; tbv++) {
        // the final phase's work of this wave: instruction k (0..3) = pixel pair pp = w * 4 + k of the tile block's 32 (tile row pp >> 3 ... ):
        // rows = (board 0, pixel), (board 0, pixel ^ 1), (board 1, pixel), (board 1, pixel ^ 1) -- the four share their tile's index
        f32x4 skip[4];
        if (HAS_RES) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int pp = w * 4 + k, px = ((pp >> 2) << 3) + ((pp & 3) << 1) + (erow & 1);  // pixel y = pp >> 2, x = 2 (pp & 3) + (erow & 1)
                const size_t row = (size_t)row0 + tbv * 128 + (erow >> 1) * 64 + px;
                skip[k] = *reinterpret_cast<const f32x4*>(res + row * (size_t)cout + cout0 + pc * 4);
            }
        }
        // (1) the lh = 1 waves hand their M to their partners: [q][x4 index][lane]
        if (lh == 1) {
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int cb = 0; cb < 2; cb++)
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        f32x4 m;
#pragma unroll
                        for (int e = 0; e < 4; e++) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(m[e]) : "a"(acc[j][tbv][cb][g * 4 + e]));
                        *reinterpret_cast<f32x4*>(regM + ((size_t)(q * 16 + (j * 2 + cb) * 4 + g) * 64 + elane) * 16) = m;
                    }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // (2) the lh = 0 waves: Z[.][0] = M0 + M1 + M2, Z[.][1] = M1 - M2 - M3 (K1w's order) into K1w4's exchange layout
        // [q][c'][tile 0..31][16 units of 4 couts], the unit index XORed with the tile's index inside its board
        if (lh == 0) {
#pragma unroll
            for (int cb = 0; cb < 2; cb++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    f32x4 m0, m1;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(m0[e]) : "a"(acc[0][tbv][cb][g * 4 + e]));
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(m1[e]) : "a"(acc[1][tbv][cb][g * 4 + e]));
                    }
                    const f32x4 m2 = *reinterpret_cast<const f32x4*>(regM + ((size_t)(q * 16 + (0 * 2 + cb) * 4 + g) * 64 + elane) * 16);
                    const f32x4 m3 = *reinterpret_cast<const f32x4*>(regM + ((size_t)(q * 16 + (1 * 2 + cb) * 4 + g) * 64 + elane) * 16);
                    const f32x4 z0 = m0 + m1 + m2, z1 = m1 - m2 - m3;
                    const int unit = (cb * 8 + g * 2 + eh) ^ (en & 15);
                    char* zp = regZ + ((size_t)(q * 2) * 32 + en) * 256 + (unit << 4);
                    *reinterpret_cast<f32x4*>(zp) = z0;
                    *reinterpret_cast<f32x4*>(zp + 32 * 256) = z1;
                }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // (3) all eight waves: Y[0] = Z0 + Z1 + Z2, Y[1] = Z1 - Z2 - Z3 over the frequency rows, bias, skip, ReLU, cap, store
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int pp = w * 4 + k, y = pp >> 2, x = ((pp & 3) << 1) + (erow & 1);
            const int rp = y & 1, cp = x & 1, tile16 = (y >> 1) * 4 + (x >> 1), tile = (erow >> 1) * 16 + tile16;
            const char* zb = regZ + ((size_t)cp * 32 + tile) * 256 + ((pc ^ tile16) << 4);
            const f32x4 za = *reinterpret_cast<const f32x4*>(zb + (size_t)(rp + 0) * 2 * 32 * 256);
            const f32x4 zc = *reinterpret_cast<const f32x4*>(zb + (size_t)(rp + 1) * 2 * 32 * 256);
            const f32x4 zd = *reinterpret_cast<const f32x4*>(zb + (size_t)(rp + 2) * 2 * 32 * 256);
            const f32x4 a = rp == 0 ? za + zc + zd : za - zc - zd;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float xv = __builtin_fmaf(a[e], ds4[e], bias4[e]);
                if (HAS_RES) xv = xv + skip[k][e];
                xv = xv > 0.0f ? xv : 0.0f;
                v[e] = xv < WINO_ACT_MAX ? xv : WINO_ACT_MAX;
            }
            vmax = fmaxf(fmaxf(vmax, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
            const size_t row = (size_t)row0 + tbv * 128 + (erow >> 1) * 64 + y * 8 + x;
            *reinterpret_cast<f32x4*>(out + row * (size_t)cout + cout0 + pc * 4) = v;
        }
        if (tbv == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // region B is read: the next tile block's Z may come
    }
    if (vmax >= WINO_ACT_MAX) atomicAdd(sat, 1u);
#ifdef CATTUS_STAMPS
    W8_STAMP(3);
    if (lane == 0 && blockIdx.x < 1024) {
#pragma unroll
        for (int i = 0; i < 4; i++) g_stamps_w8[((size_t)blockIdx.x * 8 + w) * 4 + i] = st_[i];
    }
#endif
}

#ifdef CATTUS_STAMPS
extern "C" __attribute__((visibility("default"))) int cattus_hip_debug_stamps_w8(unsigned long long* out, size_t n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps_w8), n * sizeof(unsigned long long));
}
#endif

bool wino8_supported(uint32_t bpad, uint32_t cin, uint32_t cout, uint32_t S) {
    return S == 8 && cin >= 128 && cin % 32 == 0 && cout >= 128 && cout % 64 == 0 && bpad % 4 == 0;
}

void launch_conv3x3_wino8(const float* in, const void* wu, const float* bias, const float* res, float* out, uint32_t bpad, uint32_t cin,
                          uint32_t cout, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop, unsigned* sat) {
    typedef _Float16 H;
    const dim3 grid((bpad / 4) * (cout / 64));
    if (res)
        hipExtLaunchKernelGGL((conv3x3_wino8_kernel<true>), grid, dim3(512), W8_LDS_TOTAL, st, ev_start, ev_stop, 0, in, (const H*)wu, bias, res, out,
                              sat, (int)cin, (int)cout);
    else
        hipExtLaunchKernelGGL((conv3x3_wino8_kernel<false>), grid, dim3(512), W8_LDS_TOTAL, st, ev_start, ev_stop, 0, in, (const H*)wu, bias, res, out,
                              sat, (int)cin, (int)cout);
}

hipError_t prepare_wino8() {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino8_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, W8_LDS_TOTAL);
    const hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino8_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, W8_LDS_TOTAL);
    return err != hipSuccess ? err : e2;
}

}  // namespace cattus
