// gfx950 kernels of the Cattus leaf evaluator, part 2: the resident split-precision tower (K1rs).  See kernels.hip for the
// per-layer kernels it is built from and bit-identical to.
#include "kernels.h"
#include "device_common.h"

#include <hip/hip_ext.h>

namespace cattus {

// ------------------------------------------------------------------------------------------
// K1rs resident: the whole tower of a <= 64-filter network in ONE launch, split precision (dtype f16x2)
// ------------------------------------------------------------------------------------------
//
// The f16x2 counterpart of tower64_lds_kernel, built from conv3x3_splitw_kernel's parts.  With 64 (padded) filters a
// workgroup that owns a board owns all of its output channels, so a layer's output is the next layer's input in place:
// the activations stay in LDS as f16 pairs (two buffers of [2 chunks of 32 channels][rows][144 B: hi | lo | pad], the
// split conv's image), nothing but the weights is read in the loop and nothing is written before the last layer.
// The WEIGHTS need no LDS at all: they are in MFMA fragment order (kernels.h::split_frag_index) and go from L2 into each
// wave's register ring, SW_D stages ahead ACROSS layer boundaries (the ring never drains between layers), so the kernel
// has no loader waves: a workgroup is four consumer waves, wave w = (pixel group w >> 1, output-channel block w & 1),
// NPB 32-pixel blocks each.  NPB = 1: 64 rows = one 64-slot board per workgroup; NPB = 2: 128 rows = one 128-slot board.
// One s_barrier per layer orders a wave's output writes against its neighbours' reads of the next layer.
// Per accumulator the MFMA sequence (chunk -> tap -> k-half, a_lo w_hi, a_hi w_lo, a_hi w_hi) and the epilogue
// (fma(acc, 2^-s, bias), + skip as hi + lo, ReLU, clamp, split) are those of the per-layer kernel, operation for
// operation: results are bit-identical to it (tests/test_hip_parity.py::test_resident_tower_equals_per_layer_launches).
// After the last layer the two 1x1 head convs run on the resident f32 output (head_conv_tile<float>'s chain: the bits of the
// stand-alone launch) and write hv for the FC launch; optionally the f32 rows themselves go to HBM.
constexpr int T64S_HP = 272;  // pitch of the f32 rows staged for the head convs, and of the head weight rows: 256 B + 16 (conflict-free b128 reads down rows)
// two activation buffers | per layer [64 biases | 64 inverse scales] | head conv weights [32][64] f32 at pitch 272 | 32 head biases
// behind a chunk image's rows: a zero area (device_common.h: with one zero row 37 % of this kernel's LDS cycles were bank conflicts on hex 7x7)
constexpr int T64S_ZAREA = ZAREA_SP;
constexpr int t64s_lds_bytes(int npb, int nlayers) { return 2 * 2 * (64 * npb * SP + T64S_ZAREA) + nlayers * 512 + 32 * T64S_HP + 128; }

// D: weight stages in flight per wave (a stage is only 3 NPB MFMAs here, so the ring is deeper than the per-layer kernel's to
// cover an L2 round trip); PA: stages of look-ahead on the pixel fragments.
// BIG: 128 pixel slots per board (NPB = 2 then: the workgroup is one board); otherwise NPB = 2 is TWO 64-slot boards per
// workgroup, a wave holding a whole board's 64 pixels x 32 couts: two independent accumulator chains per wave and half the
// weight bytes per MFMA (the four one-tile waves of NPB = 1 pull 8 KiB of weights per 96-cycle stage through the CU's 64 B/clk
// vector-memory return path and run at two thirds of the MFMA rate for it).
template <int NPB, int D, int PA, bool BIG = false>
__global__ void __launch_bounds__(256, 1) tower64_split_kernel(Tower64SplitArgs A) {
    static_assert(!BIG || NPB == 2, "a 128-slot board is one 128-row workgroup");
    typedef _Float16 T;
    typedef Mfma<T>::frag frag;
    static_assert(18 % D == 0 && D <= 18, "the ring turns a whole number of times per 18-stage chunk");
    static_assert(PA == 1 || PA == 2, "one or two stages of pixel look-ahead");
    constexpr int PR = PA + 1;  // pixel fragment ring; 18 % PR == 0 for both
    constexpr int ROWS = 64 * NPB;           // tower rows of this workgroup
    constexpr int ZERO = ROWS * SP;          // a chunk image's zero area, behind its rows (a multiple of 256)
    constexpr int CHUNK = ZERO + T64S_ZAREA; // bytes of one 32-channel chunk image
    static_assert(ZERO % 256 == 0, "the zero area keeps a read's banks only if it is 256-B aligned inside the image");
    constexpr int BUF = 2 * CHUNK;           // an activation buffer: channels 0..31, 32..63
    constexpr int TABLE = 2 * BUF;           // per layer [64 biases | 64 inverse scales] behind the two buffers
    constexpr int SLOTS = BIG ? 128 : 64;    // pixel slots per board
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int S = (int)A.S, hw = S * S;
    const int row0 = blockIdx.x * ROWS;
    const int nlayers = (int)A.nlayers;
    const int cb = wave & 1, pg = wave >> 1;  // output-channel block, pixel group of this wave

    // ---- the weight ring starts first: its latency hides under the plane expansion ----
    const uint32_t voff0 = lane * 16;
    u32x4 ring[D][2];
    auto load_stage = [&](u32x4(&slot)[2], const char* p) {
        u32x4 l0, l1;
        asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024"
                     : "=&v"(l0), "=&v"(l1)
                     : "v"(voff0), "s"(p)
                     : "memory");
        slot[0] = l0, slot[1] = l1;
    };
    // all but the 2 (D - 1) youngest ring loads have returned.  The count is the same at every stage of the kernel: every stage
    // refills its slot (the last layer's last D stages re-read weights nobody uses), so that no branch ever separates a ring
    // load from its wait -- where control flow merges the compiler may copy a register, and a copy of a register whose load is
    // still in flight copies garbage (scripts/audit_inflight_regs.py checks the generated code for exactly that).
    auto wait_stage = [&](u32x4(&slot)[2]) {
        u32x4 r0 = slot[0], r1 = slot[1];
        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r0), "+v"(r1) : "n"(2 * (D - 1)));
        slot[0] = r0, slot[1] = r1;
    };
    auto layer_w = [&](int l) {  // this wave's cout block of layer l's weights, as a scalar (the ring's loads take an SGPR base)
        const Tower64SplitLayer L = A.layers[l];
        const uint64_t v = (uint64_t)(reinterpret_cast<const char*>(L.wf) + (size_t)cb * (L.nch * 18) * SW_STAGE);
        uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        // an SGPR written by v_readfirstlane needs 5 wait states before a vector-memory instruction may read it as its base; the
        // compiler pads what it schedules itself, not the inside of the ring's asm statements
        asm volatile("s_nop 4" : "+s"(lo), "+s"(hi));
        return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
    };

    const char* wcur = layer_w(0);
#pragma unroll
    for (int d = 0; d < D; d++) load_stage(ring[d], wcur + (size_t)d * SW_STAGE);  // the stem has 18 stages >= D

    // ---- bias / inverse-scale table of every layer and the head convs' operands -> LDS.  Ordinary loads, issued behind the
    // ring's first loads: where the compiler waits for one of them it waits for everything older too (loads return in order),
    // and they have all returned before this wave's first counted wait (their values are written to LDS ahead of the barrier) ----
    for (int i = tid; i < nlayers * 128; i += 256) *reinterpret_cast<float*>(smem + TABLE + i * 4) = A.bias_all[i];  // one table: no pointer chase
    const int HEADW = TABLE + nlayers * 512, HEADB = HEADW + 32 * T64S_HP;  // the fused head convs' operands
    if (A.head_w) {
        for (int i = tid; i < 32 * 64; i += 256) *reinterpret_cast<float*>(smem + HEADW + (i >> 6) * T64S_HP + (i & 63) * 4) = A.head_w[i];
        if (tid < 32) *reinterpret_cast<float*>(smem + HEADB + tid * 4) = A.head_b[tid];
    }
    // zero areas of the four chunk images
    if (tid < 4 * (T64S_ZAREA / 16))
        reinterpret_cast<f32x4*>(smem + (tid / (T64S_ZAREA / 16)) * CHUNK + ZERO)[tid % (T64S_ZAREA / 16)] = f32x4{0.f, 0.f, 0.f, 0.f};
    // ---- stem input: bitboard planes -> buffer 1, chunk 0: hi = 1.0 where the plane has the pixel's bit, lo = 0 ----
    {
        constexpr int SPT = ROWS * 4 / 256;  // 16-byte hi slots per thread (a row has 4: 32 channels)
        typedef const __attribute__((address_space(1))) uint64_t* gu64p;
#pragma unroll
        for (int q = 0; q < SPT; q++) {
            const int v = tid + q * 256, row = v >> 2, sl = v & 3;
            const uint32_t grow = (uint32_t)(row0 + row);
            const uint32_t board = grow / SLOTS, px = grow % SLOTS;
            const bool live = board < A.n && (int)px < hw;
            const gu64p pl = (gu64p)(A.planes + (size_t)(live ? board : 0) * A.C * A.w64 + (live ? (px >> 6) : 0));
            T vals[8];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const uint32_t c = sl * 8 + i;
                const uint64_t word = (live && c < A.C) ? pl[(size_t)c * A.w64] : 0ull;
                vals[i] = ((word >> (px & 63)) & 1ull) ? (T)1.0f : (T)0.0f;
            }
            char* dst = smem + BUF + row * SP + sl * 16;
            *reinterpret_cast<f32x4*>(dst) = *reinterpret_cast<f32x4*>(vals);
            *reinterpret_cast<f32x4*>(dst + 64) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // every table load has returned (and with them the ring's first D stages)

    // ---- per-lane constants ----
    const int r = lane & 31, h = lane >> 5;
    int rowa[9][NPB];  // per tap and pixel block: byte offset of the shifted pixel's row in a chunk image (+ the lane half's 16 B)
    bool pvalid[NPB];
#pragma unroll
    for (int pb = 0; pb < NPB; pb++) {
        const int lrow = (pg * NPB + pb) * 32 + r;     // row of this lane's pixel inside the workgroup
        const int board_row = lrow / SLOTS * SLOTS, p = lrow % SLOTS;
        const int ph_ = p / S, pw = p - ph_ * S;
        pvalid[pb] = p < hw;
#pragma unroll
        for (int t9 = 0; t9 < 9; t9++) {
            const int hh = ph_ + t9 / 3 - 1, ww = pw + t9 % 3 - 1;
            const bool ok = pvalid[pb] && (unsigned)hh < (unsigned)S && (unsigned)ww < (unsigned)S;
            const int at = (board_row + hh * S + ww) * SP + h * 16;  // may lie outside the image when !ok: only its low 8 bits are used then
            rowa[t9][pb] = ok ? at : ZERO + (at & 255);
        }
    }
    // epilogue addresses: this lane's accumulator element g * 4 + i is cout cb * 32 + g * 8 + h * 4 + i of pixel r: four
    // consecutive channels = 8 bytes of hi values at (g * 8 + h * 4) * 2 in the pixel's row of chunk cb, the lo values 64 further
    int erow[NPB];
#pragma unroll
    for (int pb = 0; pb < NPB; pb++) erow[pb] = cb * CHUNK + ((pg * NPB + pb) * 32 + r) * SP + h * 8;

    asm volatile("s_barrier" ::: "memory");  // planes, zero rows and the table are in LDS

    int opaque = 0;
    // One layer.  NCH (32-channel chunks of its input) is a compile-time constant -- 1 for the stem, 2 for every other layer --
    // so that the stage sequence is straight-line code.
    auto run_layer = [&](int layer, auto nch_tag) {
        constexpr int NCH = decltype(nch_tag)::value;
        const Tower64SplitLayer L = A.layers[layer];
        const bool last_layer = layer + 1 == nlayers;
        const char* wnext = last_layer ? wcur : layer_w(layer + 1);  // the last layer's look-ahead re-reads its own first stages
        // the stem reads buffer 1 and writes 0; a block's first conv reads 0 and writes 1, its second reads 1, adds 0 (the
        // block input) and writes 0
        const int ibase = ((layer & 1) ? 0 : 1) * BUF, obase = BUF - ibase;
        f32x16 acc[NPB];
#pragma unroll
        for (int j = 0; j < NPB; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[j][e] = 0.0f;
        frag ph[PR][NPB], pl[PR][NPB];
#pragma unroll
        for (int q = 0; q < PA; q++)  // stages 0 .. PA-1 of the layer: tap 0, k-half q (PA <= 2)
#pragma unroll
            for (int pb = 0; pb < NPB; pb++) {
                ph[q][pb] = *reinterpret_cast<const frag*>(smem + ibase + rowa[0][pb] + q * 32);
                pl[q][pb] = *reinterpret_cast<const frag*>(smem + ibase + rowa[0][pb] + q * 32 + 64);
            }
        // one chunk = 18 stages; LASTC: the layer's last chunk (its look-ahead loads belong to the next layer, or count the ring
        // down on the last layer)
        auto chunk = [&](int ch, auto lastc_tag) {
            constexpr bool LASTC = decltype(lastc_tag)::value;
            asm volatile("" : "+v"(opaque));  // keeps `rowa + base` from being hoisted out of the layer loop
            const int cbase = ibase + ch * CHUNK + opaque;
            const int s0 = ch * 18;
#pragma unroll
            for (int j = 0; j < 18; j++) {
                const int cur = j % PR, nxt = (j + PA) % PR;
                if (j + PA < 18 || !LASTC) {  // PA stages of look-ahead on the pixel fragments, across the chunk boundary
                    const int jn = (j + PA) % 18, t = jn >> 1, k = jn & 1;
#pragma unroll
                    for (int pb = 0; pb < NPB; pb++) {
                        const int a = rowa[t][pb] + (j + PA < 18 ? cbase : cbase + CHUNK) + k * 32;
                        ph[nxt][pb] = *reinterpret_cast<const frag*>(smem + a);
                        pl[nxt][pb] = *reinterpret_cast<const frag*>(smem + a + 64);
                    }
                }
                wait_stage(ring[j % D]);
                const frag wh = __builtin_bit_cast(frag, ring[j % D][0]);
                const frag wl = __builtin_bit_cast(frag, ring[j % D][1]);
                // term by term across the pixel blocks: per accumulator the sequence is the per-layer kernel's (a_lo w_hi, a_hi w_lo,
                // a_hi w_hi), and with two blocks a wave's consecutive MFMAs belong to different accumulator chains
#pragma unroll
                for (int pb = 0; pb < NPB; pb++) Mfma<T>::mac(wl, ph[cur][pb], acc[pb]);
#pragma unroll
                for (int pb = 0; pb < NPB; pb++) Mfma<T>::mac(wh, pl[cur][pb], acc[pb]);
#pragma unroll
                for (int pb = 0; pb < NPB; pb++) Mfma<T>::mac(wh, ph[cur][pb], acc[pb]);
                // D stages ahead: this layer's, or -- from its last D stages -- the next layer's first
                if (LASTC && j >= 18 - D) load_stage(ring[j % D], wnext + (size_t)(j - (18 - D)) * SW_STAGE);
                else load_stage(ring[j % D], wcur + (size_t)(s0 + j + D) * SW_STAGE);
            }
        };
        if constexpr (NCH == 2) chunk(0, std::false_type{});
        chunk(NCH - 1, std::true_type{});

        // ---- layer epilogue ----
        const char* tab = smem + TABLE + layer * 512 + (cb * 32 + h * 4) * 4;
        if (!last_layer) {
#pragma unroll
            for (int pb = 0; pb < NPB; pb++) {
                char* orow = smem + obase + erow[pb];
#pragma unroll
                for (int half = 0; half < 2; half++) {  // 8 couts at a time (the unit of the saturation count and of the skip reads)
                    float v[8];
                    typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
                    f16x4 sh[2], sl[2];
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const int g = half * 2 + q;
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(tab + g * 32);
                        const f32x4 dv = *reinterpret_cast<const f32x4*>(tab + 256 + g * 32);
                        if (L.res) {
                            sh[q] = *reinterpret_cast<const f16x4*>(orow + g * 16);
                            sl[q] = *reinterpret_cast<const f16x4*>(orow + g * 16 + 64);
                        }
#pragma unroll
                        for (int i = 0; i < 4; i++) v[q * 4 + i] = __builtin_fmaf(acc[pb][g * 4 + i], dv[i], bv[i]);
                    }
                    if (L.res) {
#pragma unroll
                        for (int j = 0; j < 8; j++) v[j] = v[j] + ((float)sh[j >> 2][j & 3] + (float)sl[j >> 2][j & 3]);  // hi + lo is exact in f32
                    }
                    float y[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) y[j] = v[j] > 0.0f ? v[j] : 0.0f;
                    note_saturation(y, pvalid[pb], A.sat);
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        const int g = half * 2 + q;
                        f16x4 hi, lo;
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const float yc = y[q * 4 + i] < 65504.0f ? y[q * 4 + i] : 65504.0f;
                            hi[i] = (T)yc;
                            lo[i] = (T)(yc - (float)hi[i]);
                        }
                        *reinterpret_cast<f16x4*>(orow + g * 16) = hi;
                        *reinterpret_cast<f16x4*>(orow + g * 16 + 64) = lo;
                    }
                }
            }
            // every wave's output rows are written (and its reads of this layer's input done) before anybody starts the next layer
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
            // the tower's last layer: f32 values (pixel slots past the board zero), as the per-layer tower's last launch writes them
            f32x4 yv[NPB][4];
#pragma unroll
            for (int pb = 0; pb < NPB; pb++) {
                const char* srow = smem + obase + erow[pb];
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
                    const f32x4 bv = *reinterpret_cast<const f32x4*>(tab + g * 32);
                    const f32x4 dv = *reinterpret_cast<const f32x4*>(tab + 256 + g * 32);
                    f32x4 y;
#pragma unroll
                    for (int i = 0; i < 4; i++) y[i] = __builtin_fmaf(acc[pb][g * 4 + i], dv[i], bv[i]);
                    if (L.res) {
                        const f16x4 sh = *reinterpret_cast<const f16x4*>(srow + g * 16);
                        const f16x4 sl = *reinterpret_cast<const f16x4*>(srow + g * 16 + 64);
#pragma unroll
                        for (int i = 0; i < 4; i++) y[i] = y[i] + ((float)sh[i] + (float)sl[i]);
                    }
#pragma unroll
                    for (int i = 0; i < 4; i++) y[i] = pvalid[pb] && y[i] > 0.0f ? y[i] : 0.0f;
                    yv[pb][g] = y;
                }
            }
            if (A.out) {  // the rows themselves -> HBM [row][64] f32 (for the stand-alone head conv launch)
#pragma unroll
                for (int pb = 0; pb < NPB; pb++) {
                    float* of = A.out + ((size_t)row0 + (pg * NPB + pb) * 32 + r) * 64 + cb * 32 + h * 4;
#pragma unroll
                    for (int g = 0; g < 4; g++) *reinterpret_cast<f32x4*>(of + g * 8) = yv[pb][g];
                }
            }
            if (A.head_w) {
                // K3 fused: the two 1x1 head convs (+ folded BN + ReLU) in exact f32 on the resident output -- the rows are staged
                // in LDS (every wave is done with the activation buffers), then the wave of each pixel group that holds cout block
                // 0 runs head_conv_tile<float>'s MFMA chain (k in 8-groups, v_mfma_f32_32x32x2_f32 x 4 per group) and writes hv
                // in the FC launch's fragment order: the bits of the stand-alone head conv launch.
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
                for (int pb = 0; pb < NPB; pb++)
#pragma unroll
                    for (int g = 0; g < 4; g++)
                        *reinterpret_cast<f32x4*>(smem + ((pg * NPB + pb) * 32 + r) * T64S_HP + (cb * 32 + g * 8 + h * 4) * 4) = yv[pb][g];
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (cb == 0) {
                    f32x4 hb[4];
#pragma unroll
                    for (int g = 0; g < 4; g++) hb[g] = *reinterpret_cast<const f32x4*>(smem + HEADB + (8 * g + 4 * h) * 4);
#pragma unroll
                    for (int pb = 0; pb < NPB; pb++) {
                        const int lrow = (pg * NPB + pb) * 32 + r;
                        f32x4 fa[8], fb[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            fa[u] = *reinterpret_cast<const f32x4*>(smem + HEADW + r * T64S_HP + (u * 8 + h * 4) * 4);
                            fb[u] = *reinterpret_cast<const f32x4*>(smem + lrow * T64S_HP + (u * 8 + h * 4) * 4);
                        }
                        f32x16 hacc;
#pragma unroll
                        for (int e = 0; e < 16; e++) hacc[e] = 0.0f;
#pragma unroll
                        for (int u = 0; u < 8; u++) Mfma<float>::mac(fa[u], fb[u], hacc);
                        const uint32_t grow = (uint32_t)(row0 + lrow);
                        const uint32_t bb = grow / SLOTS, p = grow % SLOTS;
                        if ((int)p < hw) {
#pragma unroll
                            for (int e = 0; e < 16; e++) {
                                const uint32_t i = (e & 3) + 8 * (e >> 2) + 4 * h;
                                if (i >= A.ocn) continue;
                                const float y = hacc[e] + hb[e >> 2][e & 3];
                                const size_t at = i < A.vhc ? frag_packed_index<float>(bb, i * hw + p, A.kvp)
                                                            : A.hv_pol + frag_packed_index<float>(bb, (i - A.vhc) * hw + p, A.kpp);
                                A.hv[at] = y > 0.0f ? y : 0.0f;
                            }
                        }
                    }
                }
            }
        }
        wcur = wnext;
    };
    run_layer(0, std::integral_constant<int, 1>{});
    for (int layer = 1; layer < nlayers; layer++) run_layer(layer, std::integral_constant<int, 2>{});
    // the ring's last refills (nobody reads them).  The ring's registers are operands of the wait: to the compiler they are free from
    // their last MFMA on, and it would park other values in them while the loads are still on their way
#pragma unroll
    for (int d = 0; d < D; d++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ring[d][0]), "+v"(ring[d][1])::"memory");
}

void launch_tower64_split(const Tower64SplitArgs& args, uint32_t rows, int shape, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
    const bool big = tower_slots(args.S) == 128;
#define CATTUS_LAUNCH_T64S(NPB, D, PA, BIGV)                                                                                      \
    hipExtLaunchKernelGGL((tower64_split_kernel<NPB, D, PA, BIGV>), dim3(rows / (64 * NPB)), dim3(256),                           \
                          t64s_lds_bytes(NPB, (int)args.nlayers), st, ev_start, ev_stop, 0, args)
    if (big) {
        CATTUS_LAUNCH_T64S(2, 6, 1, true);
        return;
    }
    // 64-slot boards: one board per workgroup (measured, hex7 6x64: 47 / 53 / 93 us at 128 / 256 / 512 boards against 72 / 77 / 86
    // with two boards per workgroup, which only pays once every CU holds two of the small workgroups)
    if (shape == 0) shape = rows / 64 >= 512 && rows % 128 == 0 ? 2 : 1;
    if (rows % 128 != 0) shape = 1;
    if (shape == 2) CATTUS_LAUNCH_T64S(2, 6, 1, false);
    else if (shape == 9) CATTUS_LAUNCH_T64S(1, 9, 2, false);
    else CATTUS_LAUNCH_T64S(1, 6, 1, false);
#undef CATTUS_LAUNCH_T64S
}

hipError_t prepare_tower64_split() {
    hipError_t err = hipSuccess;
    auto set = [&](const void* fn, int npb) {
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, t64s_lds_bytes(npb, T64S_MAX_LAYERS));
        if (e != hipSuccess && err == hipSuccess) err = e;
    };
    set(reinterpret_cast<const void*>(&tower64_split_kernel<2, 6, 1, true>), 2);
    set(reinterpret_cast<const void*>(&tower64_split_kernel<2, 6, 1, false>), 2);
    set(reinterpret_cast<const void*>(&tower64_split_kernel<1, 6, 1, false>), 1);
    set(reinterpret_cast<const void*>(&tower64_split_kernel<1, 9, 2, false>), 1);
    return err;
}

}  // namespace cattus
