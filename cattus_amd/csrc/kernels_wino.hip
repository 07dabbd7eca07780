// gfx950 kernels of the Cattus leaf evaluator, part 3: the split-precision 3x3 conv in Winograd F(2x2, 3x3) form (K1w).
//
// conv3x3_splitw_kernel issues 36 products per 2x2 output tile and input channel (4 outputs x 9 taps), each as three f16 MFMA
// terms.  F(2x2, 3x3) needs 16: Y = A^T [ (G g G^T) . (B^T d B) ] A, with the 16 "frequencies" of a tile's 4x4 input patch d as 16
// independent GEMMs over the input channels -- 2.25x fewer MFMAs for the same layer.  What it costs was measured before this was
// written (scripts/winograd_gate_a.py -> profiles/r04_winograd_gate_a.json: error against the float64 run 1.06-1.66x the direct
// form's, inside the reference's cross-runtime tolerance; scripts/probes/winograd_probe.hip -> profiles/r04_winograd_probe.txt:
// the loop below without its transforms runs 21-23 us per 256 -> 256 layer at batch 256 where the direct loop takes 36 on the
// same harness).  The price is operand traffic: a wave has to hold all 16 accumulators of its (32 tiles x 32 couts) block until the
// output transform -- 256 registers -- so there is no room for a second block and every operand fragment feeds one accumulator:
// 1.33 KiB of fragments per MFMA (the direct kernel: 0.67).  The loop is bound by the U stream -- 2 MB per CU and layer through the
// 64 B/clk vector-memory return path, 19.7 us under the MFMAs in the probe where the MFMAs alone take 12.2 -- and still 1.6x shorter
// than the direct loop.
//
// Shape: 8x8 boards (64 pixel slots, 16 tiles of 2x2), cin and cout multiples of 64 / 128.  A tower runs in this form as a whole
// (every layer but the stem): between its layers the activations are PLAIN F32 rows [row][channel] -- what the split tower's last
// layer writes for the heads anyway (CONV_OUT_F32), so the stem is the direct kernel with that flag and the heads do not change.
// The input transform then reads f32 (no hi + lo reassembly: 5 of its 21 instructions per value) and the epilogue writes f32 (no
// split); only the transformed operands U and V are (hi, lo) pairs.  The f16 range: a transformed input is a signed sum of four
// activations, so activations are capped at WINO_ACT_MAX = 65504 / 4 where they are written (this kernel's epilogue, the stem's
// CONV_WINO_IN) and counted there; the transform itself needs no clamp.
//   workgroup = 4 waves = 2 boards (32 tiles) x 128 couts; wave w holds the 16 accumulators of (32 tiles, couts 32w .. 32w+31)
//   U (weights): G g G^T in float64 on the host, scaled per cout by a power of two, split (hi, lo), in MFMA fragment order
//       [cout / 32][k-step x 16 + f][hi | lo][lane][8 f16]; from L2 straight into a register ring, WN_D stages ahead
//   d (activations): 32-channel chunks of the 128 pixel rows by LDS-DMA (asm: the compiler neither sees nor counts them), two buffers
//   V = B^T d B: every thread transforms one (tile, channel pair) per k-step in f32, splits the 16 values into (hi, lo) and writes
//       them to the k-step's V image [f][tile][16 ch hi | 16 ch lo]; sliced over the 16 MFMA stages of the previous k-step, each slice
//       in two parts BETWEEN the stage's three MFMAs; a half wave = 8 tiles x 4 channel pairs (no LDS bank conflicts); two images
//   a stage = (k-step of 16 channels, frequency f): 2 ring fragments + 2 ds_read_b128 -> 3 MFMAs into accumulator f
//   epilogue: Y = A^T M A in registers (per lane: tile r, 16 couts), * 2^-s + bias, + skip, ReLU, cap, f32 stores.
#include "kernels.h"
#include "device_common.h"

#include <hip/hip_ext.h>

namespace cattus {

constexpr int WN_D = WINO_RING_STAGES;     // U stages in flight per wave (8 stages x 2 fragments x 4 VGPRs = 64 registers)
constexpr int WN_VP = 80;                  // V image row: 16 ch hi (32 B) | 16 ch lo (32 B) | 16 B pad (b128 reads down 16 rows conflict-free)
constexpr int WN_VF = 32 * WN_VP;          // one frequency: 32 tiles
constexpr int WN_VIMG = 16 * WN_VF;        // one k-step's V: 40,960 B
constexpr int WN_DROWS = 128;              // pixel rows of a workgroup: 2 boards x 64 slots
constexpr int WN_RP = 8 * SP + 64;         // pitch of a board row (8 pixel rows of 144 B) in the chunk image: 64 B of padding put the two tile
                                           // rows of a read group on opposite halves of the 256-B bank row (see the transform's item)
constexpr int WN_DZERO = 16 * WN_RP;       // behind the 16 board rows: a zero AREA -- a patch pixel off the board is read at zero area +
constexpr int WN_ZAREA = 4608;             // (its own address mod 256: the same banks) + the pixel's offset inside the patch (<= 3 WN_RP + 3 SP + 64)
constexpr int WN_DBUF = WN_DZERO + WN_ZAREA;  // 24,064 B
static_assert(WN_DZERO % 256 == 0 && WN_DBUF % 256 == 0 && (2 * WN_VIMG) % 256 == 0, "the zero area keeps a read's banks only if it is 256-B aligned");
constexpr int WN_LDS_D = 2 * WN_VIMG;
constexpr int WN_LDS_TOTAL = WN_LDS_D + 2 * WN_DBUF;  // 130,048 B
constexpr int WN_PA = 2;                   // stages of look-ahead on the V fragments (1, 2 and 3 time the same since the conflict-free transform)
constexpr int WN_P = 5;                    // LDS-DMA pieces per wave and chunk: 4 x 5 = 20 >= the image's 19 KiB pieces

// LDS-DMA of 64 x 16 bytes, hidden from the compiler (it would otherwise order this wave's later LDS reads behind a vmcnt(0) of
// its own, which also waits for the whole register ring): M0 = the wave-uniform LDS byte address, each lane its own source.
__device__ __forceinline__ void glds16_asm(const char* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2v;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4v;

template <bool HAS_RES>
__global__ void __launch_bounds__(256, 1)
    conv3x3_wino_kernel(const float* __restrict__ in, const _Float16* __restrict__ wu, const float* __restrict__ bias,
                        const float* res, float* out, unsigned* __restrict__ sat, int cin, int cout) {
    // res and out may be the same rows (a residual block's output over its skip rows): an element's skip value is read and its
    // result written by one lane, the read first
    typedef _Float16 T;
    typedef Mfma<T>::frag frag;
    constexpr int D = WN_D;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;

    const int nblk = gridDim.x, ncg = cout >> 7;
    int logical = blockIdx.x;
    if ((nblk & 7) == 0) logical = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);  // one XCD: all cout groups of a range of boards
    const int cout0 = (logical % ncg) * 128 + wave * 32;  // this wave's 32 output channels
    const int row0 = (logical / ncg) * WN_DROWS;          // first tower row of the workgroup's two boards

    const int nch = cin >> 5;        // 32-channel chunks
    const int nst = (cin >> 4) * 16; // stages: k-steps x 16 frequencies
    const uint32_t row_bytes = (uint32_t)cin * 4;

    // ---- the U ring ----
    const char* wblk = reinterpret_cast<const char*>(wu) + (size_t)(cout0 >> 5) * nst * SW_STAGE;
    const uint32_t voff0 = lane * 16;
    u32x4 ring[D][2];
    auto load_stage = [&](u32x4(&slot)[2], const char* p) __attribute__((always_inline)) {
        u32x4 l0, l1;
        asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024"
                     : "=&v"(l0), "=&v"(l1)
                     : "v"(voff0), "s"(p)
                     : "memory");
        slot[0] = l0, slot[1] = l1;
    };
    // ---- LDS-DMA of an activation chunk: pixel rows of 144 B (eight 16-byte pieces of data, the ninth re-reads the eighth), 76
    // pieces per board row (8 pixels + 4 pieces of padding that re-read too): 1,216 pieces = 19 instructions of 64 ----
    const char* abase0 = reinterpret_cast<const char*>(in) + (size_t)row0 * row_bytes;
    uint32_t off_a[WN_P], dst_a[WN_P];
#pragma unroll
    for (int i = 0; i < WN_P; i++) {
        const int id = min(wave * WN_P + i, 18);
        const int sidx = id * 64 + lane, brow = sidx / 76, w = sidx - brow * 76, x = min(w / 9, 7), c = min(w - (w / 9) * 9, 7);
        off_a[i] = (uint32_t)(brow * 8 + x) * row_bytes + c * 16;
        dst_a[i] = id * 1024;
    }
    auto issue_chunk = [&](int ch, int buf) __attribute__((always_inline)) {  // chunk ch -> buffer buf
        const char* src = abase0 + (size_t)ch * 128;
        const uint32_t dst = WN_LDS_D + buf * WN_DBUF;
#pragma unroll
        for (int i = 0; i < WN_P; i++) glds16_asm(src + off_a[i], dst + dst_a[i]);
    };
    // The layer's last two DMA issues have no chunk left to fetch (they re-read the last one so that the count of loads in flight stays
    // fixed).  With skip rows they fetch HALF OF THE SKIP TILE instead, in the epilogue's final layout (piece = k-iteration k of this wave:
    // lane = (pixel k * 8 + (lane >> 3), couts 4 (lane & 7) ..)): k = 0..3 into the buffer the first of them overwrites, 4..7 into the
    // other's; the fifth instruction of a wave repeats its fourth.  Same instructions at the same places -- the wait counts do not change --
    // and the epilogue asks the fabric for half as many skip bytes at the moment every workgroup of the chip asks.
    uint32_t off_s[WN_P];
#pragma unroll
    for (int i = 0; i < WN_P; i++) off_s[i] = (uint32_t)((min(i, 3) * 8 + (lane >> 3)) * cout) * 4 + (lane & 7) * 16;
    const char* sbase0 = reinterpret_cast<const char*>(res) + ((size_t)row0 * cout + cout0) * 4;
    auto issue_chunk_or_skip = [&](int c) __attribute__((always_inline)) {
        const int sl = c - (nch - 2);  // 0, 1: the last two issues
        const bool sk = HAS_RES && sl >= 0;
        const char* src = sk ? sbase0 + (size_t)sl * 32 * cout * 4 : abase0 + (size_t)min(c + 2, nch - 1) * 128;
        const uint32_t dst = WN_LDS_D + (c & 1) * WN_DBUF;
#pragma unroll
        for (int i = 0; i < WN_P; i++) glds16_asm(src + (sk ? off_s[i] : off_a[i]), dst + dst_a[i]);
    };
    for (int i = tid; i < 2 * (WN_ZAREA / 16); i += 256)  // the two zero areas
        reinterpret_cast<f32x4*>(smem + WN_LDS_D + (i / (WN_ZAREA / 16)) * WN_DBUF + WN_DZERO)[i % (WN_ZAREA / 16)] = f32x4{0.f, 0.f, 0.f, 0.f};
    // chunk 0 goes out first, then the ring's first D stages, then chunk 1: V of k-step 0 needs chunk 0 alone and is made while the rest
    // is still on its way (-0.15 us per launch against waiting for everything)
    issue_chunk(0, 0);
#pragma unroll
    for (int d = 0; d < D; d++) load_stage(ring[d], wblk + (size_t)d * SW_STAGE);
    issue_chunk(1, 1);  // cin >= 64: at least two chunks

    // ---- the transform's item: tile tt (board tt >> 4, tile row (tt >> 2) & 3, tile column tt & 3), channel pair chp of the k-step's 8 ----
    // A 32-lane half of a wave -- the unit the LDS serves a ds_read_b64 / ds_write_b32 in -- is 8 consecutive tiles (two tile rows)
    // x 4 consecutive channel pairs, and that makes both of the transform's accesses free of bank conflicts (SQ_LDS_BANK_CONFLICT: 61 %
    // of the kernel's LDS cycles with 32 tiles x 1 pair per half):
    //   V writes (bank = dword mod 32): dword 20 tt + chp -- 20 tt mod 32 runs through the 8 multiples of 4, chp fills the 4 between
    //   patch reads (bank = dword mod 64, 2 dwords per lane): dword 36 (2 tx) + 304 (2 ty) + 2 chp + const -- 8 tx + 32 (ty & 1) + (0..7):
    //   the 64 B of padding per board row (304 = 8 x 36 + 16) are what separates the two tile rows
    const int tt = ((tid >> 5) & 3) * 8 + (tid & 7), chp = ((tid >> 7) << 2) + ((tid >> 3) & 3);
    // The patch pixel (i, j) of the tile is at base + i WN_RP + j SP; rows 0 / 3 and columns 0 / 3 of the patch can lie off the board
    // (first / last tile row or column): those reads go to the zero area, at the read's own address mod 256 (the same banks) + the
    // same offset.  One base register and four masks instead of sixteen addresses.
    const int tbase = ((tt >> 4) * 8 + 2 * ((tt >> 2) & 3) - 1) * WN_RP + (2 * (tt & 3) - 1) * SP + chp * 8;
    const int zbase = WN_DZERO + chp * 8;
    const int tdelta = tbase - zbase;  // address = zero area + (tdelta & mask): the masks are all ones where the patch row / column is on the board
    const int mr0 = ((tt >> 2) & 3) != 0 ? -1 : 255, mr3 = ((tt >> 2) & 3) != 3 ? -1 : 255, mc0 = (tt & 3) != 0 ? -1 : 255, mc3 = (tt & 3) != 3 ? -1 : 255;
    const int vwr = tt * WN_VP + chp * 4;  // where this item's hi pair goes inside a frequency's block (the lo pair 32 further)
    f32x2 dd[16];                          // the patch as f32, then (in place) B^T d, then B^T d B
    float vmax = 0.0f;                     // largest activation this thread has written
    // kp: which half (16 channels) of the chunk; slices 0..3 read a patch row each, 4..7 do the row transform of a column,
    // 8..15 the column transform of half a row each: two frequencies split into (hi, lo) and written.
    // A slice comes in two parts, issued behind the first and the second MFMA of its stage: the wave issues in order and the three
    // MFMAs of a stage wait for each other (one accumulator), so VALU work has to sit BETWEEN them to run in their shadow -- left
    // behind the third it overlapped with that one only, and the transform cost the loop 5 of its 34 us.
    auto transform_slice = [&](int slice, int part, int dbase, int kp, int vbase) __attribute__((always_inline)) {
        if (slice < 4) {
            if (part != 0) return;
            const int i = slice;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int mask = (i == 0 ? mr0 : i == 3 ? mr3 : -1) & (j == 0 ? mc0 : j == 3 ? mc3 : -1);
                dd[i * 4 + j] = *reinterpret_cast<const f32x2*>(smem + dbase + zbase + (tdelta & mask) + (i * WN_RP + j * SP + kp * 64));
            }
        } else if (slice < 8) {
            if (part != 1) return;
            const int j = slice - 4;  // column j: B^T over the rows
            const f32x2 d0 = dd[j], d1 = dd[4 + j], d2 = dd[8 + j], d3 = dd[12 + j];
            dd[j] = d0 - d2, dd[4 + j] = d1 + d2, dd[8 + j] = d2 - d1, dd[12 + j] = d1 - d3;
        } else {
            // slices 8..15: row (slice - 8) / 2, B over the columns, frequency 2 ((slice - 8) & 1) + part of that row
            const int i = (slice - 8) >> 1, l = ((slice - 8) & 1) * 2 + part;
            const f32x2 t0 = dd[i * 4], t1 = dd[i * 4 + 1], t2 = dd[i * 4 + 2], t3 = dd[i * 4 + 3];
            f32x2 x = l == 0 ? t0 - t2 : l == 1 ? t1 + t2 : l == 2 ? t2 - t1 : t1 - t3;
            // |x| <= 65504: x is a signed sum of four activations, and whoever wrote those capped them at WINO_ACT_MAX = 65504 / 4
            // (this kernel's epilogue, the stem's CONV_WINO_IN) -- no clamp, no range check among the loop's instructions
            const f16x2v hi = __builtin_convertvector(x, f16x2v);
            // lo = f16(x - hi): the difference is exact in f32 (hi is x rounded to 11 bits), so one mixed-precision fma per element
            // (v_fma_mixlo/hi_f16: f16 operand in, f32 arithmetic, f16 out) gives the bits of convert - subtract - convert
            // in asm: the compiler rewrites the fma back into convert - subtract - convert (5 instructions instead of 2).  The s_nop covers
            // the partial-register write ahead of whatever reads lo next (the compiler does not see inside).
            f16x2v lo;
            asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, -%1, 1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 0"
                : "=&v"(lo)
                : "v"(hi), "v"(x[0]), "v"(x[1]));
            char* q = smem + vbase + (i * 4 + l) * WN_VF + vwr;
            *reinterpret_cast<f16x2v*>(q) = hi;
            *reinterpret_cast<f16x2v*>(q + 32) = lo;
        }
    };

    f32x16 acc[16];
#pragma unroll
    for (int f = 0; f < 16; f++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[f][e] = 0.0f;

    // ---- prologue: V of k-step 0 ----
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * D + WN_P) : "memory");  // chunk 0 has landed (all but the 2 D + WN_P loads behind it)
#pragma unroll
    for (int s = 0; s < 16; s++) {
        transform_slice(s, 0, WN_LDS_D, 0, 0);
        transform_slice(s, 1, WN_LDS_D, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    const int vrd = r * WN_VP + h * 16;  // this lane's fragment inside a frequency's block: tile r, channels 8 h .. 8 h + 7 (lo 32 further)
    // One k-step: 16 stages on the V image at `vimg`, with the transform of the NEXT k-step (chunk image at `dbase`, half kp, into
    // the other V image) sliced in between.  JB: the stage's index inside the chunk body (0 or 16), which fixes the wait counts.
    auto kstep = [&](const char* wchunk, int vimg, int dbase, int kp, int vnext, auto jb_tag) __attribute__((always_inline)) {
        constexpr int JB = decltype(jb_tag)::value;
        // WN_PA stages of look-ahead on the V fragments (never across the k-step: the next image is being written): an LDS read
        // queues behind the transform slices' reads and writes of this wave (the LDS queue is in order), so one stage is not enough
        constexpr int PA = WN_PA, PR = PA + 1;
        frag vh[PR], vl[PR];
#pragma unroll
        for (int q = 0; q < PA; q++) {
            vh[q] = *reinterpret_cast<const frag*>(smem + vimg + q * WN_VF + vrd);
            vl[q] = *reinterpret_cast<const frag*>(smem + vimg + q * WN_VF + vrd + 32);
        }
#pragma unroll
        for (int f = 0; f < 16; f++) {
            const int cur = f % PR, nxt = (f + PA) % PR;
            if (f + PA < 16) {
                vh[nxt] = *reinterpret_cast<const frag*>(smem + vimg + (f + PA) * WN_VF + vrd);
                vl[nxt] = *reinterpret_cast<const frag*>(smem + vimg + (f + PA) * WN_VF + vrd + 32);
            }
            // all but the youngest 2 (D - 1) ring loads have returned -- plus, for the D stages whose own loads went out before this
            // body's LDS-DMA (issued between its stages 15 and 16), those WN_P younger DMA instructions
            {
                const int j = JB + f;  // a constant once the loop is unrolled: the branch below folds
                u32x4 r0 = ring[f % D][0], r1 = ring[f % D][1];
                if (j >= 16 && j < 16 + D) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r0), "+v"(r1) : "n"(2 * (D - 1) + WN_P));
                else asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r0), "+v"(r1) : "n"(2 * (D - 1)));
                ring[f % D][0] = r0, ring[f % D][1] = r1;
            }
            const frag uh = __builtin_bit_cast(frag, ring[f % D][0]);
            const frag ul = __builtin_bit_cast(frag, ring[f % D][1]);
            Mfma<T>::mac(ul, vh[cur], acc[f]);
            __builtin_amdgcn_sched_barrier(0);
            transform_slice(f, 0, dbase, kp, vnext);
            __builtin_amdgcn_sched_barrier(0);
            Mfma<T>::mac(uh, vl[cur], acc[f]);
            __builtin_amdgcn_sched_barrier(0);
            transform_slice(f, 1, dbase, kp, vnext);
            __builtin_amdgcn_sched_barrier(0);
            Mfma<T>::mac(uh, vh[cur], acc[f]);
            // refill D stages ahead: chunk pointer + a constant (two scalar instructions; clamped to the layer's last stage it was a
            // dependent chain of six per stage in a wave that issues in order).  Past the layer's end that reads the next cout block's
            // first stages, or the WINO_RING_STAGES stages of padding behind the last block: nobody uses them, and the count of loads
            // in flight stays fixed
            load_stage(ring[f % D], wchunk + (size_t)(JB + f + D) * SW_STAGE);
        }
    };
    for (int c = 0; c < nch; c++) {
        const int dcur = WN_LDS_D + (c & 1) * WN_DBUF, dnext = WN_LDS_D + ((c + 1) & 1) * WN_DBUF;
        const char* wchunk = wblk + (size_t)c * 32 * SW_STAGE;  // this chunk's 32 stages of U
        // k-step 2c on image 0; meanwhile V of k-step 2c + 1 (the chunk's second half) -> image 1
        kstep(wchunk, 0, dcur, 1, WN_VIMG, std::integral_constant<int, 0>{});
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // image 1 is complete, image 0 and chunk c are free
        issue_chunk_or_skip(c);  // chunk c + 2 -> the buffer chunk c was in; the last two issues: half of the skip tile (or a re-read nobody uses)
        // k-step 2c + 1 on image 1; meanwhile V of k-step 2c + 2 (the next chunk's first half) -> image 0
        kstep(wchunk, WN_VIMG, dnext, 0, 0, std::integral_constant<int, 16>{});
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    // the ring's last refills (nobody reads them) and the last DMA.  The ring's registers are operands of the wait: to the compiler
    // they are free from their last MFMA on, and it would park epilogue values in them while the loads are still on their way
    __builtin_amdgcn_sched_barrier(0);  // nothing of the epilogue above the drain: the ring's 64 registers are still taken there
#pragma unroll
    for (int d = 0; d < D; d++) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ring[d][0]), "+v"(ring[d][1])::"memory");
    __builtin_amdgcn_sched_barrier(0);
    // ---- epilogue ----
    // Everything the epilogue derives from the lane index is derived from an opaque copy made HERE: computed ahead of the loop (where
    // the compiler would hoist it) it would stay live through the loop, which has no register to spare.
    int elane = lane;
    asm volatile("" : "+v"(elane));
    const int er = elane & 31, eh = elane >> 5;
    // Output transform Y = A^T M A over the 16 frequencies f = 4 i + l (A^T = [1 1 1 0; 0 1 -1 -1]: Z[i][0] = M[i][0] + M[i][1] + M[i][2],
    // Z[i][1] = M[i][1] - M[i][2] - M[i][3], then the same over i), four accumulator elements (couts 8 g + 4 h ..) at a time, each
    // result straight into the transpose: lane (tile r, half h) holds couts 8 g + 4 h + i of its tile's four pixels, the wave's 128
    // pixels x 32 couts go through LDS as f32 (the V images are free now: 16 KiB per wave, 16-byte slots XOR-swizzled by the pixel) and
    // come back as (pixel, 8 consecutive couts) per lane -- whole 64-byte runs per store instruction, the direct kernel's epilogue.
    // (Written straight from the accumulator layout the output is 32 eight-byte stores per lane: 11.6 of 48 us.)
    // (no barrier here: every wave left the V images, which the transpose reuses, behind the last k-step's barrier)
    char* stage = smem + wave * 16384;
    const int board = er >> 4, ty = (er >> 2) & 3, tx = er & 3;
    // Final layout: lane = (pixel k * 8 + (lane >> 3), couts 4 pc .. 4 pc + 3 with pc = lane & 7): eight lanes cover a pixel's 128 bytes
    // (this wave's 32 couts), so every skip load and every store instruction moves whole 128-byte lines.  (As (pixel, 8 couts) per lane
    // and two 16-byte stores each, a store instruction wrote 16 bytes out of every 32: half-written lines cost 1.5 us per launch.)
    const int pc = elane & 7;
    const f32x4 bias4 = *reinterpret_cast<const f32x4*>(bias + cout0 + pc * 4);
    const f32x4 ds4 = *reinterpret_cast<const f32x4*>(bias + cout + cout0 + pc * 4);
    // the skip rows: requested AHEAD of the output transform (the ring's 64 registers are free from the drain on), used behind the
    // transpose: their latency hides behind the transform's ~800 VALU instructions
    f32x4 skip[16];
    constexpr int SK0 = 8;  // k-iterations whose skip rows are in LDS already (issue_chunk_or_skip)
    if (HAS_RES) {
#pragma unroll
        for (int k = SK0; k < 16; k++)
            skip[k] = *reinterpret_cast<const f32x4*>(res + ((size_t)row0 + k * 8 + (elane >> 3)) * (size_t)cout + cout0 + pc * 4);
    }
    // four accumulator elements (couts 8 g + 4 h ..) of every frequency at a time, each read out of its AGPR by an asm statement:
    // left to itself the compiler copies all sixteen 16-register accumulators into VGPRs at once and spills
#pragma unroll
    for (int g = 0; g < 4; g++) {
        f32x4 m[16];
#pragma unroll
        for (int f = 0; f < 16; f++)
#pragma unroll
            for (int i = 0; i < 4; i++) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(m[f][i]) : "a"(acc[f][g * 4 + i]));
        f32x4 z[4][2];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            z[i][0] = m[i * 4] + m[i * 4 + 1] + m[i * 4 + 2];
            z[i][1] = m[i * 4 + 1] - m[i * 4 + 2] - m[i * 4 + 3];
        }
#pragma unroll
        for (int pq = 0; pq < 4; pq++) {
            const int q = pq & 1;
            const f32x4 yv = (pq >> 1) == 0 ? z[0][q] + z[1][q] + z[2][q] : z[1][q] - z[2][q] - z[3][q];
            const int px = board * 64 + (2 * ty + (pq >> 1)) * 8 + 2 * tx + q;  // pixel row inside the workgroup's 128
            *reinterpret_cast<f32x4*>(stage + px * 128 + (((g * 2 + eh) ^ (px & 7)) << 4)) = yv;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // a wave reads back only what it wrote itself
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int px = k * 8 + (elane >> 3);
        const f32x4 a = *reinterpret_cast<const f32x4*>(stage + px * 128 + ((pc ^ (px & 7)) << 4));
        f32x4 sk4 = {0.f, 0.f, 0.f, 0.f};
        if (HAS_RES) {
            if (k < SK0)  // piece (k & 3) of the issue (k >> 2): buffer (nch - 2 + (k >> 2)) & 1, slot min(wave * WN_P + (k & 3), 18), this lane's 16 bytes
                sk4 = *reinterpret_cast<const f32x4*>(smem + WN_LDS_D + ((nch + (k >> 2)) & 1) * WN_DBUF + min(wave * WN_P + (k & 3), 18) * 1024 + elane * 16);
            else sk4 = skip[k];
        }
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            // the inverse weight scale is a power of two: the product is exact, the fma rounds once
            float x = __builtin_fmaf(a[j], ds4[j], bias4[j]);
            if (HAS_RES) x = x + sk4[j];
            x = x > 0.0f ? x : 0.0f;
            v[j] = x < WINO_ACT_MAX ? x : WINO_ACT_MAX;  // the next layer's transform relies on it (and counts nothing itself)
        }
        vmax = fmaxf(fmaxf(vmax, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out + ((size_t)row0 + px) * (size_t)cout + cout0 + pc * 4));
    }
    if (vmax >= WINO_ACT_MAX) atomicAdd(sat, 1u);  // an activation reached the cap somewhere in this thread's share
}

bool wino_supported(uint32_t bpad, uint32_t cin, uint32_t cout, uint32_t S) {
    return S == 8 && cin >= 64 && cin % 32 == 0 && cout % 128 == 0 && bpad % 2 == 0;
}

void launch_conv3x3_wino(const float* in, const void* wu, const float* bias, const float* res, float* out, uint32_t bpad, uint32_t cin,
                         uint32_t cout, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop, unsigned* sat) {
    typedef _Float16 H;
    const dim3 grid((bpad / 2) * (cout / 128));
    if (res)
        hipExtLaunchKernelGGL((conv3x3_wino_kernel<true>), grid, dim3(256), WN_LDS_TOTAL, st, ev_start, ev_stop, 0, in, (const H*)wu, bias, res, out,
                              sat, (int)cin, (int)cout);
    else
        hipExtLaunchKernelGGL((conv3x3_wino_kernel<false>), grid, dim3(256), WN_LDS_TOTAL, st, ev_start, ev_stop, 0, in, (const H*)wu, bias, res, out,
                              sat, (int)cin, (int)cout);
}

hipError_t prepare_wino() {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, WN_LDS_TOTAL);
    const hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, WN_LDS_TOTAL);
    return err != hipSuccess ? err : e2;
}

}  // namespace cattus
