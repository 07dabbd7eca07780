// gfx950 kernels of the Cattus leaf evaluator, part 4: the Winograd F(2x2, 3x3) split conv with a 2 x 2 block per frequency (K1w4).
//
// conv3x3_wino_kernel (kernels_wino.hip) gives a wave the 16 frequencies of ONE (32 tiles x 32 couts) block: all 256 accumulator
// registers, every operand fragment feeds one accumulator, and the U stream -- 2 MB per CU and layer -- runs at what the XCDs' L2s
// deliver (bench.py: roofline.delivery 0.97-1.1 of the guide's 16.8-18.8 TB/s) while the matrix pipe is 0.3 busy.  This kernel
// spends the same 256 registers on 4 frequencies x (2 tile blocks x 2 cout blocks): a U fragment feeds two tile blocks, and the
// transformed input never exists in memory at all --
//   workgroup = 4 waves = 4 boards (64 tiles, 256 pixel rows) x 64 couts; wave q owns frequency ROW q (f = 4 q + l, l = 0..3) of
//       both tile blocks (boards 0-1, 2-3) and both cout blocks: 16 accumulators acc[l][tb][cb]
//   U: the same buffer and fragment order as K1w ([cout / 32][k-step x 16 + f][hi | lo][lane][8 f16]); a wave streams ITS four
//       frequencies of its two cout blocks: 1 MB per CU and layer, half of K1w's, each byte of it loaded by exactly one wave
//   V = B^T d B IN REGISTERS, in MFMA operand layout: lane (tile n, k-half h) needs row q of B^T d B for channels 8h..8h+7 of
//       its own tile -- the row combination t = d[ra] +- d[rb] of its four patch columns (read from LDS, ds_read_b128, 4 channels),
//       combined across the columns in f32 (the order of K1w: rows first, then columns), split into (hi, lo) and written into the
//       registers the MFMAs read.  No V image, no V write, no V fragment read, no barrier for V.
//   d: 32-channel chunks of the 256 pixel rows, global -> registers -> LDS -- and on the way ROW-COMBINED: wave q fetches board q, a
//       thread all eight rows of one column (8 loads per thread and chunk), forms the 4 frequency rows x 4 tile rows = 16 values t of
//       its column (neighbouring tiles share two of their four columns: 128 distinct values per board and channel where the tiles
//       would compute 256) and stores one image PER FREQUENCY ROW; wave q reads its own.  Layout [frequency row][tile block][board,
//       tile row][column]: pixel pitch 144 B (9 sixteen-byte slots), row pitch 72 slots = 8 mod 16, the second board one slot to the
//       right, a zero area behind each image for patch columns off the board -- found by enumeration over the lane groups of
//       ds_read_b128: with 32 DIFFERENT tiles per half wave that keeps the 16 lanes of every group on 16 different bank slots
//   per k-step (16 channels) a wave issues 48 MFMAs (as K1w), 16 + 4 global_load_dwordx4 (K1w: 32) -- ONE PER MFMA GAP, never in a
//       row -- and 16 ds_read_b128 in bursts of four, between which sit ~225 VALU instructions of transform: per V value a column
//       combination, half a conversion for the hi half, an f32 difference and half a conversion for the lo half, and per chunk the
//       row combinations (scripts/probes/gap_cost_probe.hip prices every one of them: the loop is the sum of its issue costs)
//   stage order of a k-step: (tb0,l0) (tb0,l1) (tb1,l0) (tb1,l1) (tb0,l2) (tb0,l3) (tb1,l2) (tb1,l3): a U stage (4 loads, 16
//       registers) lives for three stages, is refilled in the gaps of the stage behind its last use and has five stages to land -- a
//       ring of ONE k-step (64 registers); V of k-step s+1 is made in 24-gap phases (tb0: stages 2-5 of k-step s, tb1: stages 6-7 and
//       the next 0-1) into the l = 0,1 registers as they fall free and into the other of two l = 2,3 sets
//   epilogue: Z[q][c'] = row q of M A per lane, exchanged through LDS (128 KB, XOR-swizzled), Y = A^T Z summed across the waves
//       in K1w's order, * 2^-s + bias, + skip, ReLU, cap, whole-line f32 stores.
// Per accumulator the MFMA sequence is K1w's (k ascending; U_lo V_hi, U_hi V_lo, U_hi V_hi), V and Y are combined in K1w's order:
// the two kernels agree BIT FOR BIT (tests/test_hip_parity.py::test_winograd_kernels_agree_bit_for_bit).
#include "kernels.h"
#include "device_common.h"

#include <hip/hip_ext.h>

namespace cattus {

// A chunk in LDS is held ROW-COMBINED: for every frequency row q the image T_q[board][tile row ty][column x] = d[2 ty - 1 + ra_q][x] +-
// d[2 ty - 1 + rb_q][x] (B^T row q: what every tile of that tile row needs of its column x), made ONCE per chunk by the threads that
// fetched the rows -- instead of once per wave and tile column by the transform (neighbouring tiles share two of their four columns).
constexpr int W4_TROW = 8 * SP;               // pitch of an image row (a board's tile row: 8 pixels of 144 B) = 72 sixteen-byte slots = 8 mod 16
constexpr int W4_IMG = 9472;                  // one (frequency row, tile block) image: 2 boards x 4 tile rows x 1,152 B + the second board's 16 B, to 256
constexpr int W4_ZAREA = 512;                 // behind each image: its zero area (patch columns off the board)
constexpr int W4_IMGZ = W4_IMG + W4_ZAREA;    // 9,984 B
constexpr int W4_DBUF = 8 * W4_IMGZ;          // a chunk buffer: 4 frequency rows x 2 tile blocks = 79,872 B
constexpr int W4_LDS_LOOP = 2 * W4_DBUF;      // 159,744 B of the CU's 163,840
constexpr int W4_LDS_Z = 4 * 2 * 64 * 256;    // the epilogue's exchange: [wave][c'][tile][64 couts f32] = 131,072 B
constexpr int W4_LDS_TOTAL = W4_LDS_Z > W4_LDS_LOOP ? W4_LDS_Z : W4_LDS_LOOP;
static_assert(W4_IMG % 256 == 0 && W4_IMGZ % 256 == 0 && W4_DBUF % 256 == 0, "the zero area keeps a read's banks only if everything is 256-B aligned");
static_assert(W4_IMG >= 8 * W4_TROW + 16 && W4_ZAREA >= 256 + 128 && W4_LDS_LOOP <= 160 * 1024, "image / zero area / LDS sizes");

typedef __attribute__((ext_vector_type(2))) _Float16 w4_f16x2;

template <int V>
using w4_int = std::integral_constant<int, V>;

// ---- diagnostic build only (-DCATTUS_STAMPS, python -m cattus_amd.build --diag; scripts/stamps_w4.py): per-wave cycle stamps ----
#ifdef CATTUS_STAMPS
__device__ unsigned long long g_stamps_w4[1024 * 4 * 10];
#define W4_STAMP(i) st_[i] = __builtin_amdgcn_s_memtime()
#define W4_STAMP_RT(i) st_[i] = __builtin_amdgcn_s_memrealtime()
#else
#define W4_STAMP(i)
#define W4_STAMP_RT(i)
#endif

// What a workgroup of the PERSISTENT tower (tower_wino4_kernel below) knows besides the layer: where its board group's producers of
// the previous layer count themselves, and where it counts itself when its rows are written.
struct W4Sync {
    const unsigned* wait_word;  // ready[layer - 1][board group] (null: the layer's input was complete before the launch)
    unsigned* done_word;        // ready[layer][board group]
    unsigned* err;              // the launch's error word: set by a workgroup whose wait ran out of its budget
    unsigned wait_target;       // producers per board group: the cout groups
    unsigned spin_budget;       // polls (with s_sleep) before a wait gives up
};

// One layer on one workgroup.  RES: 0 = no skip rows, 1 = skip rows.
// PERSIST: called layer after layer from tower_wino4_kernel -- the activation loads and skip loads carry sc1 (they read what other CUs
// wrote during this launch: past this CU's L1), the stores carry sc1 (written through), and the workgroup waits for its board
// group's producers before its first activation load and counts itself in behind its last store.
template <int RES, bool PERSIST>
__device__ __forceinline__ void w4_layer(const float* __restrict__ in, const _Float16* __restrict__ wu, const float* __restrict__ bias,
                                         const float* res, float* out, unsigned* __restrict__ sat, int cin, int cout, char* smem,
                                         const W4Sync& sync, int wg, int nblk) {
    // wg of nblk: the workgroup's index in the layer's grid -- blockIdx.x of the per-layer launch; in the tower kernel a workgroup takes
    // several of them per layer when the layer has more tiles than the device has CUs
    typedef _Float16 T;
    typedef Mfma<T>::frag frag;
    constexpr bool has_res = RES == 1;

#ifdef CATTUS_STAMPS
    unsigned long long st_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // 8: behind the hand-off wait, 9: behind the last store's drain
#endif
    W4_STAMP(0);
    W4_STAMP_RT(6);
    int tid = threadIdx.x;
    // (an opaque copy: in the tower kernel everything derived from the thread index would otherwise be hoisted out of the layer loop
    // and stay live through all of it -- some 40 registers the loop does not have)
    asm volatile("" : "+v"(tid));
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);  // the wave = the frequency row it owns
    const int lane = tid & 63;

    const int ncg = cout >> 6;
    int logical = wg;
    if ((nblk & 7) == 0) logical = (wg & 7) * (nblk >> 3) + (wg >> 3);  // one XCD: all cout groups of a range of boards
    const int cout0 = (logical % ncg) * 64;   // the workgroup's 64 output channels
    const int row0 = (logical / ncg) * 256;   // first tower row of its four boards

    const int nch = cin >> 5;  // 32-channel chunks
    const int nks = cin >> 4;  // k-steps
    const uint32_t row_bytes = (uint32_t)cin * 4;

    // ---- the U ring: slot l = the U stage (k-step, frequency 4 q + l) = [cb0 hi, cb0 lo, cb1 hi, cb1 lo] ----
    const char* wb0 = reinterpret_cast<const char*>(wu) + ((size_t)(cout0 >> 5) * nks * 16 + 4 * q) * SW_STAGE;
    const size_t wcb = (size_t)nks * 16 * SW_STAGE;  // from cout block 0 to cout block 1
    const uint32_t voff0 = lane * 16, voff1 = lane * 16 + 4096;
    u32x4 ring[4][4];
    // the four U stages of a k-step are 2 KiB apart: l = 0, 1 as immediate offsets from the k-step's base, l = 2, 3 through a second
    // lane offset 4 KiB further (one base pointer per k-step and cout block: every scalar addition in this loop costs the wave an
    // issue slot that nothing hides)
    // One load of a U stage: j = 0..3 = [cb0 hi, cb0 lo, cb1 hi, cb1 lo].  In the loop a stage's four loads go out ONE PER MFMA GAP: four
    // in a row cost the wave ~21 cycles apiece (the CU's address unit takes a 1 KiB wave-load per 16 cycles, from all four waves), one
    // per gap ~14 (scripts/probes/gap_cost_probe.hip, modes 41-43)
    auto load_u1 = [&](u32x4& reg, const char* pk, int l, int j) __attribute__((always_inline)) {  // pk: the k-step's U (this wave's frequency row)
        const char* pj = j >> 1 ? pk + wcb : pk;
        const uint32_t voff = l >> 1 ? voff1 : voff0;
        const int imm = (l & 1) * 2048 + (j & 1) * 1024;
        u32x4 r;
        if (imm == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(r) : "v"(voff), "s"(pj) : "memory");
        else if (imm == 1024) asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=&v"(r) : "v"(voff), "s"(pj) : "memory");
        else if (imm == 2048) asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=&v"(r) : "v"(voff), "s"(pj) : "memory");
        else asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072" : "=&v"(r) : "v"(voff), "s"(pj) : "memory");
        reg = r;
    };
    auto load_ustage = [&](u32x4(&slot)[4], const char* pk, int l) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; j++) load_u1(slot[j], pk, l, j);
    };

    // ---- activation chunks: global -> registers -> LDS.  (LDS-DMA first: 10 pieces of 1 KiB per wave and chunk cost the issuing wave
    // ~60 cycles each, 5,000 of the loop's 41,000 cycles -- a global_load_dwordx4 + ds_write_b128 pair costs a third of that, moves no
    // padding and needs no per-lane offset table.)  A chunk = 256 pixel rows x 128 B = 2,048 sixteen-byte pieces, 8 per thread: piece
    // i of thread t = pixel row 32 i + (t >> 3), piece t & 7 -- a wave instruction reads eight whole 128-byte rows.  Image layout per
    // tile block as in K1w: pixel pitch 144 B, board rows of 8 pixels + 64 B, the second board one 16-B piece to the right, zero area behind.
    const char* abase0 = reinterpret_cast<const char*>(in) + (size_t)row0 * row_bytes;
    // wave q fetches board q of the workgroup's four: thread = (column x = (t >> 3) & 7, 16-byte piece t & 7), load i = board row i -- a wave
    // instruction reads the eight whole 128-byte rows of one board row; a thread ends up with all eight rows of its column
    const uint32_t aoff = (uint32_t)(q * 64 + ((tid >> 3) & 7)) * row_bytes + (tid & 7) * 16;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;  // the workgroup's LDS base (0 here: an add the compiler cannot fold)
    // where this thread's T values go: tile block q >> 1, board q & 1 of it (four image rows down, 16 B to the right), column, piece
    const uint32_t wdst = lds0 + (uint32_t)((q >> 1) * W4_IMGZ + (q & 1) * (4 * W4_TROW + 16) + ((tid >> 3) & 7) * SP + (tid & 7) * 16);
    constexpr int W4_NP = 8;  // loads per thread and chunk: they count in vmcnt like the ring's
    auto load_piece = [&](u32x4& reg, int ch, int i) __attribute__((always_inline)) {
        const char* src = abase0 + (size_t)ch * 128 + (size_t)i * 8 * row_bytes;
        u32x4 r;
        if (PERSIST) asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=&v"(r) : "v"(aoff), "s"(src) : "memory");
        else asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(r) : "v"(aoff), "s"(src) : "memory");
        reg = r;
    };
    auto load_chunk = [&](u32x4(&regs)[W4_NP], int ch) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < W4_NP; i++) load_piece(regs[i], ch, i);
    };
    // T unit u = (frequency row fq = u >> 2, tile row ty = u & 3) of this thread's column: t = fma(d[rb], sg, d[ra]) on rows 2 ty - 1 + (ra, rb)
    // -- K1w's instruction on K1w's operands (B^T row fq: 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3); a row off the board is +0.0:
    // above the board (fq 0, ty 0) the addend is the constant, below it (fq 3, ty 3) fma(+0, -1, d) is d itself
    auto t_unit = [&](const u32x4(&regs)[W4_NP], uint32_t target, int u) __attribute__((always_inline)) {
        const int fq = u >> 2, ty = u & 3;
        const int ra = fq == 0 ? 0 : fq == 2 ? 2 : 1, rb = fq == 3 ? 3 : fq == 2 ? 1 : 2;
        const int ya = 2 * ty - 1 + ra, yb = 2 * ty - 1 + rb;
        u32x4 t;
        if (yb > 7) t = regs[ya];
        else {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                if (ya < 0) asm("v_fma_f32 %0, %1, -1.0, 0" : "=v"(t[e]) : "v"(regs[yb][e]));
                else if (fq == 1) asm("v_fma_f32 %0, %1, 1.0, %2" : "=v"(t[e]) : "v"(regs[yb][e]), "v"(regs[ya][e]));
                else asm("v_fma_f32 %0, %1, -1.0, %2" : "=v"(t[e]) : "v"(regs[yb][e]), "v"(regs[ya][e]));
            }
        }
        typedef __attribute__((address_space(3))) u32x4 w4_lu32x4;
        *(w4_lu32x4*)(uintptr_t)(target + (uint32_t)(fq * 2 * W4_IMGZ + ty * W4_TROW)) = t;
    };
    for (int i = tid; i < 16 * (W4_ZAREA / 16); i += 256)  // the sixteen zero areas
        reinterpret_cast<f32x4*>(smem + (i / (W4_ZAREA / 16)) * W4_IMGZ + W4_IMG)[i % (W4_ZAREA / 16)] = f32x4{0.f, 0.f, 0.f, 0.f};
    // The ring's l = 0, 1 first -- the weights do not depend on anybody --, then (PERSIST) the wait for the board group's producers of the
    // previous layer, then chunk 0, chunk 1 and the ring's l = 2: the ORDER of the steady state (per k-step: l = 0 in stage 3, l = 1 in
    // stage 4, the odd k-step's chunk in stages 5-6, l = 2 in stage 7, l = 3 in the next k-step's stage 0 -- k-step 0 fetches its own), so
    // that the counted waits of k-step 0 are those of every even k-step (scripts/audit_inflight_regs.py caught the one order in which
    // they were not).  `pend` holds the chunk on its way throughout the loop: written to the freed buffer in stage 4 of the even k-step
    u32x4 first[W4_NP], pend[W4_NP];
    const char* wks = wb0;  // U of the k-step being multiplied
    load_ustage(ring[0], wks, 0);
    load_ustage(ring[1], wks, 1);
    if (PERSIST && sync.wait_word) {
        // one lane polls (sc1: past this CU's L1) until every cout group of this board group has counted itself in for the previous
        // layer, with a budget.  A wait that runs out raises the launch's error word AND GOES ON -- on rows that are not ready: the
        // launch then finishes at once instead of timing out workgroup after workgroup, every address it touches is its own, and the
        // host, which reads the error word behind every launch, throws the batch's outputs away and runs it on the per-layer launches.
        // A raised error word ends every other wait too.  The barrier stands between the poll and every load of the rows.
        if (tid == 0) {
            unsigned budget = sync.spin_budget, seen = 0, bad = 0;
            for (;;) {
                asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(seen) : "v"(sync.wait_word) : "memory");
                if (seen >= sync.wait_target) break;
                if ((budget & 63) == 0) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(bad) : "v"(sync.err) : "memory");
                if (bad != 0 || --budget == 0) {
                    atomicExch(sync.err, 1u);
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        asm volatile("s_barrier" ::: "memory");
    }
    W4_STAMP(8);
    load_chunk(first, 0);
    load_chunk(pend, 1);  // cin >= 128: at least four chunks
    load_ustage(ring[2], wks, 2);

    // ---- the transform's geometry: lane = (tile n of the tile block, k-half hh): board n >> 4, tile row (n >> 2) & 3, column n & 3 ----
    // Wave q reads ITS image: the four columns 2 tx - 1 .. 2 tx + 2 of image row (board, ty), at image + tbase + c SP.  Columns 0 / 3 can
    // lie off the board: those reads go to the image's zero area at the read's own address mod 256 (the same banks) -- folded into three
    // base registers (column class 0 / 1-2 / 3) that already point into the current chunk buffer.  (Pixel pitch 9 slots of 16 B, row pitch
    // 72 = 8 mod 16, the second board one slot to the right: the 16 lanes of every ds_read_b128 lane group on 16 different slots.)
    const int n = lane & 31, hh = lane >> 5;
    const int b2 = n >> 4, ty = (n >> 2) & 3, tx = n & 3;
    uint32_t cur[3];
    {
        const int tbase = (b2 * 4 + ty) * W4_TROW + (2 * tx - 1) * SP + b2 * 16 + hh * 32;
        const int tdelta = tbase - W4_IMG;
        const int mc0 = tx != 0 ? -1 : 255, mc3 = tx != 3 ? -1 : 255;
        const uint32_t img = lds0 + (uint32_t)(q * 2 * W4_IMGZ + W4_IMG);
        cur[0] = img + (uint32_t)(tdelta & mc0);
        cur[1] = img + (uint32_t)(tdelta + SP);
        cur[2] = img + (uint32_t)(((tdelta + 3 * SP) & mc3));
    }
    int bufstep = W4_DBUF;  // cur[] += bufstep at every chunk change, bufstep = -bufstep

    // V registers (MFMA B operands): [tb][l] for l = 0, 1; two sets [k-step parity][tb][l - 2] for l = 2, 3
    u32x4 vh01[2][2], vl01[2][2], vh23[2][2][2], vl23[2][2][2];
    f32x4 tbuf[2][4];           // a 4-channel group's row-combined patch, one f32x4 per patch column: [group parity], read a group ahead
    f32x4 xx;                   // a frequency's four values between the two halves of its slice
    float vmax = 0.0f;

    // The transform as a stream of PHASES (phase ph = 2 s + tb makes V of k-step s, tile block tb) of 24 slots each, one slot per MFMA
    // gap; a phase handles its two 4-channel groups one after the other (`slot` below).  Every LDS read issued during k-step s belongs
    // to V of k-step s + 1.
    auto freq_slot = [&](int sp, int tbv, int g, int l, int half) __attribute__((always_inline)) {
        // sp = parity of the k-step the phase makes V for (which l = 2,3 set), g = 4-channel group, l = frequency column
        if (half == 0) {
            // (one asm statement, conversions included: the compiler packs the C form of the column combination into v_pk_add_f32 -- 14
            // cycles of issue beside MFMAs for what two 4-cycle instructions do -- and puts an s_nop behind every asm statement whose
            // registers its own next instruction reads)
            const f32x4(&tt)[4] = tbuf[g];
            const int ca = l == 0 ? 0 : l == 2 ? 2 : 1, cb = l == 0 ? 2 : l == 1 ? 2 : l == 2 ? 1 : 3;
            // |x| <= 65504: x is a signed sum of four activations, capped where they were written at WINO_ACT_MAX = 65504 / 4
            float x0, x1, x2, x3;
            uint32_t h0, h1;
            if (l == 1)
                asm("v_add_f32 %2, %6, %10\n\tv_add_f32 %3, %7, %11\n\tv_add_f32 %4, %8, %12\n\tv_add_f32 %5, %9, %13\n\t"
                    "v_cvt_pk_f16_f32 %0, %2, %3\n\tv_cvt_pk_f16_f32 %1, %4, %5"
                    : "=&v"(h0), "=&v"(h1), "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3)
                    : "v"(tt[ca][0]), "v"(tt[ca][1]), "v"(tt[ca][2]), "v"(tt[ca][3]), "v"(tt[cb][0]), "v"(tt[cb][1]), "v"(tt[cb][2]), "v"(tt[cb][3]));
            else
                asm("v_sub_f32 %2, %6, %10\n\tv_sub_f32 %3, %7, %11\n\tv_sub_f32 %4, %8, %12\n\tv_sub_f32 %5, %9, %13\n\t"
                    "v_cvt_pk_f16_f32 %0, %2, %3\n\tv_cvt_pk_f16_f32 %1, %4, %5"
                    : "=&v"(h0), "=&v"(h1), "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3)
                    : "v"(tt[ca][0]), "v"(tt[ca][1]), "v"(tt[ca][2]), "v"(tt[ca][3]), "v"(tt[cb][0]), "v"(tt[cb][1]), "v"(tt[cb][2]), "v"(tt[cb][3]));
            xx = f32x4{x0, x1, x2, x3};
            u32x4& dst = l < 2 ? vh01[tbv][l] : vh23[sp][tbv][l - 2];
            dst[2 * g] = h0;
            dst[2 * g + 1] = h1;
        } else {
            // lo = f16(x - hi): the difference is exact in f32 (v_fma_mix_f32 takes hi as the f16 it is: no conversion back), one
            // v_cvt_pk_f16_f32 rounds two of them.  The same bits as v_fma_mixlo/hi_f16 (K1w), which round the same f32 difference -- but
            // those write half a register each and hold the vector issue for 9-10 cycles apiece where these take 4 and 5
            // (scripts/probes/gap_cost_probe.hip): 27 cycles per four values instead of 38.  In asm: the compiler turns the C form
            // into convert - subtract.
            const u32x4& hsrc = l < 2 ? vh01[tbv][l] : vh23[sp][tbv][l - 2];
            u32x4& dst = l < 2 ? vl01[tbv][l] : vl23[sp][tbv][l - 2];
            // (one statement, conversions included: behind an asm statement the compiler puts an s_nop in front of its own next VALU
            // instruction -- four cycles of issue each)
            float r0, r1, r2, r3;
            uint32_t lo0, lo1;
            asm("v_fma_mix_f32 %2, -%6, 1.0, %8 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %3, -%6, 1.0, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                "v_fma_mix_f32 %4, -%7, 1.0, %10 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %5, -%7, 1.0, %11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                "v_cvt_pk_f16_f32 %0, %2, %3\n\tv_cvt_pk_f16_f32 %1, %4, %5"
                : "=&v"(lo0), "=&v"(lo1), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
                : "v"(hsrc[2 * g]), "v"(hsrc[2 * g + 1]), "v"(xx[0]), "v"(xx[1]), "v"(xx[2]), "v"(xx[3]));
            dst[2 * g] = lo0, dst[2 * g + 1] = lo1;
        }
    };
    auto read_group = [&](f32x4(&dstv)[4], int tbv, int kp, int g) __attribute__((always_inline)) {
        const int imm = tbv * W4_IMGZ + kp * 64 + g * 16;
        // (cur[] are LDS addresses, the workgroup's base included: as `smem + offset` every read pair paid a v_add of the base)
        typedef const __attribute__((address_space(3))) f32x4 w4_lf32x4;
        dstv[0] = *(w4_lf32x4*)(uintptr_t)(cur[0] + imm);
        dstv[1] = *(w4_lf32x4*)(uintptr_t)(cur[1] + imm);
        dstv[2] = *(w4_lf32x4*)(uintptr_t)(cur[1] + imm + SP);
        dstv[3] = *(w4_lf32x4*)(uintptr_t)(cur[2] + imm);
    };
    // slot j of phase (sp = parity of the target k-step, tbv).  A group's twelve slots: 0 takes over its four columns (read ten slots
    // earlier: an LDS read that four waves issue at once takes ~200 cycles to come back, and a wave that waits for it stops issuing
    // MFMAs); 2 reads the columns of the NEXT group (the next phase's group 0 behind group 1); 4..11 the four frequencies
    auto slot = [&](int sp, int tbv, int j) __attribute__((always_inline)) {
        const int g = j / 12, jj = j % 12;
        if (jj == 0) {
            // (all four reads waited for once, as operands of one statement -- the compiler would count them down read by read)
            asm volatile("" : "+v"(tbuf[g][0]), "+v"(tbuf[g][1]), "+v"(tbuf[g][2]), "+v"(tbuf[g][3]));
        } else if (jj == 2) {
            // behind (s, tb0) comes (s, tb1), behind (s, tb1) comes (s + 1, tb0)
            const int nsp = g == 0 || tbv == 0 ? sp : sp ^ 1, ntb = g == 0 ? tbv : tbv ^ 1, ng = g ^ 1;
            read_group(tbuf[ng], ntb, nsp, ng);
        } else if (jj >= 4) {
            freq_slot(sp, tbv, g, (jj - 4) >> 1, (jj - 4) & 1);
        }
    };

    f32x16 acc[4][2][2];  // [l][tb][cb]
#pragma unroll
    for (int l = 0; l < 4; l++)
#pragma unroll
        for (int t2 = 0; t2 < 2; t2++)
#pragma unroll
            for (int c = 0; c < 2; c++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[l][t2][c][e] = 0.0f;

    // ---- prologue: chunk 0 has landed (everything but chunk 1's 8 loads and the ring's last 4), its T images go to buffer 0; the columns
    // of the first group, phase (0, tb0) whole, the first half of phase (0, tb1) -- its second half sits in stages 0, 1 of k-step 0, as in
    // every k-step ----
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(first[0]), "+v"(first[1]), "+v"(first[2]), "+v"(first[3]), "+v"(first[4]), "+v"(first[5]), "+v"(first[6]), "+v"(first[7])
                 : "n"(W4_NP + 4));  // all but chunk 1 and the ring's l = 2
#pragma unroll
    for (int u = 0; u < 16; u++) t_unit(first, wdst, u);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    read_group(tbuf[0], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 24; j++) slot(0, 0, j);
#pragma unroll
    for (int j = 0; j < 12; j++) slot(0, 1, j);

    // One k-step: 8 stages of 6 MFMAs; SP_ = parity of the k-step (compile time), which fixes the l = 2,3 set, the chunk half the
    // slices read and the wait counts.  `wcur` / `wnext`: U of this and of the next k-step (clamped to the last: nobody uses those refills).
    // LAST: the layer's last k-step -- nothing left to fetch, refill or transform, no chunk change.
    // The loads, ONE PER GAP (even stages: gaps 2-5, beside the slices' reads and conversions; odd stages: gaps 0-3), in the order
    //   stage 0: l = 3 of THIS k-step (its slot was last used in stage 7 of the one before)   stage 3: l = 0 of the next k-step
    //   stage 4: l = 1 of the next    stages 5, 6 (odd k-step): the chunk's 8 pieces           stage 7: l = 2 of the next
    // A U stage's loads are 5 stages (~1,500 cycles) ahead of its first use, a chunk 6 1/2 ahead of its stores.
    auto kstep = [&](const char* wcur, const char* wnext, int ch_next, auto sp_tag, auto last_tag) __attribute__((always_inline)) {
        constexpr int SP_ = decltype(sp_tag)::value;
        constexpr bool LAST = decltype(last_tag)::value != 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            // stage i: (tb, l) in the order (0,0) (0,1) (1,0) (1,1) (0,2) (0,3) (1,2) (1,3)
            const int tbv = (i >> 1) & 1, l = (i & 1) + ((i >> 2) << 1);
            const bool first_use = tbv == 0;
            if (i == 0 && SP_ == 1 && !LAST) {
                // the chunk change, at the start of the odd k-step: every LDS read issued during k-step s feeds V of k-step s + 1, so chunk c
                // was last read in the even k-step; every wave has seen its own pieces of chunk c + 1 land (they are older than ring loads
                // waited for since).  Behind the barrier chunk c + 1 is everybody's and chunk c's buffer takes chunk c + 2
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
                for (int k = 0; k < 3; k++) cur[k] += bufstep;
                bufstep = -bufstep;
            }
            if (first_use) {
                constexpr int NP = W4_NP;
                u32x4 r0 = ring[l][0], r1 = ring[l][1], r2 = ring[l][2], r3 = ring[l][3];
                // younger than this U stage in the queue (see the order above): l = 0: l = 1, [the chunk], l = 2; l = 1: [the chunk], l = 2, 3;
                // l = 2: l = 3 and the next k-step's l = 0; l = 3: the next k-step's l = 0, 1 -- eight loads, and the chunk's W4_NP in front
                // of the even k-step's l = 0, 1 (it went out in stages 5, 6 of the odd k-step before)
                if (l < 2 && SP_ == 0) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "n"(8 + NP));
                else if (l == 2 && SP_ == 0) {
                    // the even k-step's l = 2 is the first stage younger than the chunk loaded in the odd k-step before: its registers are
                    // operands of this wait, and the stores into the freed buffer follow it (this stage's gaps)
                    asm volatile("s_waitcnt vmcnt(8)"
                                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(pend[0]), "+v"(pend[1]), "+v"(pend[2]), "+v"(pend[3]), "+v"(pend[4]),
                                   "+v"(pend[5]), "+v"(pend[6]), "+v"(pend[7]));
                } else asm volatile("s_waitcnt vmcnt(8)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));
                ring[l][0] = r0, ring[l][1] = r1, ring[l][2] = r2, ring[l][3] = r3;
            }
            const frag uh0 = __builtin_bit_cast(frag, ring[l][0]), ul0 = __builtin_bit_cast(frag, ring[l][1]);
            const frag uh1 = __builtin_bit_cast(frag, ring[l][2]), ul1 = __builtin_bit_cast(frag, ring[l][3]);
            const frag vh = __builtin_bit_cast(frag, l < 2 ? vh01[tbv][l] : vh23[SP_][tbv][l - 2]);
            const frag vl = __builtin_bit_cast(frag, l < 2 ? vl01[tbv][l] : vl23[SP_][tbv][l - 2]);
            // the slices in this stage's six gaps: global gap 48 s + 6 i + e belongs to phase floor((gap + 36) / 24), slot (gap + 36) mod 24
            const int gp = 6 * i + 36;
            auto gap = [&](int e) __attribute__((always_inline)) {
                const int ph = (gp + e) / 24, j = (gp + e) % 24;  // ph = 1: (s, tb1); 2: (s + 1, tb0); 3: (s + 1, tb1)
                __builtin_amdgcn_sched_barrier(0);
                slot(ph == 1 ? SP_ : SP_ ^ 1, ph == 2 ? 0 : 1, j);
                // stage 4 of the even k-step: the pending chunk into the buffer the last chunk change left (two pieces per gap)
                // stages 4, 5, 6 of the even k-step: the pending chunk's sixteen T units into the buffer the last chunk change left, one per gap
                if (SP_ == 0 && i >= 4 && 6 * (i - 4) + e < 16) t_unit(pend, wdst + (uint32_t)(bufstep > 0 ? W4_DBUF : 0), 6 * (i - 4) + e);
                // this gap's load
                const int le = (i & 1) ? e : e - 2;
                if (le >= 0 && le < 4) {
                    if (i == 0) load_u1(ring[3][le], wcur, 3, le);
                    if (i == 3 && !LAST) load_u1(ring[0][le], wnext, 0, le);
                    if (i == 4 && !LAST) load_u1(ring[1][le], wnext, 1, le);
                    if (i == 7 && !LAST) load_u1(ring[2][le], wnext, 2, le);
                }
                if (SP_ == 1 && !LAST) {
                    if (i == 5) load_piece(pend[e], ch_next, e);
                    if (i == 6 && (e == 2 || e == 3)) load_piece(pend[4 + e], ch_next, 4 + e);
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            Mfma<T>::mac(ul0, vh, acc[l][tbv][0]);
            gap(0);
            Mfma<T>::mac(ul1, vh, acc[l][tbv][1]);
            gap(1);
            Mfma<T>::mac(uh0, vl, acc[l][tbv][0]);
            gap(2);
            Mfma<T>::mac(uh1, vl, acc[l][tbv][1]);
            gap(3);
            Mfma<T>::mac(uh0, vh, acc[l][tbv][0]);
            gap(4);
            Mfma<T>::mac(uh1, vh, acc[l][tbv][1]);
            gap(5);
        }
    };
    W4_STAMP(1);
    int c = 0;
    do {  // (a loop the compiler knows to run at least once: guarded, it keeps the prologue's V and ring values alive across it for the
          // path that skips it -- 32 registers, spilled in the tower kernel)
        const char* w1 = wks + (size_t)16 * SW_STAGE;                       // k-step 2c + 1
        const char* w2 = wks + (size_t)(c + 1 < nch ? 32 : 16) * SW_STAGE;  // k-step 2c + 2, or the last one again
        kstep(wks, w1, 0, w4_int<0>{}, w4_int<0>{});
        kstep(w1, w2, min(c + 2, nch - 1), w4_int<1>{}, w4_int<0>{});  // fetches chunk c + 2 (or the last one again: nobody reads it)
        wks = w2;
    } while (++c < nch);
    W4_STAMP(2);
    // The ring's last refills and the last chunk are on their way and nobody will use them (the loop's wait counts are the same in every
    // k-step).  Their registers stay what they are -- operands of the wait that stands behind the exchange writes below, some 3,500 cycles
    // from here: by then they have landed, where a wait at this point cost 1,500 cycles per layer.  (To the compiler the registers are free
    // from their last use on; as operands of that statement they are not, and it cannot park epilogue values in them while loads are
    // still in flight: kernels_wino.hip.)
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue ----
    int elane = lane;  // everything the epilogue derives from the lane index comes from an opaque copy made here (kernels_wino.hip)
    asm volatile("" : "+v"(elane));
    const int en = elane & 31, eh = elane >> 5;
    const int board = elane >> 4, pc = elane & 15;  // final layout: lane = (board, couts 4 pc .. 4 pc + 3) of one pixel position
    // (global address space spelled out: in the tower kernel the layer's pointers come out of a table in memory, the compiler takes them
    // for generic ones and its flat loads are followed by a wait for the LDS counter)
    typedef const __attribute__((address_space(1))) f32x4 w4_gf32x4;
    const f32x4 bias4 = *(w4_gf32x4*)(bias + cout0 + pc * 4);
    const f32x4 ds4 = *(w4_gf32x4*)(bias + cout + cout0 + pc * 4);
    // wave q finishes tile row q of every board: pixel q * 16 + k of a board for k = 0..15 (y = 2 q + (k >> 3), x = k & 7).  Row k of
    // this lane = the workgroup's base (scalar) + a 32-bit lane offset that grows by one row per k: one addition per load / store
    const size_t orow = (size_t)row0 + board * 64 + q * 16;
    const uint32_t cstride = (uint32_t)cout * 4;
    const uint32_t evoff = (uint32_t)((board * 64 + q * 16) * cout + pc * 4) * 4;
    const char* rbase = reinterpret_cast<const char*>(res + (size_t)row0 * cout + cout0);
    char* obase = reinterpret_cast<char*>(out + (size_t)row0 * cout + cout0);
    // the 16 skip rows of this lane: one load in front of each of the 16 groups of the exchange below -- issued in one burst they
    // cost the wave ~70 cycles apiece (four waves' 64 KiB through the CU's one address unit), between the groups' VALU work a slot each
    f32x4 skip[16];
    auto load_skip = [&](int k) __attribute__((always_inline)) {
        if (PERSIST) asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=&v"(skip[k]) : "v"(evoff + k * cstride), "s"(rbase) : "memory");
        else skip[k] = *reinterpret_cast<const f32x4*>(res + (orow + k) * (size_t)cout + cout0 + pc * 4);
    };
    W4_STAMP(3);
    asm volatile("s_barrier" ::: "memory");  // every wave has left the chunk buffers (no LDS read of the loop is outstanding: they fed VALU work long done)
    // Z[q][c'] = (row q of M) A: Z[.][0] = M[q][0] + M[q][1] + M[q][2], Z[.][1] = M[q][1] - M[q][2] - M[q][3] (K1w's order), four
    // accumulator elements (couts 8 g + 4 eh ..) at a time, each accumulator read out of its AGPR by an asm statement
    // exchange layout: [q][c'][tile 0..63][16 units of 4 couts], the unit index XORed with the tile's index inside its board
#pragma unroll
    for (int tbv = 0; tbv < 2; tbv++)
#pragma unroll
        for (int cb = 0; cb < 2; cb++)
#pragma unroll
            for (int g = 0; g < 4; g++) {
                if (has_res) load_skip(tbv * 8 + cb * 4 + g);
                f32x4 m[4];
#pragma unroll
                for (int l = 0; l < 4; l++)
#pragma unroll
                    for (int e = 0; e < 4; e++) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(m[l][e]) : "a"(acc[l][tbv][cb][g * 4 + e]));
                const f32x4 z0 = m[0] + m[1] + m[2], z1 = m[1] - m[2] - m[3];
                const int unit = (cb * 8 + g * 2 + eh) ^ (en & 15);
                char* zp = smem + ((size_t)(q * 2) * 64 + tbv * 32 + en) * 256 + (unit << 4);
                *reinterpret_cast<f32x4*>(zp) = z0;
                *reinterpret_cast<f32x4*>(zp + 64 * 256) = z1;
            }
    // the loop's leftover loads (above) and the asm skip loads (the compiler does not count either)
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(ring[0][0]), "+v"(ring[0][1]), "+v"(ring[0][2]), "+v"(ring[0][3]), "+v"(ring[1][0]), "+v"(ring[1][1]), "+v"(ring[1][2]),
                   "+v"(ring[1][3]), "+v"(ring[2][0]), "+v"(ring[2][1]), "+v"(ring[2][2]), "+v"(ring[2][3]), "+v"(ring[3][0]), "+v"(ring[3][1]),
                   "+v"(ring[3][2]), "+v"(ring[3][3]), "+v"(pend[0]), "+v"(pend[1]), "+v"(pend[2]), "+v"(pend[3]), "+v"(pend[4]), "+v"(pend[5]),
                   "+v"(pend[6]), "+v"(pend[7])::"memory");
    if (PERSIST && has_res) {
        asm volatile(""
                     : "+v"(skip[0]), "+v"(skip[1]), "+v"(skip[2]), "+v"(skip[3]), "+v"(skip[4]), "+v"(skip[5]), "+v"(skip[6]), "+v"(skip[7]), "+v"(skip[8]),
                       "+v"(skip[9]), "+v"(skip[10]), "+v"(skip[11]), "+v"(skip[12]), "+v"(skip[13]), "+v"(skip[14]), "+v"(skip[15]));
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    W4_STAMP(4);
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int rp = k >> 3, txo = (k & 7) >> 1, cp = k & 1;  // output row parity inside the tile, tile column, output column parity
        const int tile16 = q * 4 + txo;
        const char* zb = smem + ((size_t)cp * 64 + board * 16 + tile16) * 256 + ((pc ^ tile16) << 4);
        const f32x4 za = *reinterpret_cast<const f32x4*>(zb + (size_t)(rp + 0) * 2 * 64 * 256);
        const f32x4 zc = *reinterpret_cast<const f32x4*>(zb + (size_t)(rp + 1) * 2 * 64 * 256);
        const f32x4 zd = *reinterpret_cast<const f32x4*>(zb + (size_t)(rp + 2) * 2 * 64 * 256);
        const f32x4 a = rp == 0 ? za + zc + zd : za - zc - zd;  // Y[0] = Z0 + Z1 + Z2, Y[1] = Z1 - Z2 - Z3
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float x = __builtin_fmaf(a[j], ds4[j], bias4[j]);  // the inverse weight scale is a power of two: the fma rounds once
            if (has_res) x = x + skip[k][j];
            x = x > 0.0f ? x : 0.0f;
            v[j] = x < WINO_ACT_MAX ? x : WINO_ACT_MAX;  // the next layer's transform relies on it
        }
        vmax = fmaxf(fmaxf(vmax, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
        // PERSIST: written through (sc1), every 128-byte line whole by this one instruction -- what the next layer's sc1 loads on other CUs
        // may read once this workgroup has counted itself in.  Else plain stores: the rows are read again by the four cout-group workgroups
        // of this board group in the next launch, which run on this XCD (the block-index map) -- left in its L2 they cost 0.4 us per launch
        // less than non-temporal ones (K1w: the reverse)
        // (the s_nop: a store of more than 8 bytes reads its data registers a cycle or two after it issues, and gfx940 wants two wait
        // states before a VALU instruction overwrites them -- the compiler pads its own stores, it cannot see into this one: without it
        // the next pixel's values, computed into the same registers, went out with this pixel's address)
        if (PERSIST) asm volatile("global_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(evoff + k * cstride), "v"(v), "s"(obase) : "memory");
        else *reinterpret_cast<f32x4*>(out + (orow + k) * (size_t)cout + cout0 + pc * 4) = v;
    }
    if (vmax >= WINO_ACT_MAX) atomicAdd(sat, 1u);  // an activation reached the cap somewhere in this thread's share
    if (PERSIST) {
        // every storing wave drains its stores, the workgroup meets, ONE lane counts the workgroup in (agent scope): the order the
        // guide's hand-off table was measured in.  The barrier also keeps the next layer's LDS writes behind this layer's last LDS reads
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        W4_STAMP(9);
        asm volatile("s_barrier" ::: "memory");
        if (tid == 0) __hip_atomic_fetch_add(sync.done_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef CATTUS_STAMPS
    W4_STAMP(5);
    W4_STAMP_RT(7);
    if (lane == 0 && blockIdx.x < 1024) {
#pragma unroll
        for (int i = 0; i < 10; i++) g_stamps_w4[((size_t)blockIdx.x * 4 + q) * 10 + i] = st_[i];
    }
#endif
}

template <bool HAS_RES>
__global__ void __launch_bounds__(256, 1)
    conv3x3_wino4_kernel(const float* __restrict__ in, const _Float16* __restrict__ wu, const float* __restrict__ bias,
                         const float* res, float* out, unsigned* __restrict__ sat, int cin, int cout) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const W4Sync none{};
    w4_layer<HAS_RES ? 1 : 0, false>(in, wu, bias, res, out, sat, cin, cout, smem, none, (int)blockIdx.x, (int)gridDim.x);
}

// ---- the whole Winograd tower in ONE launch (round 5; the review's "persistent tower") ----
// grid = one workgroup per (board group, cout group), all resident (one per CU: the evaluator uses it only while the grid fits the
// device and serialises such launches per device); a workgroup keeps its tile through all layers.  Its only dependency is on the
// other cout groups of its own board group -- their outputs are its next input -- so the hand-off is a counter per (layer, board
// group): a workgroup counts itself in behind its last (sc1) store and waits for the count to reach the number of cout groups
// before its first activation load of the next layer; the ring's first U stages go out BEFORE that wait.  Board groups drift apart:
// no chip-wide moment at which every workgroup reads its first operands or writes its tile.  Every wait has a budget
// (W4Sync::spin_budget); a launch one of whose waits ran out says so in `err` (and finishes at once, on garbage) and the host runs
// the batch again on the per-layer launches.
__global__ void __launch_bounds__(256, 1)
    tower_wino4_kernel(const Wino4TowerLayer* __restrict__ layers, int nlayers, unsigned* ready, unsigned* err, unsigned* __restrict__ sat, int filters,
                       unsigned spin_budget, int tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // `tiles` = workgroups of a layer's grid (boards / 4 x cout groups).  More of them than the device has CUs (config 5's shape: 768):
    // the launch is as many workgroups as fit, and a workgroup takes tile blockIdx.x + r gridDim.x in round r of every layer.  The
    // hand-off never waits forward: a tile of layer n depends on tiles of layer n - 1 only, every workgroup walks layers and rounds in
    // the same order, so whoever is waited for is running or done.  (tiles and gridDim.x multiples of 8: a tile's XCD is blockIdx.x & 7
    // in every round, and the cout groups of a board group sit on one XCD as in the per-layer launches.)
    const int ncg = filters >> 6, ngroups = tiles / ncg;
    auto group_of = [&](int wg) {
        int logical = wg;
        if ((tiles & 7) == 0) logical = (wg & 7) * (tiles >> 3) + (wg >> 3);  // the map of w4_layer
        return logical / ncg;
    };
    // a residual block = two layers: the first without skip rows, the second with (two instantiations of the layer: decided at run time,
    // the skip registers' undefined values on the path without them cost 32 spilled registers)
    auto sync_of = [&](int layer, int group) {
        W4Sync sync;
        sync.wait_word = layer > 0 ? ready + (size_t)(layer - 1) * ngroups + group : nullptr;
        sync.done_word = ready + (size_t)layer * ngroups + group;
        sync.err = err, sync.wait_target = (unsigned)ncg, sync.spin_budget = spin_budget;
        return sync;
    };
    for (int layer = 0; layer + 1 < nlayers; layer += 2) {
        const Wino4TowerLayer A = layers[layer];
        for (int wg = blockIdx.x; wg < tiles; wg += gridDim.x)
            w4_layer<0, true>(A.in, reinterpret_cast<const _Float16*>(A.wu), A.bias, nullptr, A.out, sat, filters, filters, smem, sync_of(layer, group_of(wg)), wg, tiles);
        const Wino4TowerLayer B = layers[layer + 1];
        for (int wg = blockIdx.x; wg < tiles; wg += gridDim.x)
            w4_layer<1, true>(B.in, reinterpret_cast<const _Float16*>(B.wu), B.bias, B.res, B.out, sat, filters, filters, smem, sync_of(layer + 1, group_of(wg)), wg, tiles);
    }
}

bool wino4_supported(uint32_t bpad, uint32_t cin, uint32_t cout, uint32_t S) {
    return S == 8 && cin >= 128 && cin % 32 == 0 && cout >= 128 && cout % 64 == 0 && bpad % 4 == 0;  // 64 filters: the resident tower
}

void launch_conv3x3_wino4(const float* in, const void* wu, const float* bias, const float* res, float* out, uint32_t bpad, uint32_t cin,
                          uint32_t cout, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop, unsigned* sat) {
    typedef _Float16 H;
    const dim3 grid((bpad / 4) * (cout / 64));
    if (res)
        hipExtLaunchKernelGGL((conv3x3_wino4_kernel<true>), grid, dim3(256), W4_LDS_TOTAL, st, ev_start, ev_stop, 0, in, (const H*)wu, bias, res, out,
                              sat, (int)cin, (int)cout);
    else
        hipExtLaunchKernelGGL((conv3x3_wino4_kernel<false>), grid, dim3(256), W4_LDS_TOTAL, st, ev_start, ev_stop, 0, in, (const H*)wu, bias, res, out,
                              sat, (int)cin, (int)cout);
}

// The launch's grid for `tiles` tiles on `cus` CUs: all of them when they fit, else the largest multiple of 8 that does (one workgroup
// per CU, all resident: what the hand-off counters rely on), each workgroup taking several tiles per layer.
static uint32_t tower_grid(uint32_t tiles, uint32_t cus) { return tiles <= cus ? tiles : cus & ~7u; }

bool wino4_tower_fits(uint32_t bpad, uint32_t filters, uint32_t cus) {
    const uint32_t tiles = (bpad / 4) * (filters / 64);
    return tiles <= cus || (cus >= 8 && tiles % 8 == 0);  // (tiles % 8: the block-index map keeps a tile on one XCD through the rounds)
}

void launch_tower_wino4(const Wino4TowerLayer* d_layers, uint32_t nlayers, unsigned* ready, unsigned* err, unsigned* sat, uint32_t bpad, uint32_t filters,
                        uint32_t spin_budget, uint32_t cus, hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
    const uint32_t tiles = (bpad / 4) * (filters / 64);
    hipExtLaunchKernelGGL(tower_wino4_kernel, dim3(tower_grid(tiles, cus)), dim3(256), W4_LDS_TOTAL, st, ev_start, ev_stop, 0, d_layers, (int)nlayers, ready, err,
                          sat, (int)filters, spin_budget, (int)tiles);
}

#ifdef CATTUS_STAMPS
extern "C" __attribute__((visibility("default"))) int cattus_hip_debug_stamps_w4(unsigned long long* out, size_t n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps_w4), n * sizeof(unsigned long long));
}
#endif

hipError_t prepare_wino4() {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino4_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS_TOTAL);
    const hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino4_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS_TOTAL);
    const hipError_t e3 = hipFuncSetAttribute(reinterpret_cast<const void*>(&tower_wino4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS_TOTAL);
    return err != hipSuccess ? err : e2 != hipSuccess ? e2 : e3;
}

}  // namespace cattus
