// Gate B for a Winograd F(2x2, 3x3) split-precision conv: can a CU be fed?
//
// In the transformed domain a layer is 16 independent GEMMs (one per frequency f) M_f[cout][tile] = sum_cin U_f[cout][cin] V_f[tile][cin],
// and a workgroup has to hold all 16 accumulators of its (tile, cout) block until the output transform: 16 x (32 x 32) f32 = 256
// registers per wave for ONE 32 x 32 block.  There is no room for a 2 x 2 block per wave (1,024 registers), so every operand
// fragment feeds exactly one block: per (k-step of 16 channels, frequency) a wave needs a U fragment pair (hi, lo: 2 KiB) and a V
// fragment pair (2 KiB) for 3 MFMAs -- 1.33 KiB per MFMA against 0.67 KiB in conv3x3_splitw_kernel (2 x 2 blocks, 12 MFMAs per 8 KiB).
// This probe runs exactly that loop and nothing else (no transforms, no epilogue, operands resident): per "layer" 16 k-steps x 16
// frequencies = 256 stages of {2 global_load_dwordx4 (U, fragment order, L2-resident, ring of D stages), 2 ds_read_b128 (V),
// 3 v_mfma_f32_32x32x16_f16 into accumulator f}.  MFMA floor per layer: 256 x 3 x 32 = 24,576 cycles (the direct form: 55,296).
// Two wave arrangements: SHARE_V (the 4 waves of a workgroup hold 4 cout blocks of the same 32 tiles: V is one LDS image, U is
// 4 different streams) and SHARE_U (2 tile blocks x 2 cout blocks: each U stream is read by two waves).
// Also measured: the same loop without MFMAs (what the operand paths deliver by themselves) and the direct form's stage mix
// (4 + 4 fragments per 12 MFMAs) as the baseline on the same box.
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/bin/winograd_probe scripts/probes/winograd_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int KSTEPS = 16, NF = 16;
constexpr int STAGES = KSTEPS * NF;             // per layer
constexpr int STAGE_BYTES = 2048;               // U hi + lo fragment of one cout block
constexpr int V_BYTES = NF * 33 * 144;          // one 32-tile block's V image for one 32-channel chunk: [f][32 tiles + pad][144 B]

// MODE 0: Winograd loop; 1: the same without MFMAs (operand delivery alone); SHARE_U: waves (tile block w >> 1, cout block w & 1)
// VMODE 0: V fragments per stage; 1: once per 4 stages (a wave owns 4 frequencies x 4 cout blocks: a quarter of the LDS reads, the
// same U stream); 2: never (V in registers: the U stream alone under MFMAs).  UMODE 1: no U loads (the V reads alone under MFMAs).
template <int D, int MODE, bool SHARE_U, int VMODE = 0, int UMODE = 0>
__global__ void __launch_bounds__(256, 1) wino_kernel(const char* __restrict__ u, int layers, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2 * V_BYTES / 4; i += 256) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 7);
    __syncthreads();
    const int cblk = SHARE_U ? (blockIdx.x & 3) * 2 + (wave & 1) : (blockIdx.x & 1) * 4 + wave;  // one of 8 cout blocks
    const int tblk = SHARE_U ? wave >> 1 : 0;
    const char* vrow = smem + tblk * V_BYTES + (lane & 31) * 144 + (lane >> 5) * 16;
    floatx16 acc[NF];
#pragma unroll
    for (int f = 0; f < NF; f++)
#pragma unroll
        for (int j = 0; j < 16; j++) acc[f][j] = 0.0f;
    const uint32_t voff = lane * 16;
    u32x4 ring[D][2];
    u32x4 l0, l1;
    half8 puh[2] = {}, pvl[2] = {}, pvh[2] = {};  // MODE 2: operands of the two previous stages
    half8 vh = *reinterpret_cast<const half8*>(vrow), vl = *reinterpret_cast<const half8*>(vrow + 64);
#define LOAD_STAGE(slot, ptr)                                                                                   \
    do {                                                                                                        \
        asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024"             \
                     : "=&v"(l0), "=&v"(l1)                                                                     \
                     : "v"(voff), "s"(ptr)                                                                      \
                     : "memory");                                                                               \
        ring[slot][0] = l0, ring[slot][1] = l1;                                                                 \
    } while (0)
    for (int layer = 0; layer < layers; layer++) {
        const char* wp = u + ((size_t)layer * 8 + cblk) * STAGES * STAGE_BYTES;
#pragma unroll
        for (int d = 0; d < D; d++) LOAD_STAGE(d, wp + (size_t)d * STAGE_BYTES);
        for (int k = 0; k < KSTEPS; k++) {
#pragma unroll
            for (int f = 0; f < NF; f++) {
                const int s = k * NF + f;
                asm volatile("" ::: "memory");
                if (VMODE == 0 || (VMODE == 1 && (f & 3) == 0)) {
                    const int fv = VMODE == 1 ? wave * 4 + (f >> 2) : f;
                    vh = *reinterpret_cast<const half8*>(vrow + fv * 33 * 144 + (k & 1) * 32);
                    vl = *reinterpret_cast<const half8*>(vrow + fv * 33 * 144 + (k & 1) * 32 + 64);
                }
                u32x4 r0 = ring[f % D][0], r1 = ring[f % D][1];
                if (UMODE == 0) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r0), "+v"(r1) : "n"(2 * (D - 1)));
                if (MODE == 0) {
                    const half8 uh = __builtin_bit_cast(half8, r0), ul = __builtin_bit_cast(half8, r1);
                    acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ul, vh, acc[f], 0, 0, 0);
                    acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(uh, vl, acc[f], 0, 0, 0);
                    acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(uh, vh, acc[f], 0, 0, 0);
                } else if (MODE == 2) {
                    // the three terms of a stage spread over three stages: consecutive MFMAs belong to different accumulators
                    // (per accumulator the order of its terms is unchanged)
                    const half8 uh = __builtin_bit_cast(half8, r0), ul = __builtin_bit_cast(half8, r1);
                    acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ul, vh, acc[f], 0, 0, 0);
                    acc[(f + 15) & 15] = __builtin_amdgcn_mfma_f32_32x32x16_f16(puh[0], pvl[0], acc[(f + 15) & 15], 0, 0, 0);
                    acc[(f + 14) & 15] = __builtin_amdgcn_mfma_f32_32x32x16_f16(puh[1], pvh[1], acc[(f + 14) & 15], 0, 0, 0);
                    puh[1] = puh[0], pvh[1] = pvh[0];
                    puh[0] = uh, pvl[0] = vl, pvh[0] = vh;
                } else {  // consume the operands with four VALU instructions, no matrix pipe
                    acc[f][0] += __builtin_bit_cast(float, r0[0]) + __builtin_bit_cast(float, r1[0]) + (float)vh[0] + (float)vl[0];
                }
                const int sn = s + D < STAGES ? s + D : STAGES - 1;
                if (UMODE == 0) LOAD_STAGE(f % D, wp + (size_t)sn * STAGE_BYTES);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float sum = 0.0f;
#pragma unroll
    for (int f = 0; f < NF; f++)
#pragma unroll
        for (int j = 0; j < 16; j++) sum += acc[f][j];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = sum;
}

// Two waves per SIMD: 8 waves per workgroup, wave (w & 3) = cout block, (w >> 2) = which half of the 16 frequencies it accumulates
// (8 accumulators = 128 registers; the output transform would exchange partial sums through LDS once per layer).
template <int D>
__global__ void __launch_bounds__(512, 1) wino8_kernel(const char* __restrict__ u, int layers, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2 * V_BYTES / 4; i += 512) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 7);
    __syncthreads();
    const int cblk = (blockIdx.x & 1) * 4 + (wave & 3), fh = wave >> 2;
    const char* vrow = smem + (lane & 31) * 144 + (lane >> 5) * 16;
    floatx16 acc[8];
#pragma unroll
    for (int f = 0; f < 8; f++)
#pragma unroll
        for (int j = 0; j < 16; j++) acc[f][j] = 0.0f;
    const uint32_t voff = lane * 16;
    u32x4 ring[D][2];
    u32x4 l0, l1;
    constexpr int ST = KSTEPS * 8;  // this wave's stages per layer
    for (int layer = 0; layer < layers; layer++) {
        // stage (k, f') of this wave is global stage k * 16 + fh * 8 + f'
        const char* wp = u + ((size_t)layer * 8 + cblk) * STAGES * STAGE_BYTES + (size_t)fh * 8 * STAGE_BYTES;
#pragma unroll
        for (int d = 0; d < D; d++) LOAD_STAGE(d, wp + (size_t)((d >> 3) * 16 + (d & 7)) * STAGE_BYTES);
        for (int k = 0; k < KSTEPS; k++) {
#pragma unroll
            for (int f = 0; f < 8; f++) {
                const int s = k * 8 + f;
                asm volatile("" ::: "memory");
                const half8 vh = *reinterpret_cast<const half8*>(vrow + (fh * 8 + f) * 33 * 144 + (k & 1) * 32);
                const half8 vl = *reinterpret_cast<const half8*>(vrow + (fh * 8 + f) * 33 * 144 + (k & 1) * 32 + 64);
                u32x4 r0 = ring[f % D][0], r1 = ring[f % D][1];
                asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r0), "+v"(r1) : "n"(2 * (D - 1)));
                const half8 uh = __builtin_bit_cast(half8, r0), ul = __builtin_bit_cast(half8, r1);
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ul, vh, acc[f], 0, 0, 0);
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(uh, vl, acc[f], 0, 0, 0);
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(uh, vh, acc[f], 0, 0, 0);
                const int sn = s + D < ST ? s + D : ST - 1;
                LOAD_STAGE(f % D, wp + (size_t)((sn >> 3) * 16 + (sn & 7)) * STAGE_BYTES);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float sum = 0.0f;
#pragma unroll
    for (int f = 0; f < 8; f++)
#pragma unroll
        for (int j = 0; j < 16; j++) sum += acc[f][j];
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = sum;
}

// The direct form's stage on the same harness: 4 U fragments from L2 + 4 V fragments from LDS per 12 MFMAs, 144 stages per layer.
template <int D>
__global__ void __launch_bounds__(256, 1) direct_kernel(const char* __restrict__ u, int layers, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2 * V_BYTES / 4; i += 256) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 7);
    __syncthreads();
    const char* vrow = smem + (wave * 64 + (lane & 31)) * 144 + (lane >> 5) * 16;
    floatx16 acc[4];
#pragma unroll
    for (int f = 0; f < 4; f++)
#pragma unroll
        for (int j = 0; j < 16; j++) acc[f][j] = 0.0f;
    const uint32_t voff = lane * 16;
    u32x4 ring[D][4];
    u32x4 l0, l1, l2, l3;
#define LOAD_STAGE4(slot, ptr)                                                                                           \
    do {                                                                                                                 \
        asm volatile("global_load_dwordx4 %0, %4, %5\n\tglobal_load_dwordx4 %1, %4, %5 offset:1024\n\t"                  \
                     "global_load_dwordx4 %2, %4, %5 offset:2048\n\tglobal_load_dwordx4 %3, %4, %5 offset:3072"          \
                     : "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)                                                        \
                     : "v"(voff), "s"(ptr)                                                                               \
                     : "memory");                                                                                        \
        ring[slot][0] = l0, ring[slot][1] = l1, ring[slot][2] = l2, ring[slot][3] = l3;                                  \
    } while (0)
    constexpr int ST = 144;
    for (int layer = 0; layer < layers; layer++) {
        const char* wp = u + ((size_t)layer * 4 + (blockIdx.x & 3)) * ST * 4096;
#pragma unroll
        for (int d = 0; d < D; d++) LOAD_STAGE4(d, wp + (size_t)d * 4096);
        for (int s0 = 0; s0 < ST; s0 += D) {
#pragma unroll
            for (int d = 0; d < D; d++) {
                const int s = s0 + d;
                asm volatile("" ::: "memory");
                const int aoff = (d % 3) * 144 + ((d / 3) & 1) * 32;
                const half8 a0h = *reinterpret_cast<const half8*>(vrow + aoff), a0l = *reinterpret_cast<const half8*>(vrow + aoff + 64);
                const half8 a1h = *reinterpret_cast<const half8*>(vrow + 32 * 144 + aoff), a1l = *reinterpret_cast<const half8*>(vrow + 32 * 144 + aoff + 64);
                u32x4 r0 = ring[d][0], r1 = ring[d][1], r2 = ring[d][2], r3 = ring[d][3];
                asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "n"(4 * (D - 1)));
                const half8 b0 = __builtin_bit_cast(half8, r0), b1 = __builtin_bit_cast(half8, r1), b2 = __builtin_bit_cast(half8, r2),
                            b3 = __builtin_bit_cast(half8, r3);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1, a0h, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, a0l, acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, a0h, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b1, a1h, acc[1], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, a1l, acc[1], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b0, a1h, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b3, a0h, acc[2], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b2, a0l, acc[2], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b2, a0h, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b3, a1h, acc[3], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b2, a1l, acc[3], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b2, a1h, acc[3], 0, 0, 0);
                const int sn = s + D < ST ? s + D : ST - 1;
                LOAD_STAGE4(d, wp + (size_t)sn * 4096);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    float sum = 0.0f;
#pragma unroll
    for (int f = 0; f < 4; f++)
#pragma unroll
        for (int j = 0; j < 16; j++) sum += acc[f][j];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = sum;
}

template <typename K>
static double run(K kernel, const char* what, const char* dU, int grid, int layers, float* dOut, int lds, double mfma_floor_cycles, int threads = 256) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 6; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), lds, 0, dU, layers, dOut);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
    }
    const double us = best * 1000.0 / layers;
    printf("%-64s %7.2f us per layer  (MFMA floor %5.1f us at 2.0 GHz, %5.1f at 2.4)\n", what, us, mfma_floor_cycles / 2000.0, mfma_floor_cycles / 2400.0);
    return us;
}

int main(int argc, char** argv) {
    const int layers = argc > 1 ? atoi(argv[1]) : 20;
    const int grid = 256;
    const size_t bytes = (size_t)layers * 8 * STAGES * STAGE_BYTES;  // 4.2 MB per layer: U of 8 cout blocks (16 / 9 of the direct form's 2.36 MB)
    std::vector<_Float16> h(bytes / 2);
    for (size_t i = 0; i < h.size(); i++) h[i] = (_Float16)(((i * 2654435761u) >> 20 & 15) * 0.001f);
    char* dU;
    float* dOut;
    hipMalloc(&dU, bytes);
    hipMalloc(&dOut, (size_t)grid * 512 * 4);
    hipMemcpy(dU, h.data(), bytes, hipMemcpyHostToDevice);
    const int lds = 2 * V_BYTES;
    printf("U: %.1f MB per layer, %d layers; V image %d B per tile block and chunk\n", 8.0 * STAGES * STAGE_BYTES / 1e6, layers, V_BYTES);
    const double wf = 24576.0, df = 55296.0;
    run(direct_kernel<6>, "direct form, 2x2 blocks per wave, 12 MFMAs per 8 KiB, ring 6", dU, grid, layers, dOut, lds, df);
    run(wino_kernel<8, 0, false>, "winograd, waves share V (4 cout blocks), ring 8", dU, grid, layers, dOut, lds, wf);
    run(wino_kernel<16, 0, false>, "winograd, waves share V (4 cout blocks), ring 16", dU, grid, layers, dOut, lds, wf);
    run(wino_kernel<8, 0, true>, "winograd, waves 2 tile blocks x 2 cout blocks, ring 8", dU, grid, layers, dOut, lds, wf);
    run(wino_kernel<16, 0, true>, "winograd, waves 2 tile blocks x 2 cout blocks, ring 16", dU, grid, layers, dOut, lds, wf);
    run(wino_kernel<8, 2, false>, "winograd, share V, ring 8, terms pipelined over 3 stages", dU, grid, layers, dOut, lds, wf);
    run(wino_kernel<8, 2, true>, "winograd, 2 x 2, ring 8, terms pipelined over 3 stages", dU, grid, layers, dOut, lds, wf);
    run(wino_kernel<8, 0, false, 1>, "winograd, a wave = 4 frequencies x 4 cout blocks (V read once per 4 stages), ring 8", dU, grid, layers, dOut, lds, wf);
    run(wino_kernel<8, 0, false, 2>, "winograd, U stream alone under the MFMAs (V in registers), ring 8", dU, grid, layers, dOut, lds, wf);
    run(wino_kernel<8, 0, false, 0, 1>, "winograd, V reads alone under the MFMAs (U in registers)", dU, grid, layers, dOut, lds, wf);
    run(wino_kernel<8, 0, false, 2, 1>, "winograd, MFMAs alone (U and V in registers)", dU, grid, layers, dOut, lds, wf);
    run(wino8_kernel<8>, "winograd, 8 waves (2 per SIMD, 8 frequencies each), share V, ring 8", dU, grid, layers, dOut, lds, wf, 512);
    run(wino_kernel<16, 1, false>, "winograd operand streams alone (no MFMA), share V, ring 16", dU, grid, layers, dOut, lds, wf);
    run(wino_kernel<16, 1, true>, "winograd operand streams alone (no MFMA), 2 x 2, ring 16", dU, grid, layers, dOut, lds, wf);
    return 0;
}
