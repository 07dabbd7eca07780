// Would the split conv's loop run faster with the WEIGHT fragments loaded straight from L2 into a register ring
// (pre-swizzled in fragment order, 4 KiB per stage and wave, every consumer wave of a workgroup fetching the same
// bytes) and only the ACTIVATION fragments read from LDS?  Today both go through LDS (ds_read ~66 % + DMA writes ~22 %
// of the LDS bandwidth, 24 barriers per launch for the weight ring).  This probe runs just that loop: 144 stages per
// "layer", a stage = 4 global_load_dwordx4 (B: two cout blocks x hi/lo) + 4 ds_read_b128 (A) + 12 MFMAs, ring depth D.
// The MFMA floor is 144 * 12 * 32 = 55,296 cycles per layer.
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/bin/bdirect_probe scripts/probes/bdirect_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int STAGES = 144;            // 8 chunks x 9 taps x 2 halves
constexpr int STAGE_U4 = 4 * 64;       // uint4 per stage: 4 fragments of 1 KiB
constexpr int SLAB_U4 = STAGES * STAGE_U4;  // one cout slab of one layer: 590 KB
constexpr int LDS_BYTES = 2 * 257 * 144;

template <int D, int THREADS>
__global__ void __launch_bounds__(THREADS) bdirect_kernel(const uint4* __restrict__ w, int layers, float* __restrict__ out,
                                                          long long* __restrict__ cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < LDS_BYTES / 4; i += THREADS) reinterpret_cast<float*>(smem)[i] = 0.0f;
    __syncthreads();
    if (wave >= 4) return;  // stand-ins for loader waves: they only cap the register budget
    const int slab = blockIdx.x & 3;
    // A fragment addresses: row (wave * 64 + rb * 32 + lane % 32) at pitch 144, k group lane / 32, hi at +0, lo at +64
    const char* arow0 = smem + (wave * 64 + (lane & 31)) * 144 + (lane >> 5) * 16;
    const char* arow1 = arow0 + 32 * 144;
    floatx16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 16; j++) acc[i][j] = 0.0f;
    const long long t0 = clock64();
    const uint32_t voff = lane * 16;
    for (int layer = 0; layer < layers; layer++) {
        const char* wp = reinterpret_cast<const char*>(w + ((size_t)layer * 4 + slab) * SLAB_U4);
        u32x4 ring[D][4];
        u32x4 l0, l1, l2, l3;
        // loads and their waits are hand-placed (asm): the compiler's own placement sinks the refill of a ring slot down to
        // its next use, which removes the look-ahead the ring exists for
#define LOAD_STAGE(slot, ptr)                                                                                              \
    do {                                                                                                                   \
    asm volatile("global_load_dwordx4 %0, %4, %5\n\tglobal_load_dwordx4 %1, %4, %5 offset:1024\n\t"                       \
                 "global_load_dwordx4 %2, %4, %5 offset:2048\n\tglobal_load_dwordx4 %3, %4, %5 offset:3072"               \
                 : "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)                                                             \
                 : "v"(voff), "s"(ptr)                                                                                     \
                 : "memory");                                                                                              \
    ring[slot][0] = l0, ring[slot][1] = l1, ring[slot][2] = l2, ring[slot][3] = l3;                                        \
    } while (0)
#pragma unroll
        for (int d = 0; d < D; d++) LOAD_STAGE(d, wp + (size_t)d * 4096);
        half8 a0h = *reinterpret_cast<const half8*>(arow0), a0l = *reinterpret_cast<const half8*>(arow0 + 64);
        half8 a1h = *reinterpret_cast<const half8*>(arow1), a1l = *reinterpret_cast<const half8*>(arow1 + 64);
        for (int s0 = 0; s0 < STAGES; s0 += D) {
#pragma unroll
            for (int d = 0; d < D; d++) {
                const int s = s0 + d;
                // A fragments of the NEXT stage are requested before this stage's MFMAs (one stage of look-ahead, as the real loop)
                const int aoff = ((d + 1) % 3) * 144 + (((d + 1) / 3) & 1) * 32;  // a tap shift and a half, as the real loop has
                asm volatile("" ::: "memory");  // LDS contents change in the real kernel: no hoisting of the fragment reads
                const half8 n0h = *reinterpret_cast<const half8*>(arow0 + aoff);
                const half8 n0l = *reinterpret_cast<const half8*>(arow0 + aoff + 64);
                const half8 n1h = *reinterpret_cast<const half8*>(arow1 + aoff);
                const half8 n1l = *reinterpret_cast<const half8*>(arow1 + aoff + 64);
                // the D - 1 younger stages (4 loads each) may stay in flight
                u32x4 r0 = ring[d][0], r1 = ring[d][1], r2 = ring[d][2], r3 = ring[d][3];
                asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "n"(4 * (D - 1)));
                const half8 b[4] = {__builtin_bit_cast(half8, r0), __builtin_bit_cast(half8, r1), __builtin_bit_cast(half8, r2),
                                    __builtin_bit_cast(half8, r3)};
                // a_lo * w_hi, a_hi * w_lo, a_hi * w_hi for 2 row blocks x 2 cout blocks
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, b[0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, b[2], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, b[0], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, b[2], acc[3], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, b[1], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, b[3], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, b[1], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, b[3], acc[3], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, b[0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, b[2], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, b[0], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, b[2], acc[3], 0, 0, 0);
                // refill this slot D stages ahead (the tail re-reads the last stage: the count of loads in flight stays fixed)
                const int sn = s + D < STAGES ? s + D : STAGES - 1;
                a0h = n0h, a0l = n0l, a1h = n1h, a1l = n1l;
                LOAD_STAGE(d, wp + (size_t)sn * 4096);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const long long t1 = clock64();
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 16; j++) sum += acc[i][j];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = sum;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int D, int THREADS>
static void bench(const uint4* dW, int grid, int layers, float* dOut, long long* dCyc) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&bdirect_kernel<D, THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 6; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((bdirect_kernel<D, THREADS>), dim3(grid), dim3(THREADS), LDS_BYTES, 0, dW, layers, dOut, dCyc);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    std::vector<long long> cyc((size_t)grid * 4);
    hipMemcpy(cyc.data(), dCyc, cyc.size() * 8, hipMemcpyDeviceToHost);
    std::sort(cyc.begin(), cyc.end());
    // clock64() counts at 100 MHz on gfx950 (s_memrealtime); convert with the event time instead
    printf("ring %2d stages, %3d threads, grid %4d, %2d layers: %7.2f us per layer (MFMA floor at 2.0 GHz: 27.6 us); wave time min/med/max %lld/%lld/%lld ticks\n",
           D, THREADS, grid, layers, best * 1000.0 / layers, cyc.front(), cyc[cyc.size() / 2], cyc.back());
}

int main(int argc, char** argv) {
    const int layers = argc > 1 ? atoi(argv[1]) : 20;
    const int grid = argc > 2 ? atoi(argv[2]) : 256;
    const size_t n_u4 = (size_t)layers * 4 * SLAB_U4;
    std::vector<_Float16> h(n_u4 * 8);
    for (size_t i = 0; i < h.size(); i++) h[i] = (_Float16)(((i * 2654435761u) >> 20 & 15) * 0.001f);
    uint4* dW;
    float* dOut;
    long long* dCyc;
    hipMalloc(&dW, n_u4 * 16);
    hipMalloc(&dOut, (size_t)grid * 256 * 4);
    hipMalloc(&dCyc, (size_t)grid * 4 * 8);
    hipMemcpy(dW, h.data(), n_u4 * 16, hipMemcpyHostToDevice);
    printf("weights: %.1f MB per layer (4 cout slabs of %.0f KB), %d layers\n", 4.0 * SLAB_U4 * 16 / 1e6, SLAB_U4 * 16 / 1e3, layers);
    bench<6, 512>(dW, grid, layers, dOut, dCyc);
    bench<8, 256>(dW, grid, layers, dOut, dCyc);
    bench<12, 256>(dW, grid, layers, dOut, dCyc);
    bench<16, 256>(dW, grid, layers, dOut, dCyc);
    bench<6, 256>(dW, grid, layers, dOut, dCyc);
    bench<4, 512>(dW, grid, layers, dOut, dCyc);
    return 0;
}
