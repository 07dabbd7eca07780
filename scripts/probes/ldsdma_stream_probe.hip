// Gate for a board-resident 256-filter tower: how fast can every CU stream an L2-resident weight
// set into LDS with LDS-DMA, all CUs reading the SAME bytes, nothing consuming them?  A resident tower re-reads the
// layer's whole weight set (1.18 MB bf16 for 256 -> 256) per workgroup and layer; it pays only if a CU takes in well
// over 1.18 MB per 8.6 us (the MFMA time of a board's layer), i.e. >> 137 GB/s, and breaks even with the per-layer
// launches (18 us) at 66 GB/s.
// Build: hipcc --offload-arch=gfx950 -O2 -o scripts/probes/bin/ldsdma_stream_probe scripts/probes/ldsdma_stream_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int RING = 96 * 1024;  // bytes of LDS the stream cycles through

template <int NL, int INFLIGHT>
__global__ void __launch_bounds__(512) stream_kernel(const char* __restrict__ w, size_t set_bytes, int layers, int* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= NL) return;
    const size_t pieces = set_bytes / 1024;  // 1 KiB per wave-instruction
    const size_t per_wave = pieces / NL;
    constexpr int GROUP = INFLIGHT / 2;
    for (int layer = 0; layer < layers; layer++) {
        for (size_t p0 = 0; p0 < per_wave; p0 += GROUP) {
#pragma unroll
            for (int i = 0; i < GROUP; i++) {
                const size_t p = (p0 + i) * NL + wave;
                const char* src = w + p * 1024 + lane * 16;
                char* dst = smem + (p * 1024) % RING;
                __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(dst), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GROUP) : "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (sink && threadIdx.x == 0 && smem[17] == 123) *sink = 1;
}

template <int NL, int INFLIGHT>
static void bench(const char* dW, size_t set_bytes, int grid, int layers) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kernel<NL, INFLIGHT>), hipFuncAttributeMaxDynamicSharedMemorySize, RING);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((stream_kernel<NL, INFLIGHT>), dim3(grid), dim3(512), RING, 0, dW, set_bytes, layers, (int*)nullptr);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double bytes_per_wg = (double)set_bytes * layers;
    const double us_per_layer = best * 1000.0 / layers;
    printf("set %.2f MB, grid %4d, %d loader waves, %2d pieces in flight per wave: %8.2f us per layer, %6.1f GB/s per workgroup, %6.2f TB/s chip\n",
           set_bytes / 1e6, grid, NL, INFLIGHT, us_per_layer, bytes_per_wg / (best * 1e-3) / 1e9, bytes_per_wg * grid / (best * 1e-3) / 1e12);
}

int main() {
    const size_t max_bytes = 4 * 1179648;
    char* dW;
    hipMalloc(&dW, max_bytes);
    hipMemset(dW, 1, max_bytes);
    for (size_t set : {(size_t)1179648, (size_t)2 * 1179648}) {
        for (int grid : {256, 128, 512}) {
            bench<4, 16>(dW, set, grid, 41);
            bench<4, 32>(dW, set, grid, 41);
            bench<8, 16>(dW, set, grid, 41);
            bench<8, 32>(dW, set, grid, 41);
        }
    }
    hipFree(dW);
    return 0;
}
