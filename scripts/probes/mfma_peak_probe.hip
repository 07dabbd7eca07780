// What does the dense f16 MFMA pipe SUSTAIN on this part?  The 2.5 PFLOP/s figure is 2.4 GHz x 1,024 SIMDs x 1,024 FLOP per
// cycle; under load the clock is set by the power / current limits.  This probe runs nothing but back-to-back
// v_mfma_f32_32x32x16_f16 (four independent accumulators per wave, operands in registers, no memory traffic) on every SIMD,
// in launches of ~50 us -- the length of a conv layer -- repeated for seconds, and reports the rate of the last launches and
// the clock the waves saw (s_memtime cycles over s_memrealtime's 100 MHz).
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/bin/mfma_peak_probe scripts/probes/mfma_peak_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// KIND 0: v_mfma_f32_32x32x16_f16, 1: v_mfma_f32_32x32x16_bf16, 2: v_mfma_f32_32x32x2_f32 (the exact-f32 tower's)
template <int KIND>
__global__ void __launch_bounds__(256) mfma_kernel(int iters, float* out, unsigned long long* clk) {
    const int lane = threadIdx.x & 63;
    half8 a, b;
    bf16x8 ab, bb;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        a[i] = (_Float16)(0.001f * (lane + i)), b[i] = (_Float16)(0.002f * (lane - i));
        ab[i] = (__bf16)(0.001f * (lane + i)), bb[i] = (__bf16)(0.002f * (lane - i));
    }
    const float af = 0.001f * lane, bf = 0.002f * (lane - 3);
    floatx16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        if constexpr (KIND == 0) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
        } else if constexpr (KIND == 1) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c3, 0, 0, 0);
        } else {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, c3, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0, clk[1] = r1 - r0;
}

template <int KIND>
static void run(const char* name, int iters, double flop_per_mfma, double nominal_tf, double seconds, float* dOut, unsigned long long* dClk) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    const double flop = 256.0 * 4 * iters * 4 * flop_per_mfma;  // one wave per SIMD
    double rate = 0;
    float ms = 0;
    const int reps = 50;
    int launches = 0;
    for (double elapsed = 0; elapsed < seconds;) {
        (void)hipEventRecord(e0, 0);
        for (int r = 0; r < reps; r++) hipLaunchKernelGGL(mfma_kernel<KIND>, dim3(256), dim3(256), 0, 0, iters, dOut, dClk);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        elapsed += ms * 1e-3;
        launches += reps;
        rate = flop * reps / (ms * 1e-3) / 1e12;
    }
    unsigned long long clk[2];
    (void)hipMemcpy(clk, dClk, 16, hipMemcpyDeviceToHost);
    printf("%-26s %4d MFMAs per wave and launch (%.1f us per launch), after %.1f s (%d launches): %6.0f TFLOP/s = %.3f of the nominal %.0f; "
           "in-kernel clock %.2f GHz\n",
           name, iters * 4, ms * 1e3 / reps, seconds, launches, rate, rate / nominal_tf, nominal_tf, clk[0] / (clk[1] * 10.0));
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    float* dOut;
    unsigned long long* dClk;
    (void)hipMalloc(&dOut, 256 * 256 * 4);
    (void)hipMalloc(&dClk, 16);
    // ~50 us per launch each; the f16 row twice (first and last) to show the run's drift
    run<0>("v_mfma_f32_32x32x16_f16", 640, 2.0 * 32 * 32 * 16, 2500.0, seconds, dOut, dClk);
    run<1>("v_mfma_f32_32x32x16_bf16", 640, 2.0 * 32 * 32 * 16, 2500.0, seconds, dOut, dClk);
    run<2>("v_mfma_f32_32x32x2_f32", 480, 2.0 * 32 * 32 * 2, 157.3, seconds, dOut, dClk);
    run<0>("v_mfma_f32_32x32x16_f16", 640, 2.0 * 32 * 32 * 16, 2500.0, seconds, dOut, dClk);
    return 0;
}
