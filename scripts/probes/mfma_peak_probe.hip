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
    // four different operand pairs of pseudo-random values in [-1, 1): consecutive MFMAs see different inputs, as a conv's do
    // (the same pair over and over toggles nothing between instructions and flatters the power)
    uint32_t x = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    auto rnd = [&]() {
        x = x * 1664525u + 1013904223u;
        return (float)(int)(x >> 8) * (1.0f / 8388608.0f) - 1.0f;
    };
    half8 a[4], b[4];
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
    floatx16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
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
