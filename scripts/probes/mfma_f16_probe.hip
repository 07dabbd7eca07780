// Probe of v_mfma_f32_32x32x16_f16 on gfx950 (diagnostic, not product code): operand lane map, f16 subnormal
// operands, and how the instruction accumulates (against an exact f64 sum and against f32 fmaf chains).
// Build: hipcc --offload-arch=gfx950 -O2 -o scripts/probes/bin/mfma_f16_probe scripts/probes/mfma_f16_probe.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// A [32][K] row-major, B [K][32] row-major (k-major), C/D [32][32]; K a multiple of 16; one wave.
__global__ void mfma_chain(const _Float16* A, const _Float16* B, const float* C, float* D, int K) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    f32x16 acc;
    for (int e = 0; e < 16; e++) acc[e] = C[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r];
    for (int k0 = 0; k0 < K; k0 += 16) {
        f16x8 a, b;
        for (int j = 0; j < 8; j++) {
            a[j] = A[r * K + k0 + 8 * h + j];
            b[j] = B[(k0 + 8 * h + j) * 32 + r];
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    for (int e = 0; e < 16; e++) D[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = acc[e];
}

// the same sum on the exact-f32 MFMA (operands widened to f32): an in-order fmaf chain
__global__ void mfma_chain_f32(const _Float16* A, const _Float16* B, const float* C, float* D, int K) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    f32x16 acc;
    for (int e = 0; e < 16; e++) acc[e] = C[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r];
    for (int k0 = 0; k0 < K; k0 += 2) {
        const float a = (float)A[r * K + k0 + h], b = (float)B[(k0 + h) * 32 + r];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    for (int e = 0; e < 16; e++) D[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = acc[e];
}

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint64_t rnd() {
    uint64_t z = (rng_state += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
static double uni() { return (double)(rnd() >> 11) / 9007199254740992.0; }

struct Run {
    std::vector<_Float16> A, B;
    std::vector<float> C, D;
    int K;
};

static void run(Run& r, bool f32) {
    _Float16 *dA, *dB;
    float *dC, *dD;
    hipMalloc(&dA, r.A.size() * 2), hipMalloc(&dB, r.B.size() * 2), hipMalloc(&dC, 4096), hipMalloc(&dD, 4096);
    hipMemcpy(dA, r.A.data(), r.A.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dB, r.B.data(), r.B.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dC, r.C.data(), 4096, hipMemcpyHostToDevice);
    if (f32) hipLaunchKernelGGL(mfma_chain_f32, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, r.K);
    else hipLaunchKernelGGL(mfma_chain, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, r.K);
    r.D.resize(1024);
    hipMemcpy(r.D.data(), dD, 4096, hipMemcpyDeviceToHost);
    hipFree(dA), hipFree(dB), hipFree(dC), hipFree(dD);
}

static double exact(const Run& r, int i, int j) {
    long double s = r.C[i * 32 + j];
    for (int k = 0; k < r.K; k++) s += (long double)(float)r.A[i * r.K + k] * (long double)(float)r.B[k * 32 + j];
    return (double)s;
}

int main() {
    // ---- 1. lane map: small integers, exact in any order ----
    {
        Run r;
        r.K = 16;
        r.A.resize(32 * 16), r.B.resize(16 * 32), r.C.assign(1024, 0.0f);
        for (auto& x : r.A) x = (_Float16)(double)((int)(rnd() % 17) - 8);
        for (auto& x : r.B) x = (_Float16)(double)((int)(rnd() % 17) - 8);
        run(r, false);
        int bad = 0;
        for (int i = 0; i < 32; i++)
            for (int j = 0; j < 32; j++) bad += r.D[i * 32 + j] != (float)exact(r, i, j);
        printf("lane map (A[r][8h+j], B[8h+j][r], D row=(e&3)+8(e>>2)+4h col=r): %s (%d wrong of 1024)\n", bad ? "WRONG" : "ok", bad);
    }
    // ---- 2. subnormal f16 operands ----
    {
        Run r;
        r.K = 16;
        r.A.assign(32 * 16, (_Float16)9.5367431640625e-07 /* 2^-20, subnormal */), r.B.assign(16 * 32, (_Float16)1024.0), r.C.assign(1024, 0.0f);
        run(r, false);
        printf("subnormal A = 2^-20 x B = 1024, K = 16: D = %.9g (expected %.9g; 0 means flushed)\n", r.D[0], 16.0 * 9.5367431640625e-07 * 1024.0);
        r.A.assign(32 * 16, (_Float16)5.9604644775390625e-08 /* 2^-24, smallest subnormal */);
        r.B.assign(16 * 32, (_Float16)5.9604644775390625e-08);
        run(r, false);
        printf("2^-24 x 2^-24, K = 16: D = %.9g (expected %.9g)\n", r.D[0], 16.0 * 5.9604644775390625e-08 * 5.9604644775390625e-08);
        // random subnormal x normal
        for (auto& x : r.A) x = (_Float16)((uni() - 0.5) * 1e-4);
        for (auto& x : r.B) x = (_Float16)((uni() - 0.5) * 8.0);
        run(r, false);
        double worst = 0;
        for (int i = 0; i < 32; i++)
            for (int j = 0; j < 32; j++) {
                const double e = exact(r, i, j);
                worst = fmax(worst, fabs(r.D[i * 32 + j] - e) / (fabs(e) + 1e-30));
            }
        printf("random |a| < 5e-5 (subnormal) x |b| < 4: max rel error vs exact = %.3g\n", worst);
    }
    // ---- 3. one instruction: error against the exactly rounded sum, in ulps of the result ----
    for (int variant = 0; variant < 3; variant++) {
        Run r;
        r.K = 16;
        r.A.resize(32 * 16), r.B.resize(16 * 32), r.C.resize(1024);
        const double cmag = variant == 0 ? 0.0 : variant == 1 ? 4.0 : 1000.0;
        for (auto& x : r.A) x = (_Float16)(uni() * 2 - 1);
        for (auto& x : r.B) x = (_Float16)(uni() * 2 - 1);
        for (auto& x : r.C) x = (float)((uni() * 2 - 1) * cmag);
        run(r, false);
        Run q = r;
        run(q, true);
        double worst = 0, sum2 = 0, worst32 = 0, sum232 = 0;
        int exact_rounded = 0;
        for (int i = 0; i < 32; i++)
            for (int j = 0; j < 32; j++) {
                const double e = exact(r, i, j);
                const float er = (float)e;
                const double ulp = ldexp(1.0, ilogb((double)fabsf(er) + 1e-300) - 23);
                const double d = (r.D[i * 32 + j] - e) / ulp, d32 = (q.D[i * 32 + j] - e) / ulp;
                worst = fmax(worst, fabs(d)), sum2 += d * d;
                worst32 = fmax(worst32, fabs(d32)), sum232 += d32 * d32;
                exact_rounded += r.D[i * 32 + j] == er;
            }
        printf("one MFMA, |C| <= %g: f16 MFMA err max %.3f rms %.3f ulp (%d/1024 correctly rounded); f32 fmaf chain err max %.3f rms %.3f ulp\n",
               cmag, worst, sqrt(sum2 / 1024), exact_rounded, worst32, sqrt(sum232 / 1024));
    }
    // ---- 4. a conv-length chain: K = 2304 (and 6912 = the split tower's K') ----
    for (int K : {2304, 6912}) {
        Run r;
        r.K = K;
        r.A.resize(32 * K), r.B.resize(K * 32), r.C.assign(1024, 0.0f);
        for (auto& x : r.A) x = (_Float16)((uni() * 2 - 1) * 0.05);
        for (auto& x : r.B) x = (_Float16)(uni() < 0.5 ? 0.0 : uni() * 2);  // post-ReLU like
        run(r, false);
        Run q = r;
        run(q, true);
        double s16 = 0, s32 = 0, ref2 = 0;
        for (int i = 0; i < 32; i++)
            for (int j = 0; j < 32; j++) {
                const double e = exact(r, i, j);
                s16 += (r.D[i * 32 + j] - e) * (r.D[i * 32 + j] - e);
                s32 += (q.D[i * 32 + j] - e) * (q.D[i * 32 + j] - e);
                ref2 += e * e;
            }
        printf("chain K = %d: relative rms error f16 MFMA %.3g, f32 fmaf chain %.3g\n", K, sqrt(s16 / ref2), sqrt(s32 / ref2));
    }
    return 0;
}
