// Stand-alone probe (diagnostic, not product): what one wave per SIMD can issue between v_mfma_f32_32x32x16_f16 instructions before
// the gap grows past the MFMA's 32 cycles -- with the fillers of the Winograd layer's input transform (kernels_wino4.hip): v_fma_f32
// with an SGPR operand, v_sub_f32, v_cvt_pk_f16_f32, v_fma_mixlo/hi_f16, ds_read_b128 (4 waves of a workgroup at once, the kernel's
// pitch), global_load_dwordx4 -- singly, in the kernel's pattern per 12 gaps (8, 8, 4 reads, 4 reads, 6, 4, 6, 4, 6, 4, 6, 4) and in a
// balanced one (the same instructions, <= 5 + one read per gap).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/probes/bin/gap_cost_probe scripts/probes/gap_cost_probe.hip && scripts/probes/bin/gap_cost_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(4))) unsigned u4;

#define SB __builtin_amdgcn_sched_barrier(0)

struct Regs {
    float x[16];   // filler values
    float y[8];
    unsigned h[4];  // packed halves
    f4 rd[8];       // LDS read targets
    u4 g[4];        // global load targets
    double d[8];    // 64-bit pairs
};

// (every filler accumulates into a register of its own: outputs that nothing reads would all land in one register, with an s_nop between them)
#define FMA1(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r.x[i]) : "v"(r.y[(i) & 7]), "s"(sg))
#define SUB1(i) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(r.y[(i) & 7]) : "v"(r.x[i]))
#define CVT1(i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "+v"(r.h[(i) & 3]) : "v"(r.y[(2 * (i)) & 7]), "v"(r.y[(2 * (i) + 1) & 7]))
#define MIXLO(i) asm volatile("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "+v"(r.h[2 + ((i) & 1)]) : "v"(r.h[(i) & 1]), "v"(r.y[(i) & 7]))
#define MIXHI(i) asm volatile("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r.h[2 + ((i) & 1)]) : "v"(r.h[(i) & 1]), "v"(r.y[((i) + 1) & 7]))
#define RD1(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r.rd[(i) & 7]) : "v"(lds_addr), "n"(((i) & 7) * 144) : "memory")
#define GL1(i) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r.g[(i) & 3]) : "v"(goff), "s"(gbase + ((i) & 3) * 1024) : "memory")
#define WAITLDS asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define WAITVM asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define NOP4 asm volatile("s_nop 0")
#define SALU1 asm volatile("s_add_u32 %0, %0, 1" : "+s"(sdummy))

template <int MODE, int j>
__device__ __forceinline__ void filler(Regs& r, float sg, unsigned lds_addr, unsigned goff, const char* gbase, unsigned& sdummy) {
    // j = gap 0..11 of the pattern
    if constexpr (MODE == 0) {
    } else if constexpr (MODE == 1) {  // 4 v_fma (sgpr)
        FMA1(0); FMA1(1); FMA1(2); FMA1(3);
    } else if constexpr (MODE == 2) {  // 4 sub + 2 cvt
        SUB1(0); SUB1(1); SUB1(2); SUB1(3); CVT1(0); CVT1(1);
    } else if constexpr (MODE == 3) {  // 4 mix
        MIXLO(0); MIXLO(1); MIXHI(0); MIXHI(1);
    } else if constexpr (MODE == 4) {  // 8 v_fma
        FMA1(0); FMA1(1); FMA1(2); FMA1(3); FMA1(4); FMA1(5); FMA1(6); FMA1(7);
    } else if constexpr (MODE == 5 || MODE == 7) {  // the kernel's pattern (7: without the reads)
        if (j < 2) {
            if (MODE == 5) WAITLDS;
            FMA1(0); FMA1(1); FMA1(2); FMA1(3); FMA1(4); FMA1(5); FMA1(6); FMA1(7);
        } else if (j < 4) {
            if (MODE == 5) { RD1(4 * (j - 2)); RD1(4 * (j - 2) + 1); RD1(4 * (j - 2) + 2); RD1(4 * (j - 2) + 3); }
        } else if ((j & 1) == 0) {
            SUB1(0); SUB1(1); SUB1(2); SUB1(3); CVT1(0); CVT1(1);
        } else {
            MIXLO(0); MIXLO(1); MIXHI(0); MIXHI(1);
        }
    } else if constexpr (MODE == 6 || MODE == 8) {  // balanced: the same 56 VALU + 8 reads, 4-5 VALU and at most one read per gap
        switch (j) {
            case 0: if (MODE == 6) WAITLDS; FMA1(0); FMA1(1); FMA1(2); FMA1(3); break;
            case 1: FMA1(4); FMA1(5); FMA1(6); FMA1(7); FMA1(8); if (MODE == 6) RD1(0); break;
            case 2: FMA1(9); FMA1(10); FMA1(11); FMA1(12); FMA1(13); if (MODE == 6) RD1(1); break;
            case 3: FMA1(14); FMA1(15); SUB1(0); SUB1(1); SUB1(2); if (MODE == 6) RD1(2); break;
            case 4: SUB1(3); CVT1(0); CVT1(1); MIXLO(0); MIXLO(1); if (MODE == 6) RD1(3); break;
            case 5: MIXHI(0); MIXHI(1); SUB1(0); SUB1(1); SUB1(2); if (MODE == 6) RD1(4); break;
            case 6: SUB1(3); CVT1(0); CVT1(1); MIXLO(0); MIXLO(1); if (MODE == 6) RD1(5); break;
            case 7: MIXHI(0); MIXHI(1); SUB1(0); SUB1(1); SUB1(2); if (MODE == 6) RD1(6); break;
            case 8: SUB1(3); CVT1(0); CVT1(1); MIXLO(0); MIXLO(1); if (MODE == 6) RD1(7); break;
            case 9: MIXHI(0); MIXHI(1); SUB1(0); SUB1(1); break;
            case 10: SUB1(2); SUB1(3); CVT1(0); CVT1(1); break;
            default: MIXLO(0); MIXLO(1); MIXHI(0); MIXHI(1); break;
        }
    } else if constexpr (MODE == 9) {  // 5 v_fma
        FMA1(0); FMA1(1); FMA1(2); FMA1(3); FMA1(4);
    } else if constexpr (MODE == 10) {  // 6 v_fma
        FMA1(0); FMA1(1); FMA1(2); FMA1(3); FMA1(4); FMA1(5);
    } else if constexpr (MODE == 11) {  // 4 v_fma + 1 ds_read_b128 in every gap
        FMA1(0); FMA1(1); FMA1(2); FMA1(3); RD1(j);
        if (j == 11) WAITLDS;
    } else if constexpr (MODE == 12) {  // 4 v_fma, and 4 reads in every fourth gap (the same reads per 12 gaps as mode 11... a third of them)
        FMA1(0); FMA1(1); FMA1(2); FMA1(3);
        if ((j & 3) == 0) { RD1(0); RD1(1); RD1(2); RD1(3); }
        if (j == 11) WAITLDS;
    } else if constexpr (MODE == 13) {  // 4 v_fma + one global load in 5 of 12 gaps (the kernel's 20 per 48)
        FMA1(0); FMA1(1); FMA1(2); FMA1(3);
        if (j == 1 || j == 3 || j == 5 || j == 7 || j == 9) GL1(j >> 1);
        if (j == 11) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    } else if constexpr (MODE == 14) {  // 4 v_fma, 4 global loads in one gap + 1 in another
        FMA1(0); FMA1(1); FMA1(2); FMA1(3);
        if (j == 5) { GL1(0); GL1(1); GL1(2); GL1(3); }
        if (j == 9) GL1(0);
        if (j == 11) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    } else if constexpr (MODE == 15) {  // 4 v_fma + 2 SALU + s_nop
        FMA1(0); FMA1(1); FMA1(2); FMA1(3); SALU1; SALU1; NOP4;
    } else if constexpr (MODE == 16) {  // 4 v_fma + 2 SALU
        FMA1(0); FMA1(1); FMA1(2); FMA1(3); SALU1; SALU1;
    } else if constexpr (MODE == 17) {  // 2 cvt_pk + 2 fma
        CVT1(0); CVT1(1); FMA1(0); FMA1(1);
    } else if constexpr (MODE >= 20 && MODE < 40) {
        // four (or N) of one instruction per gap, each on registers of its own
#define REP4(STMT) STMT(0); STMT(1); STMT(2); STMT(3)
#define I20(i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "+v"(r.h[i]) : "v"(r.y[2 * (i)]), "v"(r.y[2 * (i) + 1]))
#define I21(i) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "+v"(r.h[i]) : "v"(r.y[2 * (i)]), "v"(r.y[2 * (i) + 1]))
#define I22(i) asm volatile("v_fma_mix_f32 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "+v"(r.x[i]) : "v"(r.h[i]), "v"(r.y[i]))
#define I23(i) asm volatile("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "+v"(r.x[i]) : "v"(r.h[i]), "v"(r.y[i]))
#define I24(i) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(r.d[i]) : "v"(r.d[4 + (i)]))
#define I25(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(r.d[i]) : "v"(r.d[4 + (i)]), "v"(r.d[4 + ((i) ^ 1)]))
#define I26(i) asm volatile("v_cvt_f32_f16 %0, %1" : "+v"(r.x[i]) : "v"(r.h[i]))
#define I27(i) asm volatile("v_pk_add_f16 %0, %1, %2" : "+v"(r.h[i]) : "v"(r.x[i]), "v"(r.y[i]))
#define I28(i) asm volatile("v_cvt_f16_f32 %0, %1" : "+v"(r.x[i]) : "v"(r.y[i]))
#define I29(i) asm volatile("v_and_b32 %0, %1, %2" : "+v"(r.x[i]) : "v"(r.y[i]), "v"(r.y[4 + (i)]))
#define I30(i) asm volatile("v_perm_b32 %0, %1, %2, %3" : "+v"(r.x[i]) : "v"(r.y[i]), "v"(r.y[4 + (i)]), "s"(sdummy))
#define I31(i) asm volatile("v_pack_b32_f16 %0, %1, %2" : "+v"(r.x[i]) : "v"(r.y[i]), "v"(r.y[4 + (i)]))
#define I32(i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "+v"(r.d[i]) : "v"(lds_addr), "n"((i) * 144) : "memory")
#define I33(i) asm volatile("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r.x[i]) : "v"(r.h[i]), "v"(r.y[i]))
#define I34(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r.x[i]) : "v"(r.y[i]), "v"(r.y[4 + (i)]))
#define I35(i) asm volatile("v_sub_f32 %0, %1, %2" : "+v"(r.x[i]) : "v"(r.y[i]), "v"(r.y[4 + (i)]))
#define I36(i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(r.d[i]) : "v"(r.d[4 + (i)]))
        if constexpr (MODE == 20) { REP4(I20); }
        if constexpr (MODE == 21) { REP4(I21); }
        if constexpr (MODE == 22) { REP4(I22); }
        if constexpr (MODE == 23) { REP4(I23); }
        if constexpr (MODE == 24) { REP4(I24); }
        if constexpr (MODE == 25) { REP4(I25); }
        if constexpr (MODE == 26) { REP4(I26); }
        if constexpr (MODE == 27) { REP4(I27); }
        if constexpr (MODE == 28) { REP4(I28); }
        if constexpr (MODE == 29) { REP4(I29); }
        if constexpr (MODE == 30) { REP4(I30); }
        if constexpr (MODE == 31) { REP4(I31); }
        if constexpr (MODE == 32) { REP4(I32); if (j == 11) WAITLDS; }
        if constexpr (MODE == 33) { REP4(I33); }
        if constexpr (MODE == 34) { REP4(I34); }
        if constexpr (MODE == 35) { REP4(I35); }
        if constexpr (MODE == 36) { REP4(I36); }
        if constexpr (MODE == 37) { I35(0); I35(1); I35(2); I35(3); I35(4); I35(5); I35(6); I35(7); }   // 8 independent v_sub
        if constexpr (MODE == 38) { I20(0); I20(1); I20(2); I20(3); I22(0); I22(1); I22(2); I22(3); }   // 4 cvt_pk + 4 mix_f32
        if constexpr (MODE == 39) { RD1(0); RD1(1); if (j == 11) WAITLDS; }                             // 2 ds_read_b128, nothing else
    } else if constexpr (MODE >= 40 && MODE < 50) {
        // the kernel's pattern with the lo half as v_fma_mix_f32 + v_cvt_pk_f16_f32, and its 5 global loads per 12 gaps
        // (16 ring + 4 chunk loads per 48): 40 none, 41 a burst of four + one, 42 one per gap in five light gaps, 43 one per gap in gaps 5-9
        if (j < 2) {
            WAITLDS;
            FMA1(0); FMA1(1); FMA1(2); FMA1(3); FMA1(4); FMA1(5); FMA1(6); FMA1(7);
        } else if (j < 4) {
            RD1(4 * (j - 2)); RD1(4 * (j - 2) + 1); RD1(4 * (j - 2) + 2); RD1(4 * (j - 2) + 3);
        } else if ((j & 1) == 0) {
            SUB1(0); SUB1(1); SUB1(2); SUB1(3); CVT1(0); CVT1(1);
        } else {
            I22(0); I22(1); I22(2); I22(3); I20(2); I20(3);
        }
        if constexpr (MODE == 41) {
            if (j == 5) { GL1(0); GL1(1); GL1(2); GL1(3); }
            if (j == 9) GL1(0);
        }
        if constexpr (MODE == 42) {
            if (j == 2 || j == 3 || j == 5 || j == 7 || j == 9) GL1(j >> 1);
        }
        if constexpr (MODE == 43) {
            if (j >= 5 && j <= 9) GL1(j);
        }
        if constexpr (MODE == 44) {  // two bursts of two + one
            if (j == 5 || j == 9) { GL1(0); GL1(1); }
            if (j == 11) GL1(2);
        }
        // (two rounds of loads stay in flight, as in the kernel's ring: the wait is for loads issued 24-36 gaps ago)
        if (MODE != 40 && j == 11) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    } else if constexpr (MODE == 18) {  // 6 plain v_add (the guide's filler)
        asm volatile("v_add_f32 %0, %1, %0" : "+v"(r.x[0]) : "v"(r.y[0]));
        asm volatile("v_add_f32 %0, %1, %0" : "+v"(r.x[1]) : "v"(r.y[1]));
        asm volatile("v_add_f32 %0, %1, %0" : "+v"(r.x[2]) : "v"(r.y[2]));
        asm volatile("v_add_f32 %0, %1, %0" : "+v"(r.x[3]) : "v"(r.y[3]));
        asm volatile("v_add_f32 %0, %1, %0" : "+v"(r.x[4]) : "v"(r.y[4]));
        asm volatile("v_add_f32 %0, %1, %0" : "+v"(r.x[5]) : "v"(r.y[5]));
    }
}

constexpr int ITERS = 64;  // x 12 gaps

template <int MODE>
__global__ void __launch_bounds__(256, 1) probe(const char* gsrc, float* sink, unsigned long long* cycles, float sgv) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x, lane = tid & 63, q = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 24064 * 2 / 16; i += 256) reinterpret_cast<f4*>(smem)[i] = f4{1.f, 2.f, 3.f, 4.f};
    __syncthreads();
    Regs r;
    for (int i = 0; i < 16; i++) r.x[i] = (float)(lane + i);
    for (int i = 0; i < 8; i++) r.y[i] = (float)(lane * 3 + i);
    for (int i = 0; i < 4; i++) r.h[i] = lane + i;
    for (int i = 0; i < 8; i++) r.rd[i] = f4{0, 0, 0, 0};
    for (int i = 0; i < 4; i++) r.g[i] = u4{0, 0, 0, 0};
    for (int i = 0; i < 8; i++) r.d[i] = (double)(lane + i);
    const float sg = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(sgv)));
    // the kernel's read geometry: lane = (tile n, k-half): 32 different tiles, pitch 144 B, second board 16 B to the right
    const int n = lane & 31, hh = lane >> 5;
    const unsigned lds_addr = (unsigned)(((n >> 4) * 8 + 2 * ((n >> 2) & 3)) * (8 * 144 + 64) + 2 * (n & 3) * 144 + (n >> 4) * 16 + hh * 32 + q * (8 * 144 + 64));
    const unsigned goff = lane * 16;
    const char* gbase = gsrc + ((size_t)(blockIdx.x & 3) * 4 + q) * 4096 * 64;  // 4 MiB that every XCD's L2 holds after the first pass
    unsigned sdummy = 0;
    h8 a, b;
    for (int i = 0; i < 8; i++) a[i] = (_Float16)(lane & 3), b[i] = (_Float16)(i & 1);
    f16v acc[16];
    for (int i = 0; i < 16; i++)
        for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it++) {
#define GAP(j)                                                                                                                     \
    acc[(2 * (j / 3) + (j & 1)) & 15] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[(2 * (j / 3) + (j & 1)) & 15], 0, 0, 0); \
    SB;                                                                                                                            \
    filler<MODE, j>(r, sg, lds_addr, goff, gbase + (size_t)(it & 63) * 4096, sdummy);                                              \
    SB;
        GAP(0) GAP(1) GAP(2) GAP(3) GAP(4) GAP(5) GAP(6) GAP(7) GAP(8) GAP(9) GAP(10) GAP(11)
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 16; i++) s += acc[i][0] + r.x[i];
    for (int i = 0; i < 8; i++) s += r.y[i] + r.rd[i][0] + r.rd[i][3];
    for (int i = 0; i < 4; i++) s += (float)r.h[i] + (float)r.g[i][0] + (float)r.g[i][3];
    for (int i = 0; i < 8; i++) s += (float)r.d[i];
    sink[blockIdx.x * 256 + tid] = s + (float)sdummy;
    if (lane == 0) cycles[blockIdx.x * 4 + q] = t1 - t0;
}

// What a CU takes in from a set its XCD's L2 holds, by how many CUs ask at once: four waves per CU, each streaming its own 256 KiB
// (global_load_dwordx4, 1 KiB per instruction, 8 in flight per wave), `blocks` workgroups (one per CU while blocks <= 256).
__global__ void __launch_bounds__(256, 1) stream_probe(const char* gsrc, float* sink, unsigned long long* cycles, int rounds, int shared) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x, lane = tid & 63, q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned goff = lane * 16;
    // shared = 1: every CU reads the same 1 MiB (4 waves x 256 KiB); 0: its own 1 MiB of a 4 MiB set (the blocks of one cout group share)
    const char* gbase = gsrc + ((size_t)(shared ? 0 : (blockIdx.x & 3)) * 4 + q) * 262144;
    u4 r[8];
    for (int i = 0; i < 8; i++) r[i] = u4{0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < rounds; it++) {
        const char* p = gbase + (size_t)(it & 31) * 8192;
#pragma unroll
        for (int i = 0; i < 8; i++) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r[i]) : "v"(goff), "s"(p + i * 1024) : "memory");
        asm volatile("s_waitcnt vmcnt(4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]));
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; i++) s += (float)r[i][0] + (float)r[i][3];
    sink[blockIdx.x * 256 + tid] = s;
    if (lane == 0) cycles[blockIdx.x * 4 + q] = t1 - t0;
}

static void run_stream(const char* gsrc, float* sink, unsigned long long* cyc, int blocks, int shared) {
    const int rounds = 512;
    std::vector<unsigned long long> h(blocks * 4);
    double best = 1e30;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(stream_probe, dim3(blocks), dim3(256), 98304, 0, gsrc, sink, cyc, rounds, shared);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        best = std::min(best, (double)h[h.size() / 2]);
    }
    std::printf("stream: %3d workgroups, %s: %6.1f B/clk per CU (4 waves x %d KiB in %.0f cycles)\n", blocks, shared ? "one 1 MiB set for all" : "1 MiB of a 4 MiB set ",
                4.0 * rounds * 8192 / best, rounds * 8, best);
}

template <int MODE>
static void run(const char* name, const char* gsrc, float* sink, unsigned long long* cyc, int blocks) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    std::vector<unsigned long long> h(blocks * 4);
    double best = 1e30;
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 98304, 0, gsrc, sink, cyc, -1.0f);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        best = std::min(best, (double)h[h.size() / 2]);
    }
    std::printf("mode %2d  %-62s %7.2f cycles per gap (median wave, best of 3)\n", MODE, name, best / (ITERS * 12));
}

int main() {
    const int blocks = 256;
    char* gsrc;
    float* sink;
    unsigned long long* cyc;
    hipMalloc(&gsrc, (size_t)blocks * 4 * 4096 * 64 + 8192);
    hipMemset(gsrc, 0, (size_t)blocks * 4 * 4096 * 64 + 8192);
    hipMalloc(&sink, blocks * 256 * 4);
    hipMalloc(&cyc, blocks * 4 * 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_probe), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    for (int shared = 0; shared < 2; shared++)
        for (int b : {256, 128, 64, 32, 8}) run_stream(gsrc, sink, cyc, b, shared);
    run<0>("bare MFMAs", gsrc, sink, cyc, blocks);
    run<18>("6 v_add_f32", gsrc, sink, cyc, blocks);
    run<1>("4 v_fma_f32 (SGPR operand)", gsrc, sink, cyc, blocks);
    run<9>("5 v_fma_f32", gsrc, sink, cyc, blocks);
    run<10>("6 v_fma_f32", gsrc, sink, cyc, blocks);
    run<4>("8 v_fma_f32", gsrc, sink, cyc, blocks);
    run<2>("4 v_sub_f32 + 2 v_cvt_pk_f16_f32", gsrc, sink, cyc, blocks);
    run<17>("2 v_cvt_pk_f16_f32 + 2 v_fma_f32", gsrc, sink, cyc, blocks);
    run<3>("2 v_fma_mixlo_f16 + 2 v_fma_mixhi_f16", gsrc, sink, cyc, blocks);
    run<16>("4 v_fma_f32 + 2 s_add_u32", gsrc, sink, cyc, blocks);
    run<15>("4 v_fma_f32 + 2 s_add_u32 + s_nop", gsrc, sink, cyc, blocks);
    run<11>("4 v_fma_f32 + 1 ds_read_b128 per gap", gsrc, sink, cyc, blocks);
    run<12>("4 v_fma_f32, 4 ds_read_b128 in every fourth gap", gsrc, sink, cyc, blocks);
    run<13>("4 v_fma_f32 + 1 global_load_dwordx4 in 5 of 12 gaps", gsrc, sink, cyc, blocks);
    run<14>("4 v_fma_f32, 4 global loads in one gap + 1 in another", gsrc, sink, cyc, blocks);
    run<35>("4 v_sub_f32 (e64, three registers)", gsrc, sink, cyc, blocks);
    run<37>("8 v_sub_f32", gsrc, sink, cyc, blocks);
    run<34>("4 v_fma_f32 (VGPR operands)", gsrc, sink, cyc, blocks);
    run<20>("4 v_cvt_pk_f16_f32", gsrc, sink, cyc, blocks);
    run<21>("4 v_cvt_pkrtz_f16_f32", gsrc, sink, cyc, blocks);
    run<28>("4 v_cvt_f16_f32", gsrc, sink, cyc, blocks);
    run<26>("4 v_cvt_f32_f16", gsrc, sink, cyc, blocks);
    run<22>("4 v_fma_mix_f32 (f16 x const + f32)", gsrc, sink, cyc, blocks);
    run<23>("4 v_fma_mixlo_f16, independent registers", gsrc, sink, cyc, blocks);
    run<33>("4 v_fma_mixhi_f16, independent registers", gsrc, sink, cyc, blocks);
    run<38>("4 v_cvt_pk_f16_f32 + 4 v_fma_mix_f32", gsrc, sink, cyc, blocks);
    run<24>("4 v_pk_add_f32", gsrc, sink, cyc, blocks);
    run<36>("4 v_pk_mul_f32", gsrc, sink, cyc, blocks);
    run<25>("4 v_pk_fma_f32", gsrc, sink, cyc, blocks);
    run<27>("4 v_pk_add_f16", gsrc, sink, cyc, blocks);
    run<29>("4 v_and_b32", gsrc, sink, cyc, blocks);
    run<30>("4 v_perm_b32", gsrc, sink, cyc, blocks);
    run<31>("4 v_pack_b32_f16", gsrc, sink, cyc, blocks);
    run<32>("4 ds_read_b64", gsrc, sink, cyc, blocks);
    run<39>("2 ds_read_b128", gsrc, sink, cyc, blocks);
    run<40>("new pattern (8,8,4r,4r,6,6,...), no global loads", gsrc, sink, cyc, blocks);
    run<41>("new pattern + 5 global loads per 12 gaps: 4 in one gap, 1 in another", gsrc, sink, cyc, blocks);
    run<44>("new pattern + 5 global loads: 2 + 2 + 1", gsrc, sink, cyc, blocks);
    run<42>("new pattern + 5 global loads: one per gap (gaps 2,3,5,7,9)", gsrc, sink, cyc, blocks);
    run<43>("new pattern + 5 global loads: one per gap (gaps 5-9)", gsrc, sink, cyc, blocks);
    run<7>("the kernel's pattern without its reads (8,8,0,0,6,4,6,4,6,4,6,4)", gsrc, sink, cyc, blocks);
    run<5>("the kernel's pattern (8,8,4r,4r,6,4,6,4,6,4,6,4)", gsrc, sink, cyc, blocks);
    run<8>("balanced pattern without reads (4,5x8,4,4,4)", gsrc, sink, cyc, blocks);
    run<6>("balanced pattern, one read per gap in gaps 1-8", gsrc, sink, cyc, blocks);
    return 0;
}
