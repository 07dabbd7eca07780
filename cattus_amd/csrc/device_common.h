// Device-side definitions shared by the kernel translation units (kernels.hip, kernels_t64s.hip): vector types, the MFMA
// wrappers, the fragment-order index, the split tower's LDS pitch and weight-stage constants, small helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace cattus {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <typename T>
struct Mfma;

template <>
struct Mfma<__bf16> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ void mac(const frag& a, const frag& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};

// f16 operands (the split-precision tower: DESIGN.md section 3, K1s): same lane map and cycles as the bf16 form
template <>
struct Mfma<_Float16> {
    typedef f16x8 frag;
    static __device__ __forceinline__ void mac(const frag& a, const frag& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

template <>
struct Mfma<float> {
    typedef f32x4 frag;
    // lane half h holds k = 4h + j in element j: the chain visits k = 0,4,1,5,2,6,3,7 of the 8-group
    static __device__ __forceinline__ void mac(const frag& a, const frag& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], c, 0, 0, 0);
    }
};

// Element index of (row, k) of a [rows][K] matrix kept in MFMA fragment order (kernels.h, HeadsMfma).
template <typename T>
__host__ __device__ constexpr size_t frag_packed_index(uint32_t row, uint32_t k, uint32_t K) {
    constexpr uint32_t KSTEP = 32 / sizeof(T), HALF = KSTEP / 2;
    return ((((size_t)(row >> 5) * (K / KSTEP) + k / KSTEP) * 2 + (k % KSTEP) / HALF) * 32 + (row & 31)) * HALF + k % HALF;
}

// f32 rows that feed a Winograd layer: capped at WINO_ACT_MAX (kernels.h), values beyond it counted like note_saturation's.
__device__ __forceinline__ void cap_wino_input(float (&y)[8], bool valid, unsigned* sat) {
    const float m = fmaxf(fmaxf(fmaxf(y[0], y[1]), fmaxf(y[2], y[3])), fmaxf(fmaxf(y[4], y[5]), fmaxf(y[6], y[7])));
    if (valid && m > WINO_ACT_MAX) {
        unsigned n = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) n += y[j] > WINO_ACT_MAX ? 1u : 0u;
        atomicAdd(sat, n);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) y[j] = y[j] < WINO_ACT_MAX ? y[j] : WINO_ACT_MAX;
}

// Off-board taps of the 3x3 kernels read zeros.  One zero ROW behind an image's rows put every such lane of a ds_read_b128 group on
// one bank slot, colliding with whichever lane's real row shares it (a quarter of the conv kernels' LDS cycles were bank conflicts).
// A zero AREA instead, 256-B aligned and read at (the off-board row's own address mod 256), gives the lane the banks its row would
// have had -- the group's addresses stay linear in the pixel index, as conflict-free as on a board without borders.  ZAREA_SP: for
// images of 144-byte rows read as (hi at +0, lo at +64): 255 + 64 + 16 bytes rounded up; 128-byte swizzled rows need 256.
constexpr int ZAREA_SP = 352;

// The f16 towers store activations as f16 and clamp them at 65504 instead of letting them overflow to infinity.  A clamped
// value is a wrong value: it is counted (atomic add on the clamp path only -- a network inside the f16 range never gets
// here) into the evaluator's sticky counter, which cattus_hip_stats reports as `saturated`.
__device__ __forceinline__ void note_saturation(const float (&y)[8], bool valid, unsigned* sat) {
    const float m = fmaxf(fmaxf(fmaxf(y[0], y[1]), fmaxf(y[2], y[3])), fmaxf(fmaxf(y[4], y[5]), fmaxf(y[6], y[7])));
    if (valid && m > 65504.0f) {
        unsigned n = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) n += y[j] > 65504.0f ? 1u : 0u;
        atomicAdd(sat, n);
    }
}

__device__ __forceinline__ void glds16(const char* src, char* lds_dst) {
    __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(lds_dst), 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vm_barrier() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit count");
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// ---- split-precision tower (K1s, K1rs): LDS image and weight stages ----
constexpr int SP = 144;                                 // LDS row pitch
constexpr int SW_D = 6;                       // weight stages in flight per consumer wave; divides the 18 stages of a chunk
constexpr int SW_STAGE = 2048;                // bytes per 32-cout block and stage: hi fragment, lo fragment
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

}  // namespace cattus
