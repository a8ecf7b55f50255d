// Shared device helpers for the pb_sed MI355X (gfx950) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));   // raw_buffer_load_b128 result

#define PBSED_OK 0
#define PBSED_E_ARG (-1)
#define PBSED_E_HIP (-2)
#define PBSED_E_UNSUPPORTED (-3)
#define PBSED_STAT_SLOTS 32   // statistics accumulators are [PBSED_STAT_SLOTS][C][2] doubles

namespace pbsed {

// v_mfma_f32_16x16x4_f32: A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15],
// D[row = (lane>>4)*4 + reg][col = lane&15].  Exact f32 (k-ordered fma chain).
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <int CTRL>
__device__ __forceinline__ float dpp_shuffled(float v) {
    // lane permutation inside a 16-lane DPP row as an operand modifier (folds into v_add_f32_dpp): no LDS crossbar trip
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

__device__ __forceinline__ float wave_sum16(float v) {
    // sum over the 16 lanes sharing lane>>4 (one DPP row): pairs, quads, then the two mirror steps; every lane ends
    // with the total
    v += dpp_shuffled<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_shuffled<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_shuffled<0x141>(v);     // row_half_mirror
    v += dpp_shuffled<0x140>(v);     // row_mirror
    return v;
}

__device__ __forceinline__ float wave_sum64(float v) {
    v = wave_sum16(v);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

__device__ __forceinline__ double wave_sum64d(double v) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

void set_error(const char* fmt, ...);
int check_launch(const char* what);

// runtime calls in front of a launch (attributes, memsets): report a failure like a failed launch
#define PBSED_HIP_TRY(expr, what)                                               \
    do {                                                                        \
        const hipError_t pbsed_e_ = (expr);                                     \
        if (pbsed_e_ != hipSuccess) {                                           \
            set_error("%s: %s", what, hipGetErrorString(pbsed_e_));             \
            return PBSED_E_HIP;                                                 \
        }                                                                       \
    } while (0)

// One-time kernel setup (opt-in to > 48 KB of dynamic LDS) is per DEVICE, not per process: a bit per device ordinal
// in a per-call-site mask, so a process driving several GPUs through the C-ABI sets the attribute on each of them.
#define PBSED_DYN_LDS_ONCE(kern, bytes)                                                                         \
    do {                                                                                                        \
        static unsigned long long pbsed_mask_ = 0ull;                                                           \
        int pbsed_dev_ = 0;                                                                                     \
        PBSED_HIP_TRY(hipGetDevice(&pbsed_dev_), "hipGetDevice");                                                \
        const unsigned long long pbsed_bit_ = 1ull << (pbsed_dev_ & 63);                                        \
        if (!(__atomic_load_n(&pbsed_mask_, __ATOMIC_ACQUIRE) & pbsed_bit_)) {                                  \
            if ((size_t)(bytes) > 48 * 1024)                                                                    \
                PBSED_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                          \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)),    \
                              "hipFuncSetAttribute");                                                           \
            __atomic_fetch_or(&pbsed_mask_, pbsed_bit_, __ATOMIC_RELEASE);                                      \
        }                                                                                                       \
    } while (0)

}  // namespace pbsed
