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

// bf16x3 operands (GRU scans, GRU weight gradients).  An fp32 value splits EXACTLY into three bf16 parts
// by truncation (8 + 8 + 8 significant bits: hi = the upper half of its word, the remainders are exact differences); the
// product a*b is then accumulated from the six part products whose weight is above 2^-24 of |a||b| on the bf16 MFMA
// (16x16x32: the wave's whole K = 32 slice in one instruction, ~17 clocks, against eight 32-clock fp32 MFMAs) with fp32
// accumulation, smallest terms first - fp32-class accuracy (the three dropped products are below the fp32 rounding of the sum).
typedef __bf16 gbf16x8 __attribute__((ext_vector_type(8)));
struct Bf3 {
    u32x4_t hi, mid, lo;          // 8 values each, packed in the order of the poll's words: (n = 0: x y z w), (n = 1: x y z w)
};
__device__ __forceinline__ void split3_pair(float v0, float v1, unsigned& hi, unsigned& mid, unsigned& lo) {
    const unsigned u0 = __float_as_uint(v0), u1 = __float_as_uint(v1);
    hi = __builtin_amdgcn_perm(u1, u0, 0x07060302u);                     // (u1 & 0xffff0000) | (u0 >> 16)
    const float r0 = v0 - __uint_as_float(u0 & 0xffff0000u), r1 = v1 - __uint_as_float(u1 & 0xffff0000u);
    const unsigned m0 = __float_as_uint(r0), m1 = __float_as_uint(r1);
    mid = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
    const float l0 = r0 - __uint_as_float(m0 & 0xffff0000u), l1 = r1 - __uint_as_float(m1 & 0xffff0000u);
    lo = __builtin_amdgcn_perm(__float_as_uint(l1), __float_as_uint(l0), 0x07060302u);
}
__device__ __forceinline__ Bf3 split3x8(float4 a, float4 b) {
    unsigned h[4], m[4], l[4];
    split3_pair(a.x, a.y, h[0], m[0], l[0]);
    split3_pair(a.z, a.w, h[1], m[1], l[1]);
    split3_pair(b.x, b.y, h[2], m[2], l[2]);
    split3_pair(b.z, b.w, h[3], m[3], l[3]);
    return Bf3{u32x4_t{h[0], h[1], h[2], h[3]}, u32x4_t{m[0], m[1], m[2], m[3]}, u32x4_t{l[0], l[1], l[2], l[3]}};
}
typedef float pbsed_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 pbsed_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16_rne(float lo, float hi) {          // v_cvt_pk_bf16_f32
    return __builtin_bit_cast(unsigned, __builtin_convertvector(pbsed_f32x2{lo, hi}, pbsed_bf16x2));
}
__device__ __forceinline__ f32x4 mfma_b16(u32x4_t a, u32x4_t b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(gbf16x8, a), __builtin_bit_cast(gbf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_x3(const Bf3& a, const Bf3& b, f32x4 c);
// Operand of a bf16-MFMA product in one of two formats: XS = 3 the exact three-way split (fp32-class), XS = 1 one part
// rounded to nearest even (plain bf16 operands, the bf16 training mode); XS = 0: placeholder of the fp32-MFMA code paths.
template <int XS> struct BfOp {};
template <> struct BfOp<3> { Bf3 v; };
template <> struct BfOp<1> { u32x4_t v; };
template <int XS> __device__ __forceinline__ BfOp<XS> make_op(float4 a, float4 b) {
    BfOp<XS> r;
    if constexpr (XS == 3) r.v = split3x8(a, b);
    if constexpr (XS == 1) r.v = u32x4_t{pack_bf16_rne(a.x, a.y), pack_bf16_rne(a.z, a.w), pack_bf16_rne(b.x, b.y), pack_bf16_rne(b.z, b.w)};
    return r;
}
template <int XS> __device__ __forceinline__ f32x4 mfma_op(const BfOp<XS>& a, const BfOp<XS>& b, f32x4 c) {
    if constexpr (XS == 3) return mfma_x3(a.v, b.v, c);
    else if constexpr (XS == 1) return mfma_b16(a.v, b.v, c);
    else return c;
}
__device__ __forceinline__ f32x4 mfma_x3(const Bf3& a, const Bf3& b, f32x4 c) {
    c = mfma_b16(a.lo, b.hi, c);
    c = mfma_b16(a.hi, b.lo, c);
    c = mfma_b16(a.mid, b.mid, c);
    c = mfma_b16(a.mid, b.hi, c);
    c = mfma_b16(a.hi, b.mid, c);
    return mfma_b16(a.hi, b.hi, c);
}


void set_error(const char* fmt, ...);
int check_launch(const char* what);
int device_cus();                                             // CUs of the CURRENT device (cached per device ordinal)
int launch_cus();                                             // ... or the caller's budget for grid sizing (pbsed_set_launch_cus), api.hip
float* scratch_for(hipStream_t stream, size_t floats);        // partial-sum scratch of (current device, stream); api.hip
float* scratch_zeroed_front(hipStream_t stream, size_t front, size_t total);     // ... whose first `front` floats are zero and stay so
constexpr size_t PBSED_SCRATCH_FRONT = (size_t)32 * (16384 + 64);               // the slotted conv weight gradients' part of it

// runtime calls in front of a launch (attributes, memsets): report a failure like a failed launch
#define PBSED_HIP_TRY(expr, what)                                               \
    do {                                                                        \
        const hipError_t pbsed_e_ = (expr);                                     \
        if (pbsed_e_ != hipSuccess) {                                           \
            set_error("%s: %s", what, hipGetErrorString(pbsed_e_));             \
            return PBSED_E_HIP;                                                 \
        }                                                                       \
    } while (0)

// Kernel setup (opt-in to > 48 KB of dynamic LDS) is per DEVICE and per SIZE: a call site remembers, per device ordinal,
// the largest size it has set, and re-issues the attribute when a later launch needs more (the front-end kernels' LDS
// grows with the number of filters / bins), so a process driving several GPUs or several configurations stays correct.
#define PBSED_DYN_LDS_ONCE(kern, bytes)                                                                         \
    do {                                                                                                        \
        static unsigned pbsed_set_[64] = {0};                                                                   \
        int pbsed_dev_ = 0;                                                                                     \
        PBSED_HIP_TRY(hipGetDevice(&pbsed_dev_), "hipGetDevice");                                                \
        unsigned* pbsed_slot_ = &pbsed_set_[pbsed_dev_ & 63];                                                   \
        const unsigned pbsed_need_ = (unsigned)(bytes);                                                         \
        if (pbsed_need_ > 48u * 1024u && __atomic_load_n(pbsed_slot_, __ATOMIC_ACQUIRE) < pbsed_need_) {        \
            PBSED_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                              \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)pbsed_need_),    \
                          "hipFuncSetAttribute");                                                               \
            unsigned pbsed_old_ = __atomic_load_n(pbsed_slot_, __ATOMIC_ACQUIRE);                               \
            while (pbsed_old_ < pbsed_need_ &&                                                                  \
                   !__atomic_compare_exchange_n(pbsed_slot_, &pbsed_old_, pbsed_need_, false, __ATOMIC_RELEASE, \
                                                __ATOMIC_ACQUIRE)) {}                                           \
        }                                                                                                       \
    } while (0)

}  // namespace pbsed
