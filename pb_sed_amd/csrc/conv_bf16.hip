// Implicit-GEMM convolution forward / data-gradient on bf16 MFMA (v_mfma_f32_16x16x32_bf16) for gfx950.
//
// Same mapping, prologue and epilogue as conv.hip (M = Cout, N = 16 consecutive t, K = (tap, cin); BN-apply +
// ReLU + seq-mask while staging; bias / pool / statistics / BN-backward epilogue shared through
// conv_epilogue.h); what changes is the operand format: activations and weights are converted to bf16 while
// they are staged into LDS (activations stay fp32 in HBM, accumulation is fp32).
//   NSPLIT = 1: plain bf16 operands - the compute dtype of BASELINE.json config 3 (BiCRNN bf16).
//   NSPLIT = 3: every fp32 operand is split exactly into three bf16 terms (8+8+8 mantissa bits); the six
//               leading partial products are accumulated in fp32, which reproduces an fp32 product to
//               ~2^-23 relative at 6/16 of the fp32-MFMA cost (opt-in).
// Channels-innermost LDS images ([row][t][32 ch + 8 pad] bf16, 80-byte position stride) give each lane its
// 8 consecutive-k operand values as one aligned ds_read_b128.  Weights are pre-split/packed by
// pack_conv_weights_bf16 to [split][tap][Cout_pad][Cin_pad] and staged one kernel row (kh) at a time.
#include <cstdlib>

#include "common.h"
#include "conv_epilogue.h"
#include "pbsed_internal.h"

namespace pbsed {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short us8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short f2bf(float x) {          // round to nearest even
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

template <int NSPLIT>
__device__ __forceinline__ void split_bf16(float x, unsigned short (&o)[NSPLIT]) {
    o[0] = f2bf(x);
    if (NSPLIT > 1) {
        float r = x - bf2f(o[0]);          // exact
        o[1] = f2bf(r);
        r -= bf2f(o[1]);                   // exact
        o[2] = f2bf(r);
    }
}

__device__ __forceinline__ f32x4 mfma_bf16(us8 a, us8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

constexpr int CB_COUT_T = 64, CB_CK = 32, CB_CKP = 40;

template <int FT, int TT, int KH, int KW, int NSPLIT, bool POOL>
struct ConvBCfg {
    static constexpr int WM = 2, WN = 2, MTW = CB_COUT_T / 16 / WM;
    static constexpr int TT16 = TT / 16, NTT = TT16 / WN, NTW = FT * NTT;
    static constexpr int KK = KH * KW, ROWS = FT + KH - 1;
    static constexpr int HALO = (KW > 1) ? 4 : 0, ROW = TT + 2 * HALO;
    static constexpr int IN_ITEMS = ROWS * ROW * (CB_CK / 8), IN_PER_T = (IN_ITEMS + 255) / 256;
    static constexpr int W_ITEMS = KW * CB_COUT_T * (CB_CK / 8), W_PER_T = (W_ITEMS + 255) / 256;
    static constexpr int IN_HALFS = NSPLIT * ROWS * ROW * CB_CKP, W_HALFS = NSPLIT * KW * CB_COUT_T * CB_CKP;
    static constexpr int FO_T = POOL ? FT / 2 : FT;
    static constexpr size_t LDS_BYTES = (size_t)(IN_HALFS + W_HALFS) * 2 + WN * CB_COUT_T * FO_T * 2 * sizeof(float);
    static_assert(TT16 % WN == 0 && (IN_HALFS % 8) == 0 && (W_HALFS % 8) == 0, "tile granularity");
};



// ---------------------------------------------------------------------------------------------------------------
// Software-pipelined plain-bf16 kernel (NSPLIT = 1; the compute dtype of BASELINE.json config 3).  Same tile, LDS
// image, MFMA loop and epilogue as conv_bf16_kernel; what differs is how the operands get there:
//  * a staging item is (8 channels, 1 row, 4 consecutive t): eight 16-byte raw buffer loads on a clip-relative
//    resource with 32-bit offsets (out-of-range rows / columns / padded channels read 0 without branches), advanced
//    by a uniform step per chunk; the loads of chunk c+1 are in flight during the MFMAs of chunk c;
//  * BN-apply + ReLU + sequence mask run on the registers, v_cvt_pk_bf16_f32 packs pairs, and the 8x4 register tile
//    goes to the channels-innermost LDS image as four 16-byte writes (the transposition costs no shuffles);
//  * BN scale / shift of all input channels sit in LDS for the life of the block;
//  * the next kernel row's weights are prefetched into registers during the MFMAs of the current one.
// ---------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t));
}

// NS = 1: plain bf16 operands; NS = 3: exact three-way splits (Bf3 in common.h: the input is split while it is staged, the
// weights arrive split from the pack), six part products per product - fp32-class results on the bf16 MFMA.
template <int NS, int FT, int TT, int KH, int KW, bool POOL>
struct ConvB2Cfg : ConvBCfg<FT, TT, KH, KW, NS, POOL> {
    using B = ConvBCfg<FT, TT, KH, KW, NS, POOL>;
    static constexpr int IN_PART = B::IN_HALFS / NS, W_PART = B::W_HALFS / NS;
    static constexpr int QR = B::ROW / 4;
    static constexpr int IN_ITEMS2 = (CB_CK / 8) * B::ROWS * QR, IN_PER_T2 = (IN_ITEMS2 + 255) / 256;
    static constexpr int W_ITEMS2 = KW * CB_COUT_T * (CB_CK / 8), W_PER_T2 = (W_ITEMS2 + 255) / 256;
};

template <int NS, int FT, int TT, int KH, int KW, bool POOL, bool DGRAD>
__global__ __launch_bounds__(256) void conv_bf16v2_kernel(ConvFwdArgs a, const unsigned short* __restrict__ wpb) {
    using C = ConvB2Cfg<NS, FT, TT, KH, KW, POOL>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned short* in_s = reinterpret_cast<unsigned short*>(smem_raw);            // [ROWS][ROW][CKP]
    unsigned short* w_s = in_s + C::IN_HALFS;                                       // [KW][64][CKP]
    float* st_s = reinterpret_cast<float*>(w_s + C::W_HALFS);                       // [WN][64][FO_T][2]
    float* sc_s = st_s + C::WN * CB_COUT_T * C::FO_T * 2;                           // [CinP] BN scale, [CinP] shift

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WN, wn = wave % C::WN;
    const int lq = lane >> 4, lr = lane & 15;
    const int nTt = (a.T + TT - 1) / TT, nFt = (a.F + FT - 1) / FT;
    int bx = blockIdx.x;
    const int t0 = (bx % nTt) * TT; bx /= nTt;
    const int f0 = (bx % nFt) * FT;
    const int b = bx / nFt;
    const int cout0 = blockIdx.y * CB_COUT_T;
    const int sl = a.seq_len ? min(a.seq_len[b], a.T) : a.T;
    const bool pro = a.scale != nullptr;
    constexpr int PADH = (KH - 1) / 2, PADW = (KW - 1) / 2;
    const bool unpool = DGRAD && a.unpool_idx != nullptr;
    const int Fsrc = unpool ? a.F / 2 : a.F;
    const int tlim = pro ? sl : a.T;
    const bool vec = (a.T & 3) == 0;
    float* sh_s = sc_s + a.CinP;
    if (pro) {
        for (int c = tid; c < a.CinP; c += 256) {
            sc_s[c] = c < a.Cin ? a.scale[c] : 0.f;
            sh_s[c] = c < a.Cin ? a.shift[c] : 0.f;
        }
    }

    f32x4 acc[C::MTW][C::NTW];
#pragma unroll
    for (int m = 0; m < C::MTW; ++m)
#pragma unroll
        for (int n = 0; n < C::NTW; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr unsigned OOB = 0x80000000u;
    const unsigned clip_elems = (unsigned)(a.Cin * Fsrc * a.T);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x) + (size_t)b * clip_elems, 0, clip_elems * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc(
        unpool ? const_cast<uint8_t*>(a.unpool_idx) + (size_t)b * clip_elems : nullptr, 0, unpool ? clip_elems : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned short*>(wpb), 0, (unsigned)(NS * C::KK * a.CinP * a.CoutP) * 2u, 0x00020000);
    const unsigned cstride = (unsigned)(Fsrc * a.T) * 4u;           // bytes between channels of one clip
    const unsigned step_x = cstride * CB_CK;

    u32x4_t rin[C::IN_PER_T2][8];
    unsigned ridx[C::IN_PER_T2][8];
    unsigned q_voff[C::IN_PER_T2];
    int q_lds[C::IN_PER_T2], q_m[C::IN_PER_T2], q_c[C::IN_PER_T2];
#pragma unroll
    for (int i = 0; i < C::IN_PER_T2; ++i) {
        const int it = tid + i * 256;
        const int oc = it / (C::ROWS * C::QR), rem = it - oc * (C::ROWS * C::QR);
        const int r = rem / C::QR, qc = rem - r * C::QR;
        const int f = f0 - PADH + r, tq = t0 - C::HALO + 4 * qc;
        const bool ok = it < C::IN_ITEMS2 && f >= 0 && f < a.F && tq >= 0 && tq < a.T;
        q_lds[i] = ((r * C::ROW + 4 * qc) * CB_CKP + oc * 8);
        q_voff[i] = ok ? (unsigned)((oc * 8 * Fsrc + (unpool ? (f >> 1) : f)) * a.T + tq) * 4u : OOB;
        q_m[i] = (ok ? min(max(tlim - tq, 0), 4) : 0) | ((f & 1) << 8) | (it < C::IN_ITEMS2 ? 0x1000 : 0);
        q_c[i] = oc * 8;
    }
    u32x4_t rw[NS][C::W_PER_T2];
    const unsigned w_part_step = (unsigned)((size_t)C::KK * a.CoutP * a.CinP * 2);      // bytes between the parts of the packed weights
    unsigned w_voff[C::W_PER_T2];
    int w_lds[C::W_PER_T2];
#pragma unroll
    for (int i = 0; i < C::W_PER_T2; ++i) {
        const int item = tid + i * 256;
        const int oc = item % (CB_CK / 8), co = (item / (CB_CK / 8)) % CB_COUT_T, kw = item / ((CB_CK / 8) * CB_COUT_T);
        // [tap][CoutP][CinP] halfs; the tap's kh part and the chunk are added per load
        w_voff[i] = item < C::W_ITEMS2 ? (unsigned)((((size_t)kw * a.CoutP + cout0 + co) * a.CinP + oc * 8) * 2) : OOB;
        w_lds[i] = (kw * CB_COUT_T + co) * CB_CKP + oc * 8;
    }
    const unsigned w_kh_step = (unsigned)((size_t)KW * a.CoutP * a.CinP * 2);

    auto load_in = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < C::IN_PER_T2; ++i) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                rin[i][e] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, q_voff[i] + e * cstride, 0, 0);
                if (unpool) {
                    if (vec) {
                        ridx[i][e] = __builtin_amdgcn_raw_buffer_load_b32(rs_i, (q_voff[i] + e * cstride) >> 2, 0, 0);
                    } else {
                        unsigned w = 0;
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            w |= (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rs_i, ((q_voff[i] + e * cstride) >> 2) + k, 0, 0) << (8 * k);
                        ridx[i][e] = w;
                    }
                }
            }
            q_voff[i] += step_x;
        }
    };
    auto store_in = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < C::IN_PER_T2; ++i) {
            if (!(q_m[i] & 0x1000)) continue;
            const int n_ok = q_m[i] & 7, par = (q_m[i] >> 8) & 1;
            float v[8][4];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e][0] = __uint_as_float(rin[i][e].x); v[e][1] = __uint_as_float(rin[i][e].y);
                v[e][2] = __uint_as_float(rin[i][e].z); v[e][3] = __uint_as_float(rin[i][e].w);
                if (unpool) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[e][k] = (int)((ridx[i][e] >> (8 * k)) & 0xffu) == par ? v[e][k] : 0.f;
                }
            }
            if (pro) {
                const float4 s0 = *reinterpret_cast<const float4*>(sc_s + c0 + q_c[i]), s1 = *reinterpret_cast<const float4*>(sc_s + c0 + q_c[i] + 4);
                const float4 h0 = *reinterpret_cast<const float4*>(sh_s + c0 + q_c[i]), h1 = *reinterpret_cast<const float4*>(sh_s + c0 + q_c[i] + 4);
                const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float t = fmaf(v[e][k], sc[e], sh[e]);
                        v[e][k] = a.relu ? fmaxf(t, 0.f) : t;
                    }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool keep = k < n_ok;                           // zero padding is post-activation
                if constexpr (NS == 3) {
                    const float z = keep ? 1.f : 0.f;
                    const Bf3 p = split3x8(make_float4(v[0][k] * z, v[1][k] * z, v[2][k] * z, v[3][k] * z),
                                           make_float4(v[4][k] * z, v[5][k] * z, v[6][k] * z, v[7][k] * z));
                    *reinterpret_cast<u32x4_t*>(in_s + q_lds[i] + k * CB_CKP) = p.hi;
                    *reinterpret_cast<u32x4_t*>(in_s + C::IN_PART + q_lds[i] + k * CB_CKP) = p.mid;
                    *reinterpret_cast<u32x4_t*>(in_s + 2 * C::IN_PART + q_lds[i] + k * CB_CKP) = p.lo;
                } else {
                    u32x4_t o;
                    o.x = keep ? pack_bf16(v[0][k], v[1][k]) : 0u;
                    o.y = keep ? pack_bf16(v[2][k], v[3][k]) : 0u;
                    o.z = keep ? pack_bf16(v[4][k], v[5][k]) : 0u;
                    o.w = keep ? pack_bf16(v[6][k], v[7][k]) : 0u;
                    *reinterpret_cast<u32x4_t*>(in_s + q_lds[i] + k * CB_CKP) = o;
                }
            }
        }
    };
    auto load_w = [&](int c0, int kh) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < NS; ++p)
#pragma unroll
            for (int i = 0; i < C::W_PER_T2; ++i)
                rw[p][i] = __builtin_amdgcn_raw_buffer_load_b128(
                    rs_w, w_voff[i] == OOB ? OOB : w_voff[i] + (unsigned)p * w_part_step + (unsigned)kh * w_kh_step + (unsigned)c0 * 2u, 0, 0);
    };
    auto store_w = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < NS; ++p)
#pragma unroll
            for (int i = 0; i < C::W_PER_T2; ++i)
                if (tid + i * 256 < C::W_ITEMS2) *reinterpret_cast<u32x4_t*>(w_s + p * C::W_PART + w_lds[i]) = rw[p][i];
    };

    load_in();
    load_w(0, 0);
    for (int c0 = 0; c0 < a.CinP; c0 += CB_CK) {
        __syncthreads();                       // previous chunk's MFMA reads of in_s / w_s are done (and sc_s is staged)
        store_in(c0);
#pragma unroll 1
        for (int kh = 0; kh < KH; ++kh) {
            if (kh > 0) __syncthreads();       // previous kernel row's MFMAs are done with w_s
            store_w();
            __syncthreads();
            // prefetch behind the barrier: next kernel row's weights, and once per chunk the next chunk's input
            if (kh + 1 < KH) load_w(c0, kh + 1);
            else if (c0 + CB_CK < a.CinP) load_w(c0 + CB_CK, 0);
            if (kh == 0 && c0 + CB_CK < a.CinP) load_in();
#pragma unroll
            for (int kw = 0; kw < KW; ++kw) {
                u32x4_t af[C::MTW][NS];
#pragma unroll
                for (int m = 0; m < C::MTW; ++m)
#pragma unroll
                    for (int p = 0; p < NS; ++p)
                        af[m][p] = *reinterpret_cast<const u32x4_t*>(w_s + p * C::W_PART + ((size_t)(kw * CB_COUT_T + (wm * C::MTW + m) * 16 + lr)) * CB_CKP + lq * 8);
#pragma unroll
                for (int n = 0; n < C::NTW; ++n) {
                    const int fl = n / C::NTT, tt = wn * C::NTT + n % C::NTT;
                    u32x4_t bfr[NS];
#pragma unroll
                    for (int p = 0; p < NS; ++p)
                        bfr[p] = *reinterpret_cast<const u32x4_t*>(
                            in_s + p * C::IN_PART + ((size_t)((fl + kh) * C::ROW + tt * 16 + lr + kw + (C::HALO - PADW))) * CB_CKP + lq * 8);
#pragma unroll
                    for (int m = 0; m < C::MTW; ++m) {
                        if constexpr (NS == 3) acc[m][n] = mfma_x3(Bf3{af[m][0], af[m][1], af[m][2]}, Bf3{bfr[0], bfr[1], bfr[2]}, acc[m][n]);
                        else acc[m][n] = mfma_b16(af[m][0], bfr[0], acc[m][n]);
                    }
                }
            }
        }
    }
    conv_epilogue<CB_COUT_T, FT, TT, C::MTW, C::NTT, C::WN, POOL, DGRAD>(a, acc, st_s, b, f0, t0, cout0, sl, wm, wn, lq, lr, tid);
}

template <int NS, int FT, int TT, int KH, int KW, bool POOL, bool DGRAD>
static int launch_b2(const ConvFwdArgs& a, const unsigned short* wpb, hipStream_t s) {
    using C = ConvB2Cfg<NS, FT, TT, KH, KW, POOL>;
    if ((size_t)a.Cout * a.F * a.T * 4 >= (1ull << 29) || (size_t)a.Cin * a.F * a.T * 4 >= (1ull << 30)) {
        set_error("conv_bf16: one clip of the input / output must stay below 1 GiB / 512 MiB (Cin=%d Cout=%d F=%d T=%d)", a.Cin, a.Cout, a.F, a.T);
        return PBSED_E_ARG;
    }
    const int nTt = (a.T + TT - 1) / TT, nFt = (a.F + FT - 1) / FT;
    dim3 grid(nTt * nFt * a.B, a.CoutP / CB_COUT_T);
    const size_t lds = C::LDS_BYTES + (size_t)2 * a.CinP * sizeof(float);
    if (lds > 160 * 1024) { set_error("conv_bf16: %d input channels need %zu B of LDS", a.Cin, lds); return PBSED_E_UNSUPPORTED; }
    auto kern = conv_bf16v2_kernel<NS, FT, TT, KH, KW, POOL, DGRAD>;
    PBSED_DYN_LDS_ONCE(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a, wpb);
    return check_launch("conv_bf16v2");
}

// w [Cout][Cin][KH][KW] fp32 -> bf16 splits [split][tap][CoutP][CinP]; dgrad: roles swapped + taps flipped.
template <int NSPLIT>
__global__ void pack_conv_weights_bf16_kernel(const float* __restrict__ w, unsigned short* __restrict__ wpb, int Cout,
                                              int Cin, int KK, int OutP, int InP, int dgrad) {
    const size_t total = (size_t)KK * OutP * InP;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = i % InP, co = (i / InP) % OutP, kk = i / ((size_t)InP * OutP);
        float v = 0.f;
        if (!dgrad) {
            if (co < Cout && ci < Cin) v = w[((size_t)co * Cin + ci) * KK + kk];
        } else {   // kernel-output channel co = layer cin, kernel-input channel ci = layer cout
            if (co < Cin && ci < Cout) v = w[((size_t)ci * Cin + co) * KK + (KK - 1 - kk)];
        }
        unsigned short h[NSPLIT];
        split_bf16<NSPLIT>(v, h);
#pragma unroll
        for (int s = 0; s < NSPLIT; ++s) wpb[(size_t)s * total + i] = h[s];
    }
}


template <int NS>
static int dispatch_b2(const ConvFwdArgs& a, const unsigned short* wpb, int KH, int KW, int pool, int dgrad, hipStream_t s) {
#define CB2(FT_, TT_, KH_, KW_)                                                   \
    do {                                                                          \
        if (dgrad) return launch_b2<NS, FT_, TT_, KH_, KW_, false, true>(a, wpb, s);  \
        if (pool) return launch_b2<NS, FT_, TT_, KH_, KW_, true, false>(a, wpb, s);   \
        return launch_b2<NS, FT_, TT_, KH_, KW_, false, false>(a, wpb, s);            \
    } while (0)
#define CB21(TT_, KW_)                                                            \
    do {                                                                          \
        if (dgrad) return launch_b2<NS, 1, TT_, 1, KW_, false, true>(a, wpb, s);      \
        return launch_b2<NS, 1, TT_, 1, KW_, false, false>(a, wpb, s);                \
    } while (0)
    if (KH == 3 && KW == 3) CB2(4, 64, 3, 3);
    if (KH == 1 && KW == 3 && !pool) CB21(128, 3);
    if (KH == 1 && KW == 1 && !pool) CB21(128, 1);
    if (KH == 1 && KW == 1 && pool && !dgrad) return launch_b2<NS, 2, 64, 1, 1, true, false>(a, wpb, s);      // 1x1 conv2d under a (2,1) pool
#undef CB2
#undef CB21
    set_error("conv_bf16: unsupported kernel %dx%d pool=%d", KH, KW, pool);
    return PBSED_E_UNSUPPORTED;
}

template <int NSPLIT>
static int dispatch_b(const ConvFwdArgs& a, const unsigned short* wpb, int KH, int KW, int pool, int dgrad, hipStream_t s) {
    return dispatch_b2<NSPLIT>(a, wpb, KH, KW, pool, dgrad, s);
}

}  // namespace pbsed

using namespace pbsed;

extern "C" {

void pbsed_conv_pack_dims_bf16(int Cin, int Cout, int dgrad, int* InP, int* OutP) {
    const int in = dgrad ? Cout : Cin, out = dgrad ? Cin : Cout;
    *InP = (in + CB_CK - 1) / CB_CK * CB_CK;
    *OutP = (out + CB_COUT_T - 1) / CB_COUT_T * CB_COUT_T;
}

// wpb: uint16 [nsplit][KH*KW][OutP][InP]
int pbsed_pack_conv_weights_bf16(const float* w, unsigned short* wpb, int Cout, int Cin, int KH, int KW, int dgrad,
                                 int nsplit, void* stream) {
    int InP, OutP;
    pbsed_conv_pack_dims_bf16(Cin, Cout, dgrad, &InP, &OutP);
    const size_t total = (size_t)KH * KW * OutP * InP;
    const int nb = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    if (nsplit == 1)
        hipLaunchKernelGGL(pack_conv_weights_bf16_kernel<1>, dim3(nb), dim3(256), 0, (hipStream_t)stream, w, wpb, Cout, Cin,
                           KH * KW, OutP, InP, dgrad);
    else if (nsplit == 3)
        hipLaunchKernelGGL(pack_conv_weights_bf16_kernel<3>, dim3(nb), dim3(256), 0, (hipStream_t)stream, w, wpb, Cout, Cin,
                           KH * KW, OutP, InP, dgrad);
    else { set_error("pack_conv_weights_bf16: nsplit must be 1 or 3"); return PBSED_E_ARG; }
    return check_launch("pack_conv_weights_bf16");
}

// Same contracts as pbsed_conv_fwd / pbsed_conv_bwd_data with bf16 (nsplit = 1) or 3-way split bf16 operands.
int pbsed_conv_fwd_bf16(const float* x, const unsigned short* wpb, const float* bias, const float* scale,
                        const float* shift, int relu, const int* seq_len, float* y, unsigned char* pool_idx,
                        double* stats, int stats_per_cf, int B, int Cin, int Cout, int F, int T, int KH, int KW, int pool,
                        int nsplit, void* stream) {
    ConvFwdArgs a{};
    a.x = x; a.bias = bias; a.scale = scale; a.shift = shift; a.seq_len = seq_len;
    a.y = y; a.pool_idx = pool_idx; a.stats = stats; a.stats_cf = stats_per_cf; a.relu = relu;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.F = F; a.T = T;
    pbsed_conv_pack_dims_bf16(Cin, Cout, 0, &a.CinP, &a.CoutP);
    if (pool && (F % 2)) { set_error("conv_fwd_bf16: pool needs even F"); return PBSED_E_ARG; }
    if (nsplit == 1) return dispatch_b<1>(a, wpb, KH, KW, pool, 0, (hipStream_t)stream);
    if (nsplit == 3) return dispatch_b<3>(a, wpb, KH, KW, pool, 0, (hipStream_t)stream);
    set_error("conv_fwd_bf16: nsplit must be 1 or 3");
    return PBSED_E_ARG;
}

// pbsed_conv_fwd_bf16 + a residual connection ending at this layer (added to the biased, pooled output before the store and
// the statistics, as in pbsed_conv_fwd_res): the 1x1 conv2d layers of the 'deep' configuration.
int pbsed_conv_fwd_bf16_res(const float* x, const unsigned short* wpb, const float* bias, const float* scale,
                            const float* shift, int relu, const int* seq_len, float* y, unsigned char* pool_idx,
                            double* stats, int stats_per_cf, int B, int Cin, int Cout, int F, int T, int KH, int KW, int pool,
                            int nsplit, const float* residual, void* stream) {
    ConvFwdArgs a{};
    a.x = x; a.bias = bias; a.scale = scale; a.shift = shift; a.seq_len = seq_len;
    a.y = y; a.pool_idx = pool_idx; a.stats = stats; a.stats_cf = stats_per_cf; a.relu = relu; a.res = residual;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.F = F; a.T = T;
    pbsed_conv_pack_dims_bf16(Cin, Cout, 0, &a.CinP, &a.CoutP);
    if (pool && (F % 2)) { set_error("conv_fwd_bf16_res: pool needs even F"); return PBSED_E_ARG; }
    if (nsplit == 1) return dispatch_b<1>(a, wpb, KH, KW, pool, 0, (hipStream_t)stream);
    if (nsplit == 3) return dispatch_b<3>(a, wpb, KH, KW, pool, 0, (hipStream_t)stream);
    set_error("conv_fwd_bf16_res: nsplit must be 1 or 3");
    return PBSED_E_ARG;
}

int pbsed_conv_bwd_data_bf16(const float* g, const unsigned short* wdb, const unsigned char* unpool_idx,
                             const int* seq_len, float* dz, const float* bx, const float* bmean, const float* binvstd,
                             const float* bscale, const float* bshift, int relu, double* stats, int B, int Cin, int Cout,
                             int F, int T, int KH, int KW, int nsplit, void* stream) {
    ConvFwdArgs a{};
    a.x = g; a.seq_len = seq_len; a.y = dz; a.unpool_idx = unpool_idx;
    a.bx = bx; a.bmean = bmean; a.binvstd = binvstd; a.bscale = bscale; a.bshift = bshift;
    a.relu = relu; a.stats = bx ? stats : nullptr;
    a.B = B; a.Cin = Cout; a.Cout = Cin; a.F = F; a.T = T;      // roles swapped
    pbsed_conv_pack_dims_bf16(Cin, Cout, 1, &a.CinP, &a.CoutP);
    if (unpool_idx && (F % 2)) { set_error("conv_bwd_data_bf16: unpool needs even F"); return PBSED_E_ARG; }
    if (nsplit == 1) return dispatch_b<1>(a, wdb, KH, KW, 0, 1, (hipStream_t)stream);
    if (nsplit == 3) return dispatch_b<3>(a, wdb, KH, KW, 0, 1, (hipStream_t)stream);
    set_error("conv_bwd_data_bf16: nsplit must be 1 or 3");
    return PBSED_E_ARG;
}

}  // extern "C"
