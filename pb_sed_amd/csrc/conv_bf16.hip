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

template <int FT, int TT, int KH, int KW, int NSPLIT, bool POOL, bool DGRAD>
__global__ __launch_bounds__(256) void conv_bf16_kernel(ConvFwdArgs a, const unsigned short* __restrict__ wpb) {
    using C = ConvBCfg<FT, TT, KH, KW, NSPLIT, POOL>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned short* in_s = reinterpret_cast<unsigned short*>(smem_raw);            // [NSPLIT][ROWS][ROW][CKP]
    unsigned short* w_s = in_s + C::IN_HALFS;                                       // [NSPLIT][KW][64][CKP]
    float* st_s = reinterpret_cast<float*>(w_s + C::W_HALFS);                       // [64][FO_T][2]

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

    f32x4 acc[C::MTW][C::NTW];
#pragma unroll
    for (int m = 0; m < C::MTW; ++m)
#pragma unroll
        for (int n = 0; n < C::NTW; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int c0 = 0; c0 < a.CinP; c0 += CB_CK) {
        __syncthreads();                       // all MFMA reads of the previous chunk are done
        // ---- stage the input halo tile for 32 channels: item = (octet, row, position), position fastest
#pragma unroll
        for (int i = 0; i < C::IN_PER_T; ++i) {
            const int item = tid + i * 256;
            if (item < C::IN_ITEMS) {
                const int p = item % C::ROW, r = (item / C::ROW) % C::ROWS, oc = item / (C::ROW * C::ROWS);
                const int f = f0 - PADH + r, t = t0 - C::HALO + p;
                const bool pos_ok = f >= 0 && f < a.F && t >= 0 && t < tlim;
                us8 o[NSPLIT];
#pragma unroll
                for (int s = 0; s < NSPLIT; ++s) o[s] = us8{0, 0, 0, 0, 0, 0, 0, 0};
                if (pos_ok) {
                    const size_t off = ((size_t)(unpool ? (f >> 1) : f)) * a.T + t;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int cin = c0 + oc * 8 + e;
                        float v = 0.f;
                        if (cin < a.Cin) {
                            const size_t o1 = (size_t)(b * a.Cin + cin) * Fsrc * a.T + off;
                            v = a.x[o1];
                            if (unpool) v = (a.unpool_idx[o1] == (uint8_t)(f & 1)) ? v : 0.f;
                            if (pro) {
                                v = fmaf(v, a.scale[cin], a.shift[cin]);
                                if (a.relu) v = fmaxf(v, 0.f);
                            }
                        }
                        unsigned short h[NSPLIT];
                        split_bf16<NSPLIT>(v, h);
#pragma unroll
                        for (int s = 0; s < NSPLIT; ++s) o[s][e] = h[s];
                    }
                }
#pragma unroll
                for (int s = 0; s < NSPLIT; ++s)
                    *reinterpret_cast<us8*>(in_s + ((size_t)((s * C::ROWS + r) * C::ROW + p)) * CB_CKP + oc * 8) = o[s];
            }
        }
#pragma unroll 1
        for (int kh = 0; kh < KH; ++kh) {
            if (kh > 0) __syncthreads();       // previous kernel row's MFMAs are done with w_s
#pragma unroll
            for (int i = 0; i < C::W_PER_T; ++i) {
                const int item = tid + i * 256;
                if (item < C::W_ITEMS) {
                    const int oc = item % (CB_CK / 8), co = (item / (CB_CK / 8)) % CB_COUT_T, kw = item / ((CB_CK / 8) * CB_COUT_T);
#pragma unroll
                    for (int s = 0; s < NSPLIT; ++s) {
                        const us8 v = *reinterpret_cast<const us8*>(
                            wpb + (((size_t)(s * C::KK + kh * KW + kw) * a.CoutP + cout0 + co) * a.CinP + c0 + oc * 8));
                        *reinterpret_cast<us8*>(w_s + ((size_t)((s * KW + kw) * CB_COUT_T + co)) * CB_CKP + oc * 8) = v;
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int kw = 0; kw < KW; ++kw) {
                us8 af[C::MTW][NSPLIT];
#pragma unroll
                for (int m = 0; m < C::MTW; ++m)
#pragma unroll
                    for (int s = 0; s < NSPLIT; ++s)
                        af[m][s] = *reinterpret_cast<const us8*>(
                            w_s + ((size_t)((s * KW + kw) * CB_COUT_T + (wm * C::MTW + m) * 16 + lr)) * CB_CKP + lq * 8);
#pragma unroll
                for (int n = 0; n < C::NTW; ++n) {
                    const int fl = n / C::NTT, tt = wn * C::NTT + n % C::NTT;
                    us8 bfr[NSPLIT];
#pragma unroll
                    for (int s = 0; s < NSPLIT; ++s)
                        bfr[s] = *reinterpret_cast<const us8*>(
                            in_s + ((size_t)((s * C::ROWS + fl + kh) * C::ROW + tt * 16 + lr + kw + (C::HALO - PADW))) * CB_CKP + lq * 8);
#pragma unroll
                    for (int m = 0; m < C::MTW; ++m) {
                        f32x4 c = acc[m][n];
                        if (NSPLIT == 3) {                 // small terms first
                            c = mfma_bf16(af[m][1], bfr[1], c);
                            c = mfma_bf16(af[m][2], bfr[0], c);
                            c = mfma_bf16(af[m][0], bfr[2], c);
                            c = mfma_bf16(af[m][1], bfr[0], c);
                            c = mfma_bf16(af[m][0], bfr[1], c);
                        }
                        c = mfma_bf16(af[m][0], bfr[0], c);
                        acc[m][n] = c;
                    }
                }
            }
        }
    }
    conv_epilogue<CB_COUT_T, FT, TT, C::MTW, C::NTT, C::WN, POOL, DGRAD>(a, acc, st_s, b, f0, t0, cout0, sl, wm, wn, lq, lr, tid);
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

template <int FT, int TT, int KH, int KW, int NSPLIT, bool POOL, bool DGRAD>
static int launch_b(const ConvFwdArgs& a, const unsigned short* wpb, hipStream_t s) {
    using C = ConvBCfg<FT, TT, KH, KW, NSPLIT, POOL>;
    if ((size_t)a.Cout * a.F * a.T * 4 >= (1ull << 29)) {     // conv_epilogue addresses one clip with 32-bit offsets
        set_error("conv_bf16: one clip of the output must stay below 512 MiB (Cout=%d F=%d T=%d)", a.Cout, a.F, a.T);
        return PBSED_E_ARG;
    }
    const int nTt = (a.T + TT - 1) / TT, nFt = (a.F + FT - 1) / FT;
    dim3 grid(nTt * nFt * a.B, a.CoutP / CB_COUT_T);
    auto kern = conv_bf16_kernel<FT, TT, KH, KW, NSPLIT, POOL, DGRAD>;
    PBSED_DYN_LDS_ONCE(kern, C::LDS_BYTES);
    hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, s, a, wpb);
    return check_launch("conv_bf16");
}

template <int NSPLIT>
static int dispatch_b(const ConvFwdArgs& a, const unsigned short* wpb, int KH, int KW, int pool, int dgrad, hipStream_t s) {
#define CB(FT_, TT_, KH_, KW_)                                                             \
    do {                                                                                   \
        if (dgrad) return launch_b<FT_, TT_, KH_, KW_, NSPLIT, false, true>(a, wpb, s);    \
        if (pool) return launch_b<FT_, TT_, KH_, KW_, NSPLIT, true, false>(a, wpb, s);     \
        return launch_b<FT_, TT_, KH_, KW_, NSPLIT, false, false>(a, wpb, s);              \
    } while (0)
#define CB1(TT_, KW_)                                                                      \
    do {                                                                                   \
        if (dgrad) return launch_b<1, TT_, 1, KW_, NSPLIT, false, true>(a, wpb, s);        \
        return launch_b<1, TT_, 1, KW_, NSPLIT, false, false>(a, wpb, s);                  \
    } while (0)
    if (KH == 3 && KW == 3) CB(NSPLIT == 3 ? 2 : 4, 64, 3, 3);
    if (KH == 1 && KW == 3 && !pool) CB1(128, 3);
    if (KH == 1 && KW == 1 && !pool) CB1(128, 1);
#undef CB
#undef CB1
    set_error("conv_bf16: unsupported kernel %dx%d pool=%d", KH, KW, pool);
    return PBSED_E_UNSUPPORTED;
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
