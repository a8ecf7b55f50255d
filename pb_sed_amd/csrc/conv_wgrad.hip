// Convolution weight gradient on fp32 MFMA for gfx950:  dW[co][ci][kh][kw] += sum_{b,f,t}
// dY[b,co,f,t] * a[b,ci,f+kh-P,t+kw-P] with a = prologue(x) (BN-apply + ReLU + seq mask recomputed
// while staging, never materialised) and dY the un-pooled output gradient (gathered through the
// forward's pool argmax).  This is the backward twin of the op sites listed in conv.hip
// (torch autograd of Conv2d/Conv1d in the reference: pb_sed/models/weak_label/crnn.py:93).
//
// GEMM view: M = Cout (A = dY), N = (cin,kh,kw) (B = shifted a), K = spatial (b,f,t), split over
// blocks; partial results are reduced with float atomics into the [Cout,Cin,KH,KW] gradient.
#include "common.h"
#include "pbsed_internal.h"

namespace pbsed {

constexpr int pad2mod32(int n) { return ((n + 29) / 32) * 32 + 2; }

template <int KH, int KW, int WAVES, int NCG, bool TAPN>
__global__ __launch_bounds__(WAVES * 64) void conv_wgrad_kernel(ConvWgradArgs a) {
    constexpr int FT = (KH == 3) ? 2 : 1, TT = 64;
    constexpr int KK = KH * KW, NT = WAVES * 64, COUT_T = WAVES * 16;
    constexpr int ROWS = FT + KH - 1, ROW = TT + KW - 1;
    constexpr int PLANE_Y = pad2mod32(FT * TT), PLANE_A = pad2mod32(ROWS * ROW);
    constexpr int PADH = (KH - 1) / 2, PADW = (KW - 1) / 2;
    constexpr int NACC = TAPN ? 1 : NCG * KK;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dy_s = smem;                          // [COUT_T][PLANE_Y]
    float* a_s = smem + COUT_T * PLANE_Y;        // [NCG*16][PLANE_A]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane >> 4, lr = lane & 15;
    const int cin0 = blockIdx.y * NCG * 16, cout0 = blockIdx.z * COUT_T;
    const int nTt = (a.T + TT - 1) / TT, nFt = (a.F + FT - 1) / FT;
    const int nChunks = a.B * nFt * nTt;
    const bool pro = a.scale != nullptr;
    const bool do_bias = (a.db != nullptr) && (blockIdx.y == 0);
    const int Fg = a.unpool_idx ? a.F / 2 : a.F;

    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 accb = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int chunk = blockIdx.x; chunk < nChunks; chunk += gridDim.x) {
        int c = chunk;
        const int t0 = (c % nTt) * TT; c /= nTt;
        const int f0 = (c % nFt) * FT;
        const int b = c / nFt;
        const int sl = a.seq_len ? min(a.seq_len[b], a.T) : a.T;
        __syncthreads();
        for (int idx = tid; idx < COUT_T * FT * TT; idx += NT) {
            const int cl = idx / (FT * TT), rem = idx % (FT * TT);
            const int fl = rem / TT, tc = rem % TT;
            const int cout = cout0 + cl, f = f0 + fl, t = t0 + tc;
            float v = 0.f;
            if (cout < a.Cout && f < a.F && t < a.T) {
                if (a.unpool_idx) {
                    const size_t o = ((size_t)(b * a.Cout + cout) * Fg + (f >> 1)) * a.T + t;
                    v = (a.unpool_idx[o] == (uint8_t)(f & 1)) ? a.g[o] : 0.f;
                } else {
                    v = a.g[((size_t)(b * a.Cout + cout) * a.F + f) * a.T + t];
                }
            }
            dy_s[cl * PLANE_Y + rem] = v;
        }
        for (int idx = tid; idx < NCG * 16 * ROWS * ROW; idx += NT) {
            const int cl = idx / (ROWS * ROW), rem = idx % (ROWS * ROW);
            const int r = rem / ROW, col = rem % ROW;
            const int cin = cin0 + cl, f = f0 - PADH + r, t = t0 - PADW + col;
            const int tlim = pro ? sl : a.T;
            float v = 0.f;
            if (cin < a.Cin && f >= 0 && f < a.F && t >= 0 && t < tlim) {
                v = a.x[((size_t)(b * a.Cin + cin) * a.F + f) * a.T + t];
                if (pro) {
                    v = fmaf(v, a.scale[cin], a.shift[cin]);
                    if (a.relu) v = fmaxf(v, 0.f);
                }
            }
            a_s[cl * PLANE_A + rem] = v;
        }
        __syncthreads();
#pragma unroll
        for (int fl = 0; fl < FT; ++fl) {
#pragma unroll 4
            for (int tq = 0; tq < TT / 4; ++tq) {
                const float af = dy_s[(wave * 16 + lr) * PLANE_Y + fl * TT + tq * 4 + lq];
                if (TAPN) {
                    const int kh = lr / KW, kw = lr % KW;
                    const float bf = (lr < KK) ? a_s[(fl + kh) * ROW + tq * 4 + lq + kw] : 0.f;
                    acc[0] = mfma16(af, bf, acc[0]);
                } else {
#pragma unroll
                    for (int g = 0; g < NCG; ++g)
#pragma unroll
                        for (int kk = 0; kk < KK; ++kk) {
                            const int kh = kk / KW, kw = kk % KW;
                            const float bf =
                                a_s[(g * 16 + lr) * PLANE_A + (fl + kh) * ROW + tq * 4 + lq + kw];
                            acc[g * KK + kk] = mfma16(af, bf, acc[g * KK + kk]);
                        }
                }
                if (do_bias) accb = mfma16(af, 1.0f, accb);
            }
        }
    }

#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int cout = cout0 + wave * 16 + lq * 4 + r;
        if (cout >= a.Cout) continue;
        if (TAPN) {
            if (lr < KK) atomicAdd(&a.dw[(size_t)cout * a.Cin * KK + lr], acc[0][r]);
        } else {
#pragma unroll
            for (int g = 0; g < NCG; ++g) {
                const int cin = cin0 + g * 16 + lr;
                if (cin < a.Cin) {
#pragma unroll
                    for (int kk = 0; kk < KK; ++kk)
                        atomicAdd(&a.dw[((size_t)cout * a.Cin + cin) * KK + kk], acc[g * KK + kk][r]);
                }
            }
        }
        if (do_bias && lr == 0) atomicAdd(&a.db[cout], accb[r]);
    }
}

template <int KH, int KW, int WAVES, int NCG, bool TAPN>
static int launch_wgrad(const ConvWgradArgs& a, hipStream_t s) {
    constexpr int FT = (KH == 3) ? 2 : 1, TT = 64;
    const int nTt = (a.T + TT - 1) / TT, nFt = (a.F + FT - 1) / FT;
    const int nChunks = a.B * nFt * nTt;
    const int gy = TAPN ? 1 : (a.Cin + NCG * 16 - 1) / (NCG * 16);
    const int gz = (a.Cout + WAVES * 16 - 1) / (WAVES * 16);
    int split = 1024 / (gy * gz);
    if (split < 1) split = 1;
    if (split > nChunks) split = nChunks;
    dim3 grid(split, gy, gz);
    constexpr int ROWS = FT + KH - 1, ROW = TT + KW - 1;
    const size_t lds = (WAVES * 16 * pad2mod32(FT * TT) + NCG * 16 * pad2mod32(ROWS * ROW)) * sizeof(float);
    auto kern = conv_wgrad_kernel<KH, KW, WAVES, NCG, TAPN>;
    static bool attr_set = false;
    if (!attr_set && lds > 48 * 1024) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, grid, dim3(WAVES * 64), lds, s, a);
    return check_launch("conv_wgrad");
}

int conv_wgrad_launch(const ConvWgradArgs& a, int KH, int KW, hipStream_t s) {
    if (a.unpool_idx && (a.F % 2)) { set_error("conv_wgrad: unpool needs even F"); return PBSED_E_ARG; }
    const bool big = a.Cout >= 64;
    const bool wide = a.Cin >= 32;
    if (KH == 3 && KW == 3) {
        if (a.Cin == 1) return big ? launch_wgrad<3, 3, 4, 1, true>(a, s) : launch_wgrad<3, 3, 1, 1, true>(a, s);
        if (big) return wide ? launch_wgrad<3, 3, 4, 2, false>(a, s) : launch_wgrad<3, 3, 4, 1, false>(a, s);
        if (a.Cout >= 32) return wide ? launch_wgrad<3, 3, 2, 2, false>(a, s) : launch_wgrad<3, 3, 2, 1, false>(a, s);
        return launch_wgrad<3, 3, 1, 1, false>(a, s);
    }
    if (KH == 1 && KW == 3) {
        return big ? launch_wgrad<1, 3, 4, 2, false>(a, s) : launch_wgrad<1, 3, 1, 2, false>(a, s);
    }
    if (KH == 1 && KW == 1) {
        return big ? launch_wgrad<1, 1, 4, 2, false>(a, s) : launch_wgrad<1, 1, 1, 2, false>(a, s);
    }
    set_error("conv_wgrad: unsupported kernel %dx%d", KH, KW);
    return PBSED_E_UNSUPPORTED;
}

}  // namespace pbsed
