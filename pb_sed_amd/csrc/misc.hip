// Small HBM-bound helper kernels: weight packing, BatchNorm finalise / backward, losses, Adam.
// Reference op sites: padertorch Normalization ('batch', eps 1e-3; pb_sed/experiments/
// weak_label_crnn/training.py:223-225), losses pb_sed/models/weak_label/crnn.py:107-206 and
// pb_sed/models/strong_label/crnn.py:106-112, optimiser training.py:264-269 (Adam + grad-norm clip).
#include "common.h"
#include "pack_elems.h"
#include "pbsed_internal.h"

namespace pbsed {

// w [Cout][Cin][KH][KW] -> wp [KK][CinP][CoutP] (zero padded).  dgrad=1: roles swapped and taps
// flipped: wp[kk'][co][ci] = w[co][ci][KH-1-kh'][KW-1-kw'] laid out as [KK][CoutP'(=in)][CinP'(=out)].
__global__ void pack_conv_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout,
                                         int Cin, int KH, int KW, int InP, int OutP, int dgrad) {
    const int KK = KH * KW;
    const size_t total = (size_t)KK * InP * OutP;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x)
        wp[i] = pack_direct_elem(w, i, Cout, Cin, KK, InP, OutP, dgrad);
}

// Every packed copy of a model in one launch: blockIdx.y = descriptor (device array), blockIdx.x strides its elements.
struct PackDesc {
    const float* src;
    float* dst;
    int Cout, Cin, KH, KW, InP, OutP;
    int mode;      // 0/1: direct forward / data-gradient layout, 2/3: Winograd forward / data-gradient layout,
                   // 4/5: bf16 [tap][OutP][InP] forward / data-gradient layout (dst holds uint16), 6/7: its three-part form,
                   // 8/9: bf16x3 Winograd fragment layout forward / data gradient (csrc/conv_winox3.hip),
                   // 10/11: bf16x3 Conv1d fragment layout forward / data gradient (csrc/conv1d_pc.hip),
                   // 12: dst [Cin][Cout] = src^T (fp32; KH, KW, InP, OutP unused),
                   // 14/15: bf16x3 few-channel 3x3 fragment layout forward / data gradient (csrc/conv_s16.hip)
    int pad_;
};

__device__ __forceinline__ unsigned short f2bf_rne(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

__global__ void pack_batched_kernel(const PackDesc* __restrict__ descs) {
    const PackDesc d = descs[blockIdx.y];
    const int KK = d.KH * d.KW;
    if (d.mode >= 14) {                                  // csrc/conv_s16.hip: fragment-ordered three-part few-channel 3x3 weights
        const size_t total = (size_t)5 * (d.OutP / 16) * 512;
        unsigned short* dst = reinterpret_cast<unsigned short*>(d.dst);
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
            pack_s16_value(d.src, dst, i, d.Cout, d.Cin, d.OutP, d.mode & 1);
        return;
    }
    if (d.mode >= 12) {                                  // plain transpose of a [Cout][Cin] matrix (the GRU scans' W^T operands)
        const size_t total = (size_t)d.Cout * d.Cin;
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
            const size_t c = i / d.Cout, r = i % d.Cout;     // dst [Cin][Cout]
            d.dst[i] = d.src[r * d.Cin + c];
        }
        return;
    }
    if (d.mode >= 10) {                                  // csrc/conv1d_pc.hip: fragment-ordered three-part Conv1d weights (KH = 1)
        const size_t total = (size_t)d.KW * d.InP * d.OutP;           // weights: a thread forms one and writes its three parts
        unsigned short* dst = reinterpret_cast<unsigned short*>(d.dst);
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
            pack_c1x3_value(d.src, dst, i, d.Cout, d.Cin, d.KW, d.InP, d.OutP, d.mode & 1);
        return;
    }
    if (d.mode >= 8) {                                   // csrc/conv_winox3.hip: fragment-ordered three-part Winograd weights
        const size_t total = (size_t)18 * d.InP * d.OutP;
        unsigned short* dst = reinterpret_cast<unsigned short*>(d.dst);
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
            pack_winox3_value(d.src, dst, i, d.Cout, d.Cin, d.InP, d.OutP, d.mode & 1);
        return;
    }
    if (d.mode >= 4) {                                   // csrc/conv_bf16.hip layout: [tap][OutP][InP], input channels innermost
        const size_t total = (size_t)KK * d.OutP * d.InP;
        unsigned short* dst = reinterpret_cast<unsigned short*>(d.dst);
        const int dgrad = d.mode & 1;
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
            const int ci = i % d.InP, co = (i / d.InP) % d.OutP, kk = i / ((size_t)d.InP * d.OutP);
            float v = 0.f;
            if (!dgrad) {
                if (co < d.Cout && ci < d.Cin) v = d.src[((size_t)co * d.Cin + ci) * KK + kk];
            } else if (co < d.Cin && ci < d.Cout) {
                v = d.src[((size_t)ci * d.Cin + co) * KK + (KK - 1 - kk)];
            }
            if (d.mode >= 6) {                             // three exact-ish parts [part][tap][OutP][InP] (bf16x3 operands)
                const unsigned short h0 = f2bf_rne(v);
                float r = v - __uint_as_float((unsigned)h0 << 16);
                const unsigned short h1 = f2bf_rne(r);
                r -= __uint_as_float((unsigned)h1 << 16);
                dst[i] = h0; dst[total + i] = h1; dst[2 * total + i] = f2bf_rne(r);
            } else {
                dst[i] = f2bf_rne(v);
            }
        }
        return;
    }
    const bool wino = d.mode >= 2;
    const size_t total = (size_t)(wino ? 18 : KK) * d.InP * d.OutP;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        d.dst[i] = wino ? pack_wino_elem(d.src, i, d.Cout, d.Cin, d.InP, d.OutP, d.mode & 1)
                        : pack_direct_elem(d.src, i, d.Cout, d.Cin, KK, d.InP, d.OutP, d.mode & 1);
}

// sums [C][2] (sum x, sum x^2 over masked positions) -> batch statistics + fused scale/shift.
__global__ void bn_finalize_kernel(const double* __restrict__ sums, double count, const float* gamma,
                                   const float* beta, float eps, float momentum, float* running_mean,
                                   float* running_power, float* mean, float* invstd, float* scale,
                                   float* shift, int C) {
    // one 32-lane half-wave per channel: lane k fetches slot k, the two sums are folded with shuffles (one memory
    // round trip instead of a chain of PBSED_STAT_SLOTS dependent additions behind their loads)
    static_assert(PBSED_STAT_SLOTS == 32, "half-wave per channel");
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) >> 5, k = threadIdx.x & 31;
    double s1 = 0, s2 = 0;
    if (c < C) { s1 = sums[((size_t)k * C + c) * 2]; s2 = sums[((size_t)k * C + c) * 2 + 1]; }
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if (c >= C || k != 0) return;
    const double m = s1 / count;
    double var = s2 / count - m * m;
    if (var < 0) var = 0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = (float)m;
    invstd[c] = is;
    const float sc = gamma[c] * is;
    scale[c] = sc;
    shift[c] = beta[c] - (float)m * sc;
    if (running_mean) {
        running_mean[c] = momentum * running_mean[c] + (1.f - momentum) * (float)m;
        running_power[c] = momentum * running_power[c] + (1.f - momentum) * (float)(var + m * m);
    }
}

__global__ void bn_eval_params_kernel(const float* gamma, const float* beta, float eps,
                                      const float* running_mean, const float* running_power,
                                      float* mean, float* invstd, float* scale, float* shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float m = running_mean[c];
    const float var = fmaxf(running_power[c] - m * m, 0.f);
    const float is = rsqrtf(var + eps);
    mean[c] = m; invstd[c] = is;
    scale[c] = gamma[c] * is;
    shift[c] = beta[c] - m * gamma[c] * is;
}

// sums [C][2] = (sum dz, sum dz*xhat) -> dgamma, dbeta (+=) and the two means used by bn_bwd_apply.
__global__ void bn_bwd_finalize_kernel(const double* __restrict__ sums, double count, float* dgamma,
                                       float* dbeta, float* m1, float* m2, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0, s2 = 0;
    for (int k = 0; k < PBSED_STAT_SLOTS; ++k) { s1 += sums[((size_t)k * C + c) * 2]; s2 += sums[((size_t)k * C + c) * 2 + 1]; }
    if (dbeta) dbeta[c] += (float)s1;
    if (dgamma) dgamma[c] += (float)s2;
    m1[c] = (float)(s1 / count);
    m2[c] = (float)(s2 / count);
}

// BN backward as a dY prologue of the consumers (pbsed_conv_bwd_weight_bng): sums -> dgamma, dbeta (+=) and, per channel,
// the three coefficients of  dx = gamma*invstd * (dz - m1 - xhat*m2) = k1 * dz + k2 * x + k3:
//   k1 = scale,  k2 = -scale * invstd * m2,  k3 = -scale * (m1 - mean * invstd * m2);     coef = [3][C].
__global__ void bn_bwd_coef_kernel(const double* __restrict__ sums, double count, const float* mean, const float* invstd,
                                   const float* scale, float* dgamma, float* dbeta, float* coef, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0, s2 = 0;
    for (int k = 0; k < PBSED_STAT_SLOTS; ++k) { s1 += sums[((size_t)k * C + c) * 2]; s2 += sums[((size_t)k * C + c) * 2 + 1]; }
    if (dbeta) dbeta[c] += (float)s1;
    if (dgamma) dgamma[c] += (float)s2;
    const float m1 = (float)(s1 / count), m2 = (float)(s2 / count);
    const float sc = scale[c], is = invstd[c], mu = mean[c];
    coef[c] = sc;
    coef[C + c] = -sc * is * m2;
    coef[2 * C + c] = -sc * (m1 - mu * is * m2);
}

// in place: dz -> dx = gamma*invstd * (dz - m1 - xhat*m2) on masked positions (0 elsewhere).
// tensors [B, C, S, T] with S = inner rows per channel (F for 2-D, 1 for 1-D).  VEC: 4 frames per thread.
template <bool VEC>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(float* __restrict__ dz, const float* __restrict__ x,
                                                           const float* mean, const float* invstd,
                                                           const float* scale, const float* m1,
                                                           const float* m2, const int* seq_len, int B,
                                                           int C, int S, int T) {
    constexpr int W = VEC ? 4 : 1;
    const int Tq = T / W;
    const size_t total = (size_t)B * C * S * Tq;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int t = (i % Tq) * W;
        const size_t row = i / Tq;
        const int c = (row / S) % C, b = row / ((size_t)S * C);
        const int sl = seq_len ? seq_len[b] : T;
        const float mu = mean[c], is = invstd[c], sc = scale[c], a1 = m1[c], a2 = m2[c];
        if (VEC) {
            float4 d = reinterpret_cast<float4*>(dz)[i];
            const float4 xv = reinterpret_cast<const float4*>(x)[i];
            d.x = (t + 0 < sl) ? sc * (d.x - a1 - (xv.x - mu) * is * a2) : 0.f;
            d.y = (t + 1 < sl) ? sc * (d.y - a1 - (xv.y - mu) * is * a2) : 0.f;
            d.z = (t + 2 < sl) ? sc * (d.z - a1 - (xv.z - mu) * is * a2) : 0.f;
            d.w = (t + 3 < sl) ? sc * (d.w - a1 - (xv.w - mu) * is * a2) : 0.f;
            reinterpret_cast<float4*>(dz)[i] = d;
        } else {
            dz[i] = (t < sl) ? sc * (dz[i] - a1 - (x[i] - mu) * is * a2) : 0.f;
        }
    }
}

// bn_bwd_finalize + bn_bwd_apply in one launch: block (c, b) sums the PBSED_STAT_SLOTS partial (sum dz, sum dz*xhat)
// of its channel itself (64 doubles), block b == 0 also accumulates dgamma / dbeta, then the [S, T] slab of (b, c) is
// rewritten in place.  15 fewer launches per backward pass.
template <bool VEC>
__global__ __launch_bounds__(256) void bn_bwd_fused_kernel(float* __restrict__ dz, const float* __restrict__ x,
                                                           const double* __restrict__ sums, double count,
                                                           const float* mean, const float* invstd, const float* scale,
                                                           float* dgamma, float* dbeta, const int* seq_len, int C, int S,
                                                           int T) {
    __shared__ float sm[2];
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    if (tid < 64) {
        const int which = tid >> 5, k = tid & 31;
        double v = (k < PBSED_STAT_SLOTS) ? sums[((size_t)k * C + c) * 2 + which] : 0.0;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) v += __shfl_xor(v, o);
        if (k == 0) {
            sm[which] = (float)(v / count);
            if (b == 0) {
                if (which == 0 && dbeta) dbeta[c] += (float)v;
                if (which == 1 && dgamma) dgamma[c] += (float)v;
            }
        }
    }
    __syncthreads();
    const float a1 = sm[0], a2 = sm[1], mu = mean[c], is = invstd[c], sc = scale[c];
    const int sl = seq_len ? seq_len[b] : T;
    const size_t base = ((size_t)b * C + c) * S * T;
    if (VEC) {
        const int Tq = T / 4, n = S * Tq;
        float4* d4 = reinterpret_cast<float4*>(dz + base);
        const float4* x4 = reinterpret_cast<const float4*>(x + base);
        for (int i = tid; i < n; i += 256) {
            const int t = (i % Tq) * 4;
            float4 d = d4[i];
            const float4 xv = x4[i];
            d.x = (t + 0 < sl) ? sc * (d.x - a1 - (xv.x - mu) * is * a2) : 0.f;
            d.y = (t + 1 < sl) ? sc * (d.y - a1 - (xv.y - mu) * is * a2) : 0.f;
            d.z = (t + 2 < sl) ? sc * (d.z - a1 - (xv.z - mu) * is * a2) : 0.f;
            d.w = (t + 3 < sl) ? sc * (d.w - a1 - (xv.w - mu) * is * a2) : 0.f;
            d4[i] = d;
        }
    } else {
        for (int i = tid; i < S * T; i += 256) {
            const int t = i % T;
            dz[base + i] = (t < sl) ? sc * (dz[base + i] - a1 - (x[base + i] - mu) * is * a2) : 0.f;
        }
    }
}

// The same for SHORT slabs (S * T below 2 048 elements: the Conv1d layers, 500 frames per (clip, channel)): a 256-thread block
// per slab leaves most lanes idle and pays the slot reduction + barrier per 2 KB of data (2048 channels: 90 us for a tensor
// the long-slab form moves in 47).  Here a block takes four channels, one per wave; the wave reduces its channel's slots
// itself, no LDS, no barrier.
template <bool VEC>
__global__ __launch_bounds__(256) void bn_bwd_fused_wave_kernel(float* __restrict__ dz, const float* __restrict__ x,
                                                                const double* __restrict__ sums, double count,
                                                                const float* mean, const float* invstd, const float* scale,
                                                                float* dgamma, float* dbeta, const int* seq_len, int C, int S,
                                                                int T) {
    const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
    if (c >= C) return;
    const int which = lane >> 5, k = lane & 31;
    double v = (k < PBSED_STAT_SLOTS) ? sums[((size_t)k * C + c) * 2 + which] : 0.0;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) v += __shfl_xor(v, o);
    if (k == 0 && b == 0) {
        if (which == 0 && dbeta) dbeta[c] += (float)v;
        if (which == 1 && dgamma) dgamma[c] += (float)v;
    }
    const float m = (float)(v / count);
    const float a1 = __shfl(m, 0), a2 = __shfl(m, 32);
    const float mu = mean[c], is = invstd[c], sc = scale[c];
    const int sl = seq_len ? seq_len[b] : T;
    const size_t base = ((size_t)b * C + c) * S * T;
    if (VEC) {
        const int Tq = T / 4, n = S * Tq;
        float4* d4 = reinterpret_cast<float4*>(dz + base);
        const float4* x4 = reinterpret_cast<const float4*>(x + base);
        for (int i = lane; i < n; i += 64) {
            const int t = (i % Tq) * 4;
            float4 d = d4[i];
            const float4 xv = x4[i];
            d.x = (t + 0 < sl) ? sc * (d.x - a1 - (xv.x - mu) * is * a2) : 0.f;
            d.y = (t + 1 < sl) ? sc * (d.y - a1 - (xv.y - mu) * is * a2) : 0.f;
            d.z = (t + 2 < sl) ? sc * (d.z - a1 - (xv.z - mu) * is * a2) : 0.f;
            d.w = (t + 3 < sl) ? sc * (d.w - a1 - (xv.w - mu) * is * a2) : 0.f;
            d4[i] = d;
        }
    } else {
        for (int i = lane; i < S * T; i += 64) {
            const int t = i % T;
            dz[base + i] = (t < sl) ? sc * (dz[base + i] - a1 - (x[base + i] - mu) * is * a2) : 0.f;
        }
    }
}

// Masked per-channel sums (sum x, sum x^2 over t < seq_len[b]) of a network INPUT [B, C, S, T] - the batch statistics of a
// first layer that carries its own pre-activation norm (padertorch CNN with input_layer=False; SURVEY.md A.4 variant (i)).
// Everywhere else the statistics come out of the producing convolution's epilogue.
__global__ __launch_bounds__(256) void channel_stats_kernel(const float* __restrict__ x, const int* __restrict__ seq_len,
                                                            double* __restrict__ stats, int C, int S, int T) {
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int sl = seq_len ? min(seq_len[b], T) : T;
    const float* p = x + ((size_t)b * C + c) * S * T;
    float s1 = 0.f, s2 = 0.f;
    for (int i = tid; i < S * T; i += 256) {
        const float v = (i % T) < sl ? p[i] : 0.f;
        s1 += v; s2 = fmaf(v, v, s2);
    }
    __shared__ float red[2][4];
    s1 = wave_sum64(s1); s2 = wave_sum64(s2);
    if ((tid & 63) == 0) { red[0][tid >> 6] = s1; red[1][tid >> 6] = s2; }
    __syncthreads();
    if (tid < 2) {
        const float v = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
        atomicAdd(&stats[((size_t)(b & (PBSED_STAT_SLOTS - 1)) * C + c) * 2 + tid], (double)v);
    }
}

// A norm + ReLU that CLOSES a stack (padertorch pre-activation CNN whose last conv is followed by the stack's final norm and
// activation; SURVEY.md A.4 variant (iii)): there is no consumer conv whose loader could apply it, so it is a launch of its
// own.  Forward: y = mask * relu(x * scale[c] + shift[c]) (the batch statistics behind scale / shift came out of the
// producing conv's epilogue as everywhere else).
__global__ __launch_bounds__(256) void bn_relu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, const int* __restrict__ seq_len,
                                                          float* __restrict__ y, int C, int S, int T, int relu) {
    const int c = blockIdx.x, b = blockIdx.y;
    const int sl = seq_len ? min(seq_len[b], T) : T;
    const size_t base = ((size_t)b * C + c) * S * T;
    const float sc = scale[c], sh = shift[c];
    for (int i = threadIdx.x; i < S * T; i += 256) {
        float z = fmaf(x[base + i], sc, sh);
        if (relu) z = fmaxf(z, 0.f);
        y[base + i] = (i % T) < sl ? z : 0.f;
    }
}

// Backward of the same: dz = dy * mask * relu'(z) written out, and the (sum dz, sum dz * xhat) partial sums BN backward needs
// (the quantities a data-gradient epilogue produces for every other norm) into stats [PBSED_STAT_SLOTS][C][2].
__global__ __launch_bounds__(256) void bn_relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const int* __restrict__ seq_len, float* __restrict__ dz,
                                                          double* __restrict__ stats, int C, int S, int T, int relu) {
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int sl = seq_len ? min(seq_len[b], T) : T;
    const size_t base = ((size_t)b * C + c) * S * T;
    const float sc = scale[c], sh = shift[c], mu = mean[c], is = invstd[c];
    float s1 = 0.f, s2 = 0.f;
    for (int i = tid; i < S * T; i += 256) {
        const float xv = x[base + i];
        const float z = fmaf(xv, sc, sh);
        const bool keep = (i % T) < sl && (!relu || z > 0.f);
        const float g = keep ? dy[base + i] : 0.f;
        dz[base + i] = g;
        s1 += g; s2 = fmaf(g, (xv - mu) * is, s2);
    }
    __shared__ float red[2][4];
    s1 = wave_sum64(s1); s2 = wave_sum64(s2);
    if ((tid & 63) == 0) { red[0][tid >> 6] = s1; red[1][tid >> 6] = s2; }
    __syncthreads();
    if (tid < 2) {
        const float v = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
        atomicAdd(&stats[((size_t)(b & (PBSED_STAT_SLOTS - 1)) * C + c) * 2 + tid], (double)v);
    }
}

// ------------------------------------------------------------------------------------ FBCRNN loss
// One block per (b, k) row.  Fused: squash (eps + (1-2eps) sigmoid), weak fwd/bwd BCE, strong
// (boundary) fwd/bwd BCE with cummax targets, blend, seq-masked time mean, class-weighted
// normalised sum, and the gradient wrt both heads' logits.
struct FbLossArgs {
    const float* lf;       // logits fwd [B,K,T]
    const float* lb;       // logits bwd or null
    const float* weak;     // [B,K]
    const float* bnd;      // [B,K,T] boundary targets or null (slat / weight 0)
    const float* cw;       // [K] class weights or null
    const int* seq_len;    // [B]
    float* yf; float* yb;  // squashed scores out (may be null)
    float* dlf; float* dlb;  // grad wrt logits (may be null -> loss only)
    float* loss;           // scalar accumulator (zeroed by caller)
    int B, K, T, slat;
    float eps, lambda, ls;
    int prob_in;           // 1: lf/lb already hold squashed scores y, gradients are wrt y
    float* summary;        // optional [3*B*K + 1]: weak mask, masked weak targets, clip-level scores, boundary label rate
};

__device__ __forceinline__ float bce_f(float p, float t) {
    return -(t * fmaxf(logf(p), -100.f) + (1.f - t) * fmaxf(logf(1.f - p), -100.f));
}
__device__ __forceinline__ float dbce_f(float p, float t) { return -t / p + (1.f - t) / (1.f - p); }

__global__ __launch_bounds__(256) void fbcrnn_loss_kernel(FbLossArgs a) {
    extern __shared__ float sh[];           // tf[T], tb[T]
    float* tf = sh; float* tb = sh + a.T;
    __shared__ float red[8];
    __shared__ float s_wsum, s_rowok;
    const int row = blockIdx.x, b = row / a.K, k = row % a.K, tid = threadIdx.x;
    const int T = a.T, L = min(a.seq_len[b], T);
    // sum of weights over all (b,k)
    float ws = 0.f;
    for (int i = tid; i < a.B * a.K; i += blockDim.x) {
        const float w = a.weak[i];
        if (w < .01f || w > .99f) ws += a.cw ? a.cw[i % a.K] : 1.f;
    }
    ws = wave_sum64(ws);
    if ((tid & 63) == 0) red[tid >> 6] = ws;
    __syncthreads();
    if (tid == 0) { float s = 0.f; for (int i = 0; i < (int)blockDim.x / 64; ++i) s += red[i]; s_wsum = s; }
    __syncthreads();
    const float wsum = s_wsum;
    float w = a.weak[row];
    const bool wm = (w < .01f) || (w > .99f);
    w = wm ? w : 0.f;
    const float wt = a.ls > 0.f ? fminf(fmaxf(w, a.ls), 1.f - a.ls) : w;
    const float Wrow = wm ? (a.cw ? a.cw[k] : 1.f) : 0.f;
    const float* lf = a.lf + (size_t)row * T;
    const float* lb = a.lb ? a.lb + (size_t)row * T : nullptr;
    const bool strong = a.lambda > 0.f && (a.slat || a.bnd);
    // boundary mask row test over ALL T padded frames + cummax targets
    float cnt = 0.f;
    if (strong) {
        for (int t = tid; t < T; t += blockDim.x) {
            const float be = a.slat ? w : a.bnd[(size_t)row * T + t];
            cnt += ((be > .99f) || (be < .01f)) ? 1.f : 0.f;
            const float bt = a.ls > 0.f ? fminf(fmaxf(be, a.ls), 1.f - a.ls) : be;
            tf[t] = bt; tb[t] = bt;
        }
    }
    __syncthreads();
    cnt = wave_sum64(cnt);
    if ((tid & 63) == 0) red[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0) {
        float s = 0.f; for (int i = 0; i < (int)blockDim.x / 64; ++i) s += red[i];
        s_rowok = (strong && (s / (float)T > .999f) && (w > .99f)) ? 1.f : 0.f;
        if (a.summary) {
            // what CRNN.review hands to the host (reference crnn.py:120-137,155-177): the weak-label mask, the masked
            // weak targets, the clip-level score y_fwd[L-1] (averaged with y_bwd[0] if there is a backward head) and
            // the share of frames whose boundary targets count, summed over rows
            const int BK = a.B * a.K;
            float yw = 0.f;
            if (L > 0) {
                const float sc_ = 1.f - 2.f * a.eps;
                const float vf = a.prob_in ? lf[L - 1] : a.eps + sc_ / (1.f + expf(-lf[L - 1]));
                yw = vf;
                if (lb) {
                    const float vb = a.prob_in ? lb[0] : a.eps + sc_ / (1.f + expf(-lb[0]));
                    yw = vf / 2.f + vb / 2.f;
                }
            }
            a.summary[row] = wm ? 1.f : 0.f;
            a.summary[BK + row] = w;
            a.summary[2 * BK + row] = yw;
            if (s_rowok > 0.f) atomicAdd(&a.summary[3 * BK], s / ((float)BK * (float)T));
        }
        if (strong) for (int t = 1; t < T; ++t) tf[t] = fmaxf(tf[t], tf[t - 1]);
    }
    if (tid == 64 && strong) for (int t = T - 2; t >= 0; --t) tb[t] = fmaxf(tb[t], tb[t + 1]);
    __syncthreads();
    const bool rowok = s_rowok > 0.f;
    const float sc = 1.f - 2.f * a.eps;
    const float coef = (L > 0 && wsum > 0.f) ? Wrow / (wsum * (float)L) : 0.f;
    float yl = 0.f;  // y_fwd at the last valid frame (no-bwd weak term)
    if (!lb && L > 0) yl = a.prob_in ? lf[L - 1] : a.eps + sc / (1.f + expf(-lf[L - 1]));
    float lsum = 0.f, glast = 0.f;
    for (int t = tid; t < T; t += blockDim.x) {
        float sf, yf, sb = 0.f, yb = 0.f;
        if (a.prob_in) {
            yf = lf[t]; sf = 0.f;
            if (lb) yb = lb[t];
        } else {
            sf = 1.f / (1.f + expf(-lf[t]));
            yf = a.eps + sc * sf;
            if (lb) { sb = 1.f / (1.f + expf(-lb[t])); yb = a.eps + sc * sb; }
        }
        const float jf = a.prob_in ? 1.f : sc * sf * (1.f - sf);
        const float jb = a.prob_in ? 1.f : sc * sb * (1.f - sb);
        if (a.yf) a.yf[(size_t)row * T + t] = yf;
        if (a.yb && lb) a.yb[(size_t)row * T + t] = yb;
        float gf = 0.f, gb = 0.f;
        if (t < L) {
            float lam = 0.f;
            if (strong) {
                const float be = a.slat ? w : a.bnd[(size_t)row * T + t];
                lam = (rowok && ((be > .99f) || (be < .01f))) ? a.lambda : 0.f;
            }
            float lw = 0.f;
            if (wm) {
                if (lb) {
                    const float ym = fmaxf(yf, yb);
                    lw = bce_f(ym, wt);
                    const float d = dbce_f(ym, wt) * (1.f - lam);
                    if (yf > yb) gf += d; else if (yb > yf) gb += d; else { gf += .5f * d; gb += .5f * d; }
                } else {
                    lw = bce_f(yl, wt);
                    glast += dbce_f(yl, wt) * (1.f - lam);
                }
            }
            float lsg = 0.f;
            if (lam > 0.f) {
                if (lb) {
                    lsg = .5f * bce_f(yf, tf[t]) + .5f * bce_f(yb, tb[t]);
                    gf += lam * .5f * dbce_f(yf, tf[t]);
                    gb += lam * .5f * dbce_f(yb, tb[t]);
                } else {
                    lsg = bce_f(yf, tf[t]);
                    gf += lam * dbce_f(yf, tf[t]);
                }
            }
            lsum += lam * lsg + (1.f - lam) * lw;
        }
        if (a.dlf) a.dlf[(size_t)row * T + t] = coef * gf * jf;
        if (a.dlb && lb) a.dlb[(size_t)row * T + t] = coef * gb * jb;
    }
    __syncthreads();
    lsum = wave_sum64(lsum);
    glast = wave_sum64(glast);
    if ((tid & 63) == 0) { red[tid >> 6] = lsum; red[4 + (tid >> 6)] = glast; }
    __syncthreads();
    if (tid == 0) {
        float s = 0.f, g = 0.f;
        for (int i = 0; i < (int)blockDim.x / 64; ++i) { s += red[i]; g += red[4 + i]; }
        atomicAdd(a.loss, coef * s);
        if (!lb && a.dlf && L > 0) {
            const float sl_ = 1.f / (1.f + expf(-lf[L - 1]));
            a.dlf[(size_t)row * T + L - 1] += coef * g * (a.prob_in ? 1.f : sc * sl_ * (1.f - sl_));
        }
    }
}

// ------------------------------------------------------------------------------------ BiCRNN loss
struct BiLossArgs {
    const float* logit; const float* st; const int* seq_len;
    float* y; float* dlogit; float* loss; double* msum; int B, K, T; int prob_in;
};
__global__ void bicrnn_mask_count_kernel(BiLossArgs a) {
    const size_t total = (size_t)a.B * a.K * a.T;
    double c = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const float s = a.st[i];
        c += ((s > .99f) || (s < .01f)) ? 1.0 : 0.0;
    }
    c = wave_sum64d(c);
    if ((threadIdx.x & 63) == 0) atomicAdd(a.msum, c);
}
__global__ void bicrnn_loss_kernel(BiLossArgs a) {
    const size_t total = (size_t)a.B * a.K * a.T;
    const float inv = 1.f / (float)(*a.msum);
    float ls = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int t = i % a.T, b = i / ((size_t)a.T * a.K);
        const float s = a.prob_in ? a.logit[i] : 1.f / (1.f + expf(-a.logit[i]));
        if (a.y) a.y[i] = s;
        const float st = a.st[i];
        float g = 0.f;
        if (t < a.seq_len[b] && ((st > .99f) || (st < .01f))) {
            ls += bce_f(s, st);
            g = dbce_f(s, st) * (a.prob_in ? 1.f : s * (1.f - s)) * inv;
        }
        if (a.dlogit) a.dlogit[i] = g;
    }
    ls = wave_sum64(ls);
    if ((threadIdx.x & 63) == 0) atomicAdd(a.loss, ls * inv);
}

// ------------------------------------------------------------------------------------ squash
// y = eps + (1-2eps) sigmoid(x)  (pb_sed/models/weak_label/crnn.py:58-59; eps = 0 -> nn.Sigmoid)
__global__ void squash_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, float eps) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        y[i] = eps + (1.f - 2.f * eps) / (1.f + expf(-x[i]));
}
__global__ void squash_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                  float* __restrict__ dx, size_t n, float eps) {
    const float sc = 1.f - 2.f * eps;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float s = (y[i] - eps) / sc;
        dx[i] = dy[i] * sc * s * (1.f - s);
    }
}

// ------------------------------------------------------------------------------------ optimiser
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, size_t n, double* out) {
    __shared__ double part[4];
    double s = 0;
    const size_t n4 = n / 4;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = g4[i];
        s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const double v = g[n4 * 4 + threadIdx.x]; s += v * v; }
    s = wave_sum64d(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    // one same-address f64 atomic per block (4096 of them used to serialise for ~50 us)
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

// torch.optim.Adam (no amsgrad / weight decay) preceded by clip_grad_norm_(max_norm):
// clip_coef = min(1, max_norm / (norm + 1e-6)); also folds the data-parallel 1/world_size average.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, size_t n,
                                                   float lr, float b1, float b2, float eps, float bc1,
                                                   float bc2_sqrt, float gscale, float max_norm,
                                                   const double* sumsq, float* norm_out,
                                                   const int* __restrict__ skip_flags, int n_flags) {
    // gradients produced behind a raised error flag (a persistent GRU scan that hit its bounded-spin time-out) are
    // never applied: the step becomes a no-op on the device, the host raises when it reads the same flags
    for (int i = 0; i < n_flags; ++i)
        if (skip_flags[i]) return;
    float coef = gscale;
    if (sumsq) {
        const float norm = (float)sqrt(*sumsq) * gscale;
        coef *= fminf(1.f, max_norm / (norm + 1e-6f));
        if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) *norm_out = norm;
    }
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * coef;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
    }
}


// BiCRNN review summary (pb_sed/models/strong_label/crnn.py:114-136): per (clip, class) the share of labelled frames
// inside the sequence (-> strongly labelled clips), the labelled-frame count over ALL T frames (-> strong_label_rate),
// and the segment-wise maxima of scores and targets over eval_segment_length frames ([B, S, K], S = T / L; segments that
// do not fit into seq_len[b] completely hold 0 and are dropped by the host).  One block per (b, k) row.
__global__ __launch_bounds__(256) void bicrnn_review_summary_kernel(const float* __restrict__ y, const float* __restrict__ st,
                                                                    const int* __restrict__ seq_len, float* __restrict__ y_seg,
                                                                    float* __restrict__ t_seg, float* __restrict__ mask_mean,
                                                                    float* __restrict__ mask_cnt, int B, int K, int T, int L) {
    const int row = blockIdx.x, b = row / K, k = row % K;
    const int sl = min(seq_len[b], T), S = T / L, s_valid = sl / L;
    const float* yr = y + (size_t)row * T;
    const float* tr = st + (size_t)row * T;
    float in_seq = 0.f, all = 0.f;
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        const float v = tr[t];
        const float m = (v > .99f || v < .01f) ? 1.f : 0.f;
        all += m;
        if (t < sl) in_seq += m;
    }
    in_seq = wave_sum64(in_seq);
    all = wave_sum64(all);
    __shared__ float red[2][4];
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = in_seq; red[1][threadIdx.x >> 6] = all; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mask_mean[row] = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (float)max(sl, 1);
        mask_cnt[row] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
        float my = 0.f, mt = 0.f;
        if (s < s_valid) {
            my = yr[s * L]; mt = tr[s * L];
            for (int j = 1; j < L; ++j) { my = fmaxf(my, yr[s * L + j]); mt = fmaxf(mt, tr[s * L + j]); }
        }
        y_seg[((size_t)b * S + s) * K + k] = my;
        t_seg[((size_t)b * S + s) * K + k] = mt;
    }
}


// Skip path of a residual connection that crosses a (2,1) frequency pool (the 'deep' net configuration): max over row
// pairs + argmax byte forward; backward routes the gradient to the argmax row and ADDS it to dx (the main path's
// gradient is already there).  x [R, 2, T] rows pairs flattened over (b, c, f/2).
__global__ void pool21_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ idx, size_t n, int T) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / T;
        const int t = i % T;
        const float v0 = x[(2 * row) * T + t], v1 = x[(2 * row + 1) * T + t];
        const bool second = v1 > v0;
        y[i] = second ? v1 : v0;
        idx[i] = (unsigned char)second;
    }
}
__global__ void pool21_bwd_add_kernel(const float* __restrict__ g, const unsigned char* __restrict__ idx, float* __restrict__ dx, size_t n, int T) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / T;
        const int t = i % T;
        dx[(2 * row + idx[i]) * T + t] += g[i];
    }
}
__global__ void add_inplace_kernel(float* __restrict__ a, const float* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] += b[i];
}


// ---- data front-end pieces.
// Scale + superposition mixing (pb_sed/data_preparation/mix.py:67-155, provider.py:195-215): output clip b is the sum of its
// components in list order; a component is a slice of the resident waveform pool, multiplied by its gain in fp32 (numpy:
// float32 array * python float), by the raised-cosine fade in float64 where it does not touch the mixture's edge
// (float32 array *= float64 array) and accumulated in fp32 - bit-exact with the reference on float32 audio.
struct MixComp { long long src_off; int length, start; float gain; int fade_in, fade_out; int pad_; };
__global__ void mix_clips_kernel(const float* __restrict__ pool, const MixComp* __restrict__ comps,
                                 const int* __restrict__ first /*[B+1]*/, float* __restrict__ out, int n_out, int fade_len) {
    const int b = blockIdx.y;
    const int c0 = first[b], c1 = first[b + 1];
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < n_out; n += gridDim.x * blockDim.x) {
        float acc = 0.f;
        for (int c = c0; c < c1; ++c) {
            const MixComp m = comps[c];
            const int i = n - m.start;
            if (i < 0 || i >= m.length) continue;
            float v = pool[m.src_off + i] * m.gain;
            if (fade_len > 0) {
                if (m.fade_in && i < fade_len)          // raised_cos[::-1][i] = raised_cos[fade_len - 1 - i]
                    v = (float)((double)v * (0.5 + cos(M_PI * (double)(fade_len - i) / (double)(fade_len + 1)) / 2.0));
                if (m.fade_out && i >= m.length - fade_len)
                    v = (float)((double)v * (0.5 + cos(M_PI * (double)(i - (m.length - fade_len) + 1) / (double)(fade_len + 1)) / 2.0));
            }
            acc += v;
        }
        out[(size_t)b * n_out + n] = acc;
    }
}

// Target encoding (pb_sed/data_preparation/transform.py:56-124): events (class, start frame, stop frame, label type
// 0 = weak / 1 = boundaries / 2 = strong) of each clip -> weak targets [B,K], boundary targets [B,K,T] (first onset ..
// last offset of the class' boundary / strong events), strong targets [B,K,T] (each strong event); classes that are only
// weakly present get 0.5 where any of their events is active, unlabeled clips 0.5 wherever the target is not 1; frames
// >= seq_len[b] stay 0 (collation padding).  One block per (clip, class).
struct TargetEvent { int cls, start, stop, type; };
__global__ __launch_bounds__(256) void encode_targets_kernel(const TargetEvent* __restrict__ ev, const int* __restrict__ ev_first /*[B+1]*/,
                                                             const int* __restrict__ unlabeled, const int* __restrict__ seq_len,
                                                             float* __restrict__ weak, float* __restrict__ bnd, float* __restrict__ strong,
                                                             int K, int T) {
    const int b = blockIdx.x / K, k = blockIdx.x % K;
    const int e0 = ev_first[b], e1 = ev_first[b + 1], sl = min(seq_len[b], T);
    const bool unl = unlabeled[b] != 0;
    int lo = 0x7fffffff, hi = -1;
    bool present = false;
    for (int e = e0; e < e1; ++e) {
        const TargetEvent v = ev[e];
        if (v.cls != k) continue;
        present = true;
        if (v.type >= 1) { lo = min(lo, v.start); hi = max(hi, v.stop); }
    }
    if (threadIdx.x == 0) {
        const float w = present ? 1.f : 0.f;
        weak[b * K + k] = unl ? w + (1.f - w) * .5f : w;
    }
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        float fb = 0.f, fs = 0.f;
        if (t < sl) {
            bool overall = false, st = false;
            for (int e = e0; e < e1; ++e) {
                const TargetEvent v = ev[e];
                if (v.cls != k || t < v.start || t >= v.stop) continue;
                overall = true;
                st = st || v.type == 2;
            }
            const float half = unl ? .5f : (overall ? .5f : 0.f);
            fb = (t >= lo && t < hi) ? 1.f : 0.f;
            fb += (1.f - fb) * half;
            fs = st ? 1.f : 0.f;
            fs += (1.f - fs) * half;
        }
        if (bnd) bnd[((size_t)b * K + k) * T + t] = fb;
        if (strong) strong[((size_t)b * K + k) * T + t] = fs;
    }
}

}  // namespace pbsed

using namespace pbsed;

static inline int nblocks(size_t n, int bs = 256, int cap = 2048) {
    size_t b = (n + bs - 1) / bs;
    return (int)(b > (size_t)cap ? cap : (b ? b : 1));
}

extern "C" {

void pbsed_conv_pack_dims(int KH, int KW, int Cin, int Cout, int dgrad, int* InP, int* OutP) {
    int ck, ct;
    if (!dgrad) {
        conv_fwd_tile_dims(KH, KW, Cin, Cout, &ck, &ct);
        *InP = (Cin + ck - 1) / ck * ck; *OutP = (Cout + ct - 1) / ct * ct;
    } else {
        conv_fwd_tile_dims(KH, KW, Cout, Cin, &ck, &ct);
        *InP = (Cout + ck - 1) / ck * ck; *OutP = (Cin + ct - 1) / ct * ct;
    }
}

int pbsed_pack_conv_weights(const float* w, float* wp, int Cout, int Cin, int KH, int KW, int dgrad,
                            void* stream) {
    int InP, OutP;
    pbsed_conv_pack_dims(KH, KW, Cin, Cout, dgrad, &InP, &OutP);
    const size_t total = (size_t)KH * KW * InP * OutP;
    hipLaunchKernelGGL(pack_conv_weights_kernel, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream,
                       w, wp, Cout, Cin, KH, KW, InP, OutP, dgrad);
    return check_launch("pack_conv_weights");
}

int pbsed_pack_conv_weights_batched(const void* descs, int n, void* stream) {
    if (n < 1) return PBSED_OK;
    hipLaunchKernelGGL(pack_batched_kernel, dim3(128, n), dim3(256), 0, (hipStream_t)stream, (const PackDesc*)descs);
    return check_launch("pack_conv_weights_batched");
}

int pbsed_bn_finalize(const double* sums, double count, const float* gamma, const float* beta, float eps,
                      float momentum, float* running_mean, float* running_power, float* mean,
                      float* invstd, float* scale, float* shift, int C, void* stream) {
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C * 32 + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums,
                       count, gamma, beta, eps, momentum, running_mean, running_power, mean, invstd, scale,
                       shift, C);
    return check_launch("bn_finalize");
}

int pbsed_bn_eval_params(const float* gamma, const float* beta, float eps, const float* running_mean,
                         const float* running_power, float* mean, float* invstd, float* scale,
                         float* shift, int C, void* stream) {
    hipLaunchKernelGGL(bn_eval_params_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma,
                       beta, eps, running_mean, running_power, mean, invstd, scale, shift, C);
    return check_launch("bn_eval_params");
}

int pbsed_bn_bwd_finalize(const double* sums, double count, float* dgamma, float* dbeta, float* m1,
                          float* m2, int C, void* stream) {
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums,
                       count, dgamma, dbeta, m1, m2, C);
    return check_launch("bn_bwd_finalize");
}

int pbsed_bn_bwd_coef(const double* sums, double count, const float* mean, const float* invstd, const float* scale,
                      float* dgamma, float* dbeta, float* coef, int C, void* stream) {
    hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, count, mean, invstd,
                       scale, dgamma, dbeta, coef, C);
    return check_launch("bn_bwd_coef");
}

int pbsed_bn_bwd_apply(float* dz, const float* x, const float* mean, const float* invstd, const float* scale,
                       const float* m1, const float* m2, const int* seq_len, int B, int C, int S, int T,
                       void* stream) {
    const size_t total = (size_t)B * C * S * T;
    if (T % 4 == 0)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(nblocks(total / 4, 256, 8192)), dim3(256), 0,
                           (hipStream_t)stream, dz, x, mean, invstd, scale, m1, m2, seq_len, B, C, S, T);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(nblocks(total, 256, 8192)), dim3(256), 0,
                           (hipStream_t)stream, dz, x, mean, invstd, scale, m1, m2, seq_len, B, C, S, T);
    return check_launch("bn_bwd_apply");
}

int pbsed_bn_bwd(float* dz, const float* x, const double* sums, double count, const float* mean, const float* invstd,
                 const float* scale, float* dgamma, float* dbeta, const int* seq_len, int B, int C, int S, int T,
                 void* stream) {
    static_assert(PBSED_STAT_SLOTS <= 32, "one wave half sums the slots");
    if (B > 65535) { set_error("bn_bwd: batch %d over the grid limit", B); return PBSED_E_ARG; }
    if ((size_t)S * T < 2048) {                       // short slabs: a wave per (clip, channel), four channels per block
        if (T % 4 == 0)
            hipLaunchKernelGGL(bn_bwd_fused_wave_kernel<true>, dim3((C + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, dz, x, sums,
                               count, mean, invstd, scale, dgamma, dbeta, seq_len, C, S, T);
        else
            hipLaunchKernelGGL(bn_bwd_fused_wave_kernel<false>, dim3((C + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, dz, x, sums,
                               count, mean, invstd, scale, dgamma, dbeta, seq_len, C, S, T);
        return check_launch("bn_bwd");
    }
    if (T % 4 == 0)
        hipLaunchKernelGGL(bn_bwd_fused_kernel<true>, dim3(C, B), dim3(256), 0, (hipStream_t)stream, dz, x, sums, count, mean,
                           invstd, scale, dgamma, dbeta, seq_len, C, S, T);
    else
        hipLaunchKernelGGL(bn_bwd_fused_kernel<false>, dim3(C, B), dim3(256), 0, (hipStream_t)stream, dz, x, sums, count, mean,
                           invstd, scale, dgamma, dbeta, seq_len, C, S, T);
    return check_launch("bn_bwd");
}

int pbsed_channel_stats(const float* x, const int* seq_len, double* stats, int B, int C, int S, int T, void* stream) {
    if (B < 1 || B > 65535 || C < 1) { set_error("channel_stats: bad B=%d C=%d", B, C); return PBSED_E_ARG; }
    hipLaunchKernelGGL(channel_stats_kernel, dim3(C, B), dim3(256), 0, (hipStream_t)stream, x, seq_len, stats, C, S, T);
    return check_launch("channel_stats");
}

int pbsed_bn_relu_fwd(const float* x, const float* scale, const float* shift, const int* seq_len, float* y, int relu, int B,
                      int C, int S, int T, void* stream) {
    if (B < 1 || B > 65535 || C < 1) { set_error("bn_relu_fwd: bad B=%d C=%d", B, C); return PBSED_E_ARG; }
    hipLaunchKernelGGL(bn_relu_fwd_kernel, dim3(C, B), dim3(256), 0, (hipStream_t)stream, x, scale, shift, seq_len, y, C, S, T, relu);
    return check_launch("bn_relu_fwd");
}

int pbsed_bn_relu_bwd(const float* dy, const float* x, const float* scale, const float* shift, const float* mean,
                      const float* invstd, const int* seq_len, float* dz, double* stats, int relu, int B, int C, int S, int T,
                      void* stream) {
    if (B < 1 || B > 65535 || C < 1) { set_error("bn_relu_bwd: bad B=%d C=%d", B, C); return PBSED_E_ARG; }
    hipLaunchKernelGGL(bn_relu_bwd_kernel, dim3(C, B), dim3(256), 0, (hipStream_t)stream, dy, x, scale, shift, mean, invstd,
                       seq_len, dz, stats, C, S, T, relu);
    return check_launch("bn_relu_bwd");
}

int pbsed_fbcrnn_loss(const float* logit_fwd, const float* logit_bwd, const float* weak_targets,
                      const float* boundary_targets, const float* class_weights, const int* seq_len,
                      float* y_fwd, float* y_bwd, float* dlogit_fwd, float* dlogit_bwd, float* loss,
                      int B, int K, int T, float minimum_score, float strong_weight, int slat,
                      float label_smoothing, int inputs_are_scores, float* summary, void* stream) {
    FbLossArgs a{logit_fwd, logit_bwd, weak_targets, boundary_targets, class_weights, seq_len, y_fwd, y_bwd,
                 dlogit_fwd, dlogit_bwd, loss, B, K, T, slat, minimum_score, strong_weight, label_smoothing,
                 inputs_are_scores, summary};
    PBSED_HIP_TRY(hipMemsetAsync(loss, 0, sizeof(float), (hipStream_t)stream), "hipMemsetAsync");
    if (summary) PBSED_HIP_TRY(hipMemsetAsync(summary + (size_t)3 * B * K, 0, sizeof(float), (hipStream_t)stream), "hipMemsetAsync");
    hipLaunchKernelGGL(fbcrnn_loss_kernel, dim3(B * K), dim3(256), 2 * T * sizeof(float), (hipStream_t)stream, a);
    return check_launch("fbcrnn_loss");
}

int pbsed_bicrnn_loss(const float* logit, const float* strong_targets, const int* seq_len, float* y,
                      float* dlogit, float* loss, double* scratch, int B, int K, int T,
                      int inputs_are_scores, void* stream) {
    BiLossArgs a{logit, strong_targets, seq_len, y, dlogit, loss, scratch, B, K, T, inputs_are_scores};
    PBSED_HIP_TRY(hipMemsetAsync(loss, 0, sizeof(float), (hipStream_t)stream), "hipMemsetAsync");
    PBSED_HIP_TRY(hipMemsetAsync(scratch, 0, sizeof(double), (hipStream_t)stream), "hipMemsetAsync");
    const size_t total = (size_t)B * K * T;
    hipLaunchKernelGGL(bicrnn_mask_count_kernel, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, a);
    hipLaunchKernelGGL(bicrnn_loss_kernel, dim3(nblocks(total)), dim3(256), 0, (hipStream_t)stream, a);
    return check_launch("bicrnn_loss");
}

int pbsed_bicrnn_review_summary(const float* y, const float* strong_targets, const int* seq_len, float* y_seg, float* t_seg,
                                float* mask_mean, float* mask_cnt, int B, int K, int T, int segment_length, void* stream) {
    if (segment_length < 1 || segment_length > T) { set_error("bicrnn_review_summary: segment length %d for T=%d", segment_length, T); return PBSED_E_ARG; }
    hipLaunchKernelGGL(bicrnn_review_summary_kernel, dim3(B * K), dim3(256), 0, (hipStream_t)stream, y, strong_targets, seq_len,
                       y_seg, t_seg, mask_mean, mask_cnt, B, K, T, segment_length);
    return check_launch("bicrnn_review_summary");
}

int pbsed_pool21_fwd(const float* x, float* y, unsigned char* idx, size_t n_out, int T, void* stream) {
    hipLaunchKernelGGL(pool21_fwd_kernel, dim3(nblocks(n_out)), dim3(256), 0, (hipStream_t)stream, x, y, idx, n_out, T);
    return check_launch("pool21_fwd");
}

int pbsed_pool21_bwd_add(const float* g, const unsigned char* idx, float* dx, size_t n_out, int T, void* stream) {
    hipLaunchKernelGGL(pool21_bwd_add_kernel, dim3(nblocks(n_out)), dim3(256), 0, (hipStream_t)stream, g, idx, dx, n_out, T);
    return check_launch("pool21_bwd_add");
}

int pbsed_add_inplace(float* a, const float* b, size_t n, void* stream) {
    hipLaunchKernelGGL(add_inplace_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, n);
    return check_launch("add_inplace");
}

int pbsed_mix_clips(const float* pool, const void* comps, const int* first, float* out, int B, int n_out, int fade_len,
                    void* stream) {
    if (B < 1 || n_out < 1 || fade_len < 0) { set_error("mix_clips: bad B=%d n_out=%d fade=%d", B, n_out, fade_len); return PBSED_E_ARG; }
    const int gx = (n_out + 255) / 256 < 1024 ? (n_out + 255) / 256 : 1024;
    hipLaunchKernelGGL(mix_clips_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, pool, (const MixComp*)comps, first, out,
                       n_out, fade_len);
    return check_launch("mix_clips");
}

int pbsed_encode_targets(const void* events, const int* ev_first, const int* unlabeled, const int* seq_len, float* weak,
                         float* boundary, float* strong, int B, int K, int T, void* stream) {
    if (B < 1 || K < 1 || T < 1 || !weak) { set_error("encode_targets: bad B=%d K=%d T=%d", B, K, T); return PBSED_E_ARG; }
    hipLaunchKernelGGL(encode_targets_kernel, dim3(B * K), dim3(256), 0, (hipStream_t)stream, (const TargetEvent*)events, ev_first,
                       unlabeled, seq_len, weak, boundary, strong, K, T);
    return check_launch("encode_targets");
}

int pbsed_squash_fwd(const float* x, float* y, size_t n, float eps, void* stream) {
    hipLaunchKernelGGL(squash_fwd_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, eps);
    return check_launch("squash_fwd");
}

int pbsed_squash_bwd(const float* y, const float* dy, float* dx, size_t n, float eps, void* stream) {
    hipLaunchKernelGGL(squash_bwd_kernel, dim3(nblocks(n)), dim3(256), 0, (hipStream_t)stream, y, dy, dx, n, eps);
    return check_launch("squash_bwd");
}

int pbsed_grad_sumsq(const float* g, size_t n, double* out, void* stream) {
    PBSED_HIP_TRY(hipMemsetAsync(out, 0, sizeof(double), (hipStream_t)stream), "hipMemsetAsync");
    hipLaunchKernelGGL(sumsq_kernel, dim3(nblocks(n / 4 + 1, 256, 256)), dim3(256), 0, (hipStream_t)stream, g, n, out);
    return check_launch("grad_sumsq");
}

int pbsed_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1,
                    float beta2, float eps, int step, float grad_scale, float max_norm,
                    const double* sumsq, float* norm_out, const int* skip_flags, int n_flags, void* stream) {
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adam_kernel, dim3(nblocks(n, 256, 2048)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                       n, lr, beta1, beta2, eps, bc1, bc2s, grad_scale, max_norm, sumsq, norm_out, skip_flags, skip_flags ? n_flags : 0);
    return check_launch("adam_step");
}

}  // extern "C"
