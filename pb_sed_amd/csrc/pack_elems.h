// Element functions of the conv weight packers (direct [kh*kw][InP][OutP] and Winograd-F(4,3) [3][6][InP][OutP]
// layouts), shared by the per-layer kernels and the batched one that refreshes every packed copy of a model in one launch.
#pragma once
#include "common.h"

namespace pbsed {

__device__ __forceinline__ float pack_direct_elem(const float* __restrict__ w, size_t i, int Cout, int Cin, int KK, int InP,
                                                  int OutP, int dgrad) {
    const int o = i % OutP, ii = (i / OutP) % InP, kk = i / ((size_t)OutP * InP);
    if (!dgrad) return (o < Cout && ii < Cin) ? w[((size_t)o * Cin + ii) * KK + kk] : 0.f;
    // kernel-input channel ii = layer cout, kernel-output channel o = layer cin, taps flipped
    return (o < Cin && ii < Cout) ? w[((size_t)ii * Cin + o) * KK + (KK - 1 - kk)] : 0.f;
}

// U[kh][xi][InP][OutP] = sum_kw G[xi][kw] * g[kh][kw]; dgrad: g = flipped kernel with in/out channels swapped.
__device__ __forceinline__ float pack_wino_elem(const float* __restrict__ w, size_t i, int Cout, int Cin, int InP, int OutP,
                                                int dgrad) {
    const int o = i % OutP, ii = (i / OutP) % InP, kx = i / ((size_t)OutP * InP);
    const int kh = kx / 6, xi = kx % 6;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (!dgrad) {
        if (o < Cout && ii < Cin) {
            const float* p = w + ((size_t)o * Cin + ii) * 9 + kh * 3;
            g0 = p[0]; g1 = p[1]; g2 = p[2];
        }
    } else if (o < Cin && ii < Cout) {
        const float* p = w + ((size_t)ii * Cin + o) * 9 + (2 - kh) * 3;
        g0 = p[2]; g1 = p[1]; g2 = p[0];
    }
    switch (xi) {
        case 0: return .25f * g0;
        case 1: return -(g0 + g1 + g2) * (1.f / 6.f);
        case 2: return (-g0 + g1 - g2) * (1.f / 6.f);
        case 3: return g0 * (1.f / 24.f) + g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
        case 4: return g0 * (1.f / 24.f) - g1 * (1.f / 12.f) + g2 * (1.f / 6.f);
        default: return g2;
    }
}

// Exact three-way bf16 split by truncation (hi + mid + lo == u); returns the parts as halfwords.
__device__ __forceinline__ void split3_halfs(float u, unsigned short& hi, unsigned short& mid, unsigned short& lo) {
    const unsigned b0 = __float_as_uint(u);
    hi = (unsigned short)(b0 >> 16);
    const float r1 = u - __uint_as_float(b0 & 0xffff0000u);
    const unsigned b1 = __float_as_uint(r1);
    mid = (unsigned short)(b1 >> 16);
    const float r2 = r1 - __uint_as_float(b1 & 0xffff0000u);
    lo = (unsigned short)(__float_as_uint(r2) >> 16);
}

// bf16x3 Winograd layout (csrc/conv_winox3.hip): halfwords [InP/32][xi 6][kh 3][OutP/16][part 3][lane 64][8] - the 16 bytes
// a lane holds of an MFMA A fragment (row = cout tile * 16 + (lane & 15), k = chunk * 32 + (lane >> 4) * 8 + j) contiguous,
// one 1 KB piece per (point, kernel row, cout tile, part).  U is formed in fp32 exactly as pack_wino_elem does and split by
// truncation into three bf16 parts (hi + mid + lo == U exactly).  One call = one weight: v = index over
// [InP/32][xi][kh][OutP/16][lane][8]; writes its three parts (1 KB apart).
__device__ __forceinline__ void pack_winox3_value(const float* __restrict__ w, unsigned short* __restrict__ up, size_t v, int Cout, int Cin,
                                                  int InP, int OutP, int dgrad) {
    const int j = (int)(v & 7), lane = (int)((v >> 3) & 63);
    size_t r = v >> 9;
    const size_t piece = r;                       // (chunk, xi, kh, cout tile): three 1 KB parts each
    const int MT = OutP / 16;
    const int mt = (int)(r % MT); r /= MT;
    const int kh = (int)(r % 3); r /= 3;
    const int xi = (int)(r % 6);
    const int c = (int)(r / 6);
    const int o = mt * 16 + (lane & 15), ii = c * 32 + (lane >> 4) * 8 + j;
    const float u = pack_wino_elem(w, ((size_t)(kh * 6 + xi) * InP + ii) * OutP + o, Cout, Cin, InP, OutP, dgrad);
    unsigned short h, m, l;
    split3_halfs(u, h, m, l);
    unsigned short* dst = up + piece * 1536 + (size_t)lane * 8 + j;
    dst[0] = h; dst[512] = m; dst[1024] = l;
}

// Conv1d bf16x3 layout (csrc/conv1d_pc.hip): halfwords [InP/32][kw][OutP/16][part 3][lane 64][8], same fragment order; split by
// truncation (hi + mid + lo == w exactly).  dgrad: in / out channels swapped, taps flipped.  v = index over
// [InP/32][kw][OutP/16][lane][8].
__device__ __forceinline__ void pack_c1x3_value(const float* __restrict__ w, unsigned short* __restrict__ up, size_t v, int Cout, int Cin,
                                                int KW, int InP, int OutP, int dgrad) {
    const int j = (int)(v & 7), lane = (int)((v >> 3) & 63);
    size_t r = v >> 9;
    const size_t piece = r;
    const int MT = OutP / 16;
    const int mt = (int)(r % MT); r /= MT;
    const int kw = (int)(r % KW);
    const int c = (int)(r / KW);
    const int o = mt * 16 + (lane & 15), ii = c * 32 + (lane >> 4) * 8 + j;
    float u = 0.f;
    if (!dgrad) {
        if (o < Cout && ii < Cin) u = w[((size_t)o * Cin + ii) * KW + kw];
    } else if (o < Cin && ii < Cout) {
        u = w[((size_t)ii * Cin + o) * KW + (KW - 1 - kw)];
    }
    unsigned short h, m, l;
    split3_halfs(u, h, m, l);
    unsigned short* dst = up + piece * 1536 + (size_t)lane * 8 + j;
    dst[0] = h; dst[512] = m; dst[1024] = l;
}

// Few-channel 3x3 bf16x3 layout (csrc/conv_s16.hip): halfwords [step 5][part 3][OutP/16][lane 64][8] - the A fragments of the five
// tap-pair steps: row = cout tile * 16 + (lane & 15), k = (lane >> 4) * 8 + j = tap 2 s + (lane >> 5), channel ((lane >> 4) & 1) * 8
// + j; the tenth tap and padded channels are zero; split by truncation (hi + mid + lo == w exactly).  dgrad: in / out channels
// swapped, taps flipped.  v = index over [step][OutP/16][lane][8].
__device__ __forceinline__ void pack_s16_value(const float* __restrict__ w, unsigned short* __restrict__ up, size_t v, int Cout, int Cin,
                                               int OutP, int dgrad) {
    const int j = (int)(v & 7), lane = (int)((v >> 3) & 63);
    const size_t r = v >> 9;
    const int MT = OutP / 16;
    const int mt = (int)(r % MT), s = (int)(r / MT);
    const int kq = lane >> 4, tap = 2 * s + (kq >> 1), ch = (kq & 1) * 8 + j, o = mt * 16 + (lane & 15);
    float u = 0.f;
    if (tap < 9) {
        if (!dgrad) {
            if (o < Cout && ch < Cin) u = w[((size_t)o * Cin + ch) * 9 + tap];
        } else if (o < Cin && ch < Cout) {
            u = w[((size_t)ch * Cin + o) * 9 + (8 - tap)];
        }
    }
    unsigned short h, m, l;
    split3_halfs(u, h, m, l);
    unsigned short* dst = up + ((size_t)(s * 3) * MT + mt) * 512 + (size_t)lane * 8 + j;
    dst[0] = h; dst[(size_t)MT * 512] = m; dst[(size_t)2 * MT * 512] = l;
}

}  // namespace pbsed
