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

}  // namespace pbsed
