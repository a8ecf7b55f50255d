// Shared epilogue of the implicit-GEMM conv kernels (fp32-MFMA conv.hip and bf16-MFMA conv_bf16.hip): the D
// fragment layout is identical (col = lane&15 -> 16 consecutive t, row = (lane>>4)*4 + reg -> cout).
// bias, (2,1) max-pool + argmax byte, masked statistics of the produced tensor, store; DGRAD: backward through
// mask -> ReLU -> BN-apply of the layer's prologue with the sums BN backward needs.
#pragma once
#include "common.h"
#include "pbsed_internal.h"

namespace pbsed {

// st_s: [WN][COUT_T][FO_T][2] floats, WN = waves along t; every (wave column, channel, row) entry is written by exactly
// one lane and the columns are summed in a fixed order, so a block's partial statistics do not depend on wave timing.
template <int COUT_T, int FT, int TT, int MTW, int NTT, int WN, bool POOL, bool DGRAD>
__device__ __forceinline__ void conv_epilogue(const ConvFwdArgs& a, f32x4 (&acc)[MTW][FT * NTT], float* st_s, int b,
                                              int f0, int t0, int cout0, int sl, int wm, int wn, int lq, int lr,
                                              int tid) {
    constexpr int FO_T = POOL ? FT / 2 : FT;
    // ---- epilogue: bias, (2,1) max-pool, BN statistics of the produced tensor, store
    const int Fo = POOL ? a.F / 2 : a.F;
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cl = (wm * MTW + m) * 16 + lq * 4 + r;
            const int cout = cout0 + cl;
            const float bias = (a.bias && cout < a.Cout) ? a.bias[cout] : 0.f;
            const bool cv = cout < a.Cout;
            const bool bnb = DGRAD && a.bx != nullptr;
            float bsc = 0.f, bsh = 0.f, bmu = 0.f, bis = 0.f;
            if (bnb && cv) { bsc = a.bscale[cout]; bsh = a.bshift[cout]; bmu = a.bmean[cout]; bis = a.binvstd[cout]; }
            // DGRAD: fetch the layer's raw forward input for the whole row group first (independent loads)
            float xin[FO_T][NTT];
            if (DGRAD) {
#pragma unroll
                for (int fo_l = 0; fo_l < FO_T; ++fo_l)
#pragma unroll
                    for (int j = 0; j < NTT; ++j) {
                        const int t = t0 + (wn * NTT + j) * 16 + lr, fo = f0 + fo_l;
                        xin[fo_l][j] = (bnb && cv && fo < Fo && t < a.T)
                                           ? a.bx[((size_t)(b * a.Cout + cout) * Fo + fo) * a.T + t] : 0.f;
                    }
            }
            float c1 = 0.f, c2 = 0.f;               // per-channel statistics: the rows are summed before the lane reduction
#pragma unroll
            for (int fo_l = 0; fo_l < FO_T; ++fo_l) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int j = 0; j < NTT; ++j) {
                    const int t = t0 + (wn * NTT + j) * 16 + lr;
                    float v;
                    int pidx = 0;
                    if (POOL) {
                        const float v0 = acc[m][(2 * fo_l) * NTT + j][r];
                        const float v1 = acc[m][(2 * fo_l + 1) * NTT + j][r];
                        pidx = v1 > v0;
                        v = (pidx ? v1 : v0) + bias;
                    } else {
                        v = acc[m][fo_l * NTT + j][r] + bias;
                    }
                    const int fo = (POOL ? f0 / 2 : f0) + fo_l;
                    if (cv && fo < Fo && t < a.T) {
                        const size_t o = ((size_t)(b * a.Cout + cout) * Fo + fo) * a.T + t;
                        if (DGRAD) {
                            if (bnb) {
                                // backward through mask -> ReLU -> BN-apply of the layer's prologue:
                                // dz = da * [z > 0] * [t < seq_len];  partial sums for BN backward
                                const float xv = xin[fo_l][j];
                                const float z = fmaf(xv, bsc, bsh);
                                const bool keep = (t < sl) && (!a.relu || z > 0.f);
                                v = keep ? v : 0.f;
                                s1 += v; s2 += v * ((xv - bmu) * bis);
                            }
                            a.y[o] = v;
                        } else {
                            a.y[o] = v;
                            if (POOL && a.pool_idx) a.pool_idx[o] = (uint8_t)pidx;
                            if (t < sl) { s1 += v; s2 += v * v; }
                        }
                    }
                }
                if (a.stats && a.stats_cf) {
                    s1 = wave_sum16(s1);
                    s2 = wave_sum16(s2);
                    if (lr == 0) {
                        st_s[((wn * COUT_T + cl) * FO_T + fo_l) * 2 + 0] = s1;
                        st_s[((wn * COUT_T + cl) * FO_T + fo_l) * 2 + 1] = s2;
                    }
                }
                c1 += s1; c2 += s2;
            }
            if (a.stats && !a.stats_cf) {
                c1 = wave_sum16(c1);
                c2 = wave_sum16(c2);
                if (lr == 0) {
                    st_s[(wn * COUT_T + cl) * FO_T * 2 + 0] = c1;
                    st_s[(wn * COUT_T + cl) * FO_T * 2 + 1] = c2;
                }
            }
        }
    }
    if (a.stats) {
        __syncthreads();
        const int Fo_ = POOL ? a.F / 2 : a.F;
        // PBSED_STAT_SLOTS copies of the accumulators spread same-address atomic contention
        const int slot = blockIdx.x & (PBSED_STAT_SLOTS - 1);
        if (a.stats_cf) {
            for (int i = tid; i < COUT_T * FO_T * 2; i += 256) {
                const int which = i & 1, fo_l = (i >> 1) % FO_T, cl = (i >> 1) / FO_T;
                const int cout = cout0 + cl, fo = (POOL ? f0 / 2 : f0) + fo_l;
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < WN; ++w) v += st_s[w * COUT_T * FO_T * 2 + i];
                if (cout < a.Cout && fo < Fo_)
                    atomicAdd(&a.stats[((size_t)slot * a.Cout * Fo_ + cout * Fo_ + fo) * 2 + which], (double)v);
            }
        } else {
            // per-channel statistics: the block's rows are summed first, one atomic per (channel, moment)
            for (int i = tid; i < COUT_T * 2; i += 256) {
                const int which = i & 1, cl = i >> 1, cout = cout0 + cl;
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < WN; ++w) v += st_s[(w * COUT_T + cl) * FO_T * 2 + which];
                if (cout < a.Cout) atomicAdd(&a.stats[((size_t)slot * a.Cout + cout) * 2 + which], (double)v);
            }
        }
    }
}

}  // namespace pbsed
