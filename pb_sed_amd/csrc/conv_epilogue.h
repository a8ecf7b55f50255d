// Shared epilogue of the implicit-GEMM conv kernels (fp32-MFMA conv.hip and bf16-MFMA conv_bf16.hip): the D
// fragment layout is identical (col = lane&15 -> 16 consecutive t, row = (lane>>4)*4 + reg -> cout).
// bias, (2,1) max-pool + argmax byte, masked statistics of the produced tensor, store; DGRAD: backward through
// mask -> ReLU -> BN-apply of the layer's prologue with the sums BN backward needs.
//
// Stores / the DGRAD re-load of the layer input are raw buffer accesses on clip-relative resources: one 32-bit
// byte offset = channel part + row part + column part, each of which is 2^31 / 2^29 when its index is out of range,
// so invalid elements drop out without branches or 64-bit address arithmetic (VALU instructions are paid in fp32-MFMA
// time on gfx950; the old per-element epilogue cost as much as the MFMAs of a 16-channel layer).
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
    constexpr unsigned OOB_C = 0x80000000u, OOB_T = 0x20000000u;     // any sum of the three parts stays below 2^32; clips are below 2^29 bytes
    const int Fo = POOL ? a.F / 2 : a.F;
    const int fo0 = POOL ? f0 / 2 : f0;
    const unsigned clip = (unsigned)(a.Cout * Fo * a.T);              // elements of y (and of the layer input bx) per clip
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)b * clip, 0, clip * 4u, 0x00020000);
    const bool want_idx = POOL && a.pool_idx != nullptr;
    const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(
        want_idx ? a.pool_idx + (size_t)b * clip : nullptr, 0, want_idx ? clip : 0u, 0x00020000);
    const bool has_res = !DGRAD && a.res != nullptr;              // residual connection ending at this layer's output
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(
        has_res ? const_cast<float*>(a.res) + (size_t)b * clip : nullptr, 0, has_res ? clip * 4u : 0u, 0x00020000);
    const bool bnb = DGRAD && a.bx != nullptr;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        bnb ? const_cast<float*>(a.bx) + (size_t)b * clip : nullptr, 0, bnb ? clip * 4u : 0u, 0x00020000);

    // column part of the offsets (per lane, shared by all channels and rows) and the sequence mask
    unsigned toff[NTT];
    bool in_seq[NTT];
#pragma unroll
    for (int j = 0; j < NTT; ++j) {
        const int t = t0 + (wn * NTT + j) * 16 + lr;
        toff[j] = t < a.T ? (unsigned)(fo0 * a.T + t) * 4u : OOB_T;
        in_seq[j] = t < sl;
    }
    const unsigned row_step = (unsigned)a.T * 4u;
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cl = (wm * MTW + m) * 16 + lq * 4 + r;
            const int cout = cout0 + cl;
            const bool cv = cout < a.Cout;
            const float bias = (a.bias && cv) ? a.bias[cout] : 0.f;
            const unsigned coff = cv ? (unsigned)(cout * Fo * a.T) * 4u : OOB_C;
            float bsc = 0.f, bsh = 0.f, bmu = 0.f, bis = 0.f;
            if (bnb && cv) { bsc = a.bscale[cout]; bsh = a.bshift[cout]; bmu = a.bmean[cout]; bis = a.binvstd[cout]; }
            // DGRAD: fetch the layer's raw forward input for the whole row group first (independent loads)
            float xin[FO_T][NTT];
            if (DGRAD && bnb) {
#pragma unroll
                for (int fo_l = 0; fo_l < FO_T; ++fo_l)
#pragma unroll
                    for (int j = 0; j < NTT; ++j) {
                        const unsigned off = coff + toff[j] + (fo0 + fo_l < Fo ? fo_l * row_step : OOB_T);
                        xin[fo_l][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_x, off, 0, 0));
                    }
            }
            float c1 = 0.f, c2 = 0.f;               // per-channel statistics: the rows are summed before the lane reduction
#pragma unroll
            for (int fo_l = 0; fo_l < FO_T; ++fo_l) {
                const bool row_ok = fo0 + fo_l < Fo;                  // uniform
                const unsigned roff = row_ok ? fo_l * row_step : OOB_T;
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int j = 0; j < NTT; ++j) {
                    const unsigned off = coff + toff[j] + roff;
                    const bool counted = in_seq[j] && row_ok;         // padded channels produce exact zeros
                    float v;
                    if (POOL) {
                        const float v0 = acc[m][(2 * fo_l) * NTT + j][r];
                        const float v1 = acc[m][(2 * fo_l + 1) * NTT + j][r];
                        const bool second = v1 > v0;
                        v = (second ? v1 : v0) + bias;
                        if (want_idx) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)second, rs_p, off >> 2, 0, 0);
                    } else {
                        v = acc[m][fo_l * NTT + j][r] + bias;
                    }
                    if (has_res) v += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_r, off, 0, 0));   // out of range: + 0
                    if (DGRAD) {
                        if (bnb) {
                            // backward through mask -> ReLU -> BN-apply of the layer's prologue:
                            // dz = da * [z > 0] * [t < seq_len];  partial sums for BN backward
                            const float xv = xin[fo_l][j];
                            const float z = fmaf(xv, bsc, bsh);
                            const bool keep = counted && (!a.relu || z > 0.f);
                            v = keep ? v : 0.f;
                            s1 += v; s2 = fmaf(v, (xv - bmu) * bis, s2);
                        }
                    } else {
                        const float vm = counted ? v : 0.f;
                        s1 += vm; s2 = fmaf(vm, vm, s2);
                    }
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs_y, off, 0, 0);
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
        // PBSED_STAT_SLOTS copies of the accumulators spread same-address atomic contention
        const int slot = blockIdx.x & (PBSED_STAT_SLOTS - 1);
        if (a.stats_cf) {
            for (int i = tid; i < COUT_T * FO_T * 2; i += 256) {
                const int which = i & 1, fo_l = (i >> 1) % FO_T, cl = (i >> 1) / FO_T;
                const int cout = cout0 + cl, fo = fo0 + fo_l;
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < WN; ++w) v += st_s[w * COUT_T * FO_T * 2 + i];
                if (cout < a.Cout && fo < Fo)
                    atomicAdd(&a.stats[((size_t)slot * a.Cout * Fo + cout * Fo + fo) * 2 + which], (double)v);
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
