// Implicit-GEMM convolution forward on fp32 MFMA (v_mfma_f32_16x16x4_f32) for gfx950.
//
// One kernel family serves every contraction of the FBCRNN/BiCRNN forward path:
//   CNN2d 3x3 (+BN/ReLU prologue, +bias, +(2,1) max-pool epilogue, +BN-statistics epilogue)
//   CNN1d k=1 / k=3, GRU input projections (k=1), per-frame heads (k=1).
// Reference op sites: pb_sed/models/weak_label/crnn.py:93 (self.cnn), :61-67 (rnn + output_net);
// layer list pb_sed/experiments/weak_label_crnn/training.py:159-169,218-260.
//
// Mapping (DESIGN.md "conv_fwd"): GEMM M = Cout (MFMA A = weights), N = spatial (MFMA B = input
// patch, 16 consecutive t per tile), K = (kh,kw,cin).  Spatial on N makes the D fragment
// t-contiguous across lanes (coalesced stores in the reference's [B,C,F,T] layout) and puts the
// two frequency rows of a (2,1) pool window into the same lane.  No im2col is materialised: the
// input halo tile [CK][FT+KH-1][TT+KW-1] is staged once in LDS (prologue applied while staging).
#include <cstdlib>

#include "common.h"
#include "conv_epilogue.h"
#include "pbsed_internal.h"

namespace pbsed {

constexpr int pad16mod32(int n) { return ((n + 15) / 32) * 32 + 16; }

template <int COUT_T, int FT, int TT, int KH, int KW, int CK, bool POOL>
struct ConvFwdCfg {
    static constexpr int WM = (COUT_T >= 32) ? 2 : 1;
    static constexpr int WN = 4 / WM;
    static constexpr int MTW = COUT_T / 16 / WM;
    static constexpr int TT16 = TT / 16;
    static constexpr int NTT = TT16 / WN;
    static constexpr int NTW = FT * NTT;
    static constexpr int KK = KH * KW;
    static constexpr int ROWS = FT + KH - 1;
    static constexpr int HALO = (KW > 1) ? 4 : 0;       // 16-byte aligned halo: row = t in [t0-4, t0+TT+4)
    static constexpr int ROW = TT + 2 * HALO;
    static constexpr int QR = ROW / 4;                  // float4 quads per row
    static constexpr int PLANE = pad16mod32(ROWS * ROW);
    static constexpr int COUT_P = pad16mod32(COUT_T);
    static constexpr int IN_Q = CK * ROWS * QR;
    static constexpr int IN_PER_T = (IN_Q + 255) / 256;
    static constexpr int W_VEC = KK * CK * COUT_T / 4;
    static constexpr int W_PER_T = (W_VEC + 255) / 256;
    static constexpr int FO_T = POOL ? FT / 2 : FT;
    static constexpr int LDS_FLOATS = CK * PLANE + KK * CK * COUT_P + WN * COUT_T * FO_T * 2;
    static_assert(TT16 % WN == 0, "t tiles must split over waves");
    static_assert(!POOL || FT % 2 == 0, "pool needs row pairs");
    static_assert(CK % 4 == 0 && COUT_T % 16 == 0, "tile granularity");
};

template <int COUT_T, int FT, int TT, int KH, int KW, int CK, bool POOL, bool DGRAD>
__global__ __launch_bounds__(256) void conv_fwd_kernel(ConvFwdArgs a) {
    using C = ConvFwdCfg<COUT_T, FT, TT, KH, KW, CK, POOL>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* in_s = smem;                               // [CK][PLANE]
    float* w_s = smem + CK * C::PLANE;                // [KK][CK][COUT_P]
    float* st_s = w_s + C::KK * CK * C::COUT_P;       // [WN][COUT_T][FO_T][2]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WN, wn = wave % C::WN;
    const int lq = lane >> 4, lr = lane & 15;

    const int nTt = (a.T + TT - 1) / TT, nFt = (a.F + FT - 1) / FT;
    int bx = blockIdx.x;
    const int t0 = (bx % nTt) * TT; bx /= nTt;
    const int f0 = (bx % nFt) * FT;
    const int b = bx / nFt;
    const int cout0 = blockIdx.y * COUT_T;
    const int sl = a.seq_len ? min(a.seq_len[b], a.T) : a.T;
    const bool pro = a.scale != nullptr;
    constexpr int PADH = (KH - 1) / 2, PADW = (KW - 1) / 2;

    f32x4 acc[C::MTW][C::NTW];
#pragma unroll
    for (int m = 0; m < C::MTW; ++m)
#pragma unroll
        for (int n = 0; n < C::NTW; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- input staging: each thread owns IN_PER_T aligned float4 quads of the halo tile; their (channel, row,
    // column) decomposition is fixed across channel chunks.  Loads are raw buffer loads on clip-relative resources
    // with one 32-bit offset per quad, advanced by a uniform step per chunk: halo rows outside the plane, quads
    // outside the row and padded channels read 0 without branches (fp32 MFMAs and VALU instructions share the
    // SIMD's fp32 pipe on gfx950, so address arithmetic and predication are paid in MFMA time).  load_chunk only
    // issues loads; prologue and masks run in store_chunk, after the MFMAs of the previous chunk.
    u32x4_t rin[C::IN_PER_T];
    unsigned ridx[C::IN_PER_T], rsc[C::IN_PER_T], rsh[C::IN_PER_T];
    u32x4_t rw[C::W_PER_T];
    unsigned q_voff[C::IN_PER_T], q_cv[C::IN_PER_T], w_voff[C::W_PER_T];
    int q_lds[C::IN_PER_T], q_m[C::IN_PER_T];        // q_m: valid elements of the quad (e < q_m & 7), bit 8 = pool-row parity
    const bool unpool = DGRAD && a.unpool_idx != nullptr;
    const int Fsrc = unpool ? a.F / 2 : a.F;
    const int tlim = pro ? sl : a.T;                 // Normalization re-masks its output (y*mask)
    const bool vec = (a.T & 3) == 0;
    constexpr unsigned OOB = 0x80000000u;
    const unsigned clip_elems = (unsigned)(a.Cin * Fsrc * a.T);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x) + (size_t)b * clip_elems, 0, clip_elems * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc(
        unpool ? const_cast<uint8_t*>(a.unpool_idx) + (size_t)b * clip_elems : nullptr, 0, unpool ? clip_elems : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.wp), 0, (unsigned)(C::KK * a.CinP * a.CoutP) * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_sc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.scale), 0, pro ? (unsigned)a.Cin * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_sh = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.shift), 0, pro ? (unsigned)a.Cin * 4u : 0u, 0x00020000);
    const unsigned step_x = (unsigned)(CK * Fsrc * a.T) * 4u, step_w = (unsigned)(CK * a.CoutP) * 4u;
#pragma unroll
    for (int i = 0; i < C::IN_PER_T; ++i) {
        const int q = tid + i * 256;
        const int c = q / (C::ROWS * C::QR), rem = q - c * (C::ROWS * C::QR);
        const int r = rem / C::QR, qc = rem - r * C::QR;
        const int f = f0 - PADH + r, tq = t0 - C::HALO + 4 * qc;
        const bool ok = q < C::IN_Q && f >= 0 && f < a.F && tq >= 0 && tq < a.T;
        q_lds[i] = c * C::PLANE + r * C::ROW + 4 * qc;
        q_voff[i] = ok ? (unsigned)((c * Fsrc + (unpool ? (f >> 1) : f)) * a.T + tq) * 4u : OOB;
        q_cv[i] = ok ? (unsigned)c * 4u : OOB;
        q_m[i] = (ok ? min(max(tlim - tq, 0), 4) : 0) | ((f & 1) << 8);
    }
#pragma unroll
    for (int i = 0; i < C::W_PER_T; ++i) {
        const int idx = tid + i * 256;
        const int q = idx % (COUT_T / 4), c = (idx / (COUT_T / 4)) % CK, kk = idx / (COUT_T / 4) / CK;
        w_voff[i] = idx < C::W_VEC ? (unsigned)((kk * a.CinP + c) * a.CoutP + cout0 + q * 4) * 4u : OOB;
    }

    auto load_chunk = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < C::IN_PER_T; ++i) {
            rin[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, q_voff[i], 0, 0);
            if (unpool) {
                if (vec) {
                    ridx[i] = __builtin_amdgcn_raw_buffer_load_b32(rs_i, q_voff[i] >> 2, 0, 0);
                } else {                             // a dword straddling the end of the clip would read 0 as a whole
                    unsigned w = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        w |= (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rs_i, (q_voff[i] >> 2) + e, 0, 0) << (8 * e);
                    ridx[i] = w;
                }
            }
            if (pro) {
                rsc[i] = __builtin_amdgcn_raw_buffer_load_b32(rs_sc, q_cv[i], 0, 0);
                rsh[i] = __builtin_amdgcn_raw_buffer_load_b32(rs_sh, q_cv[i], 0, 0);
            }
            q_voff[i] += step_x; q_cv[i] += CK * 4u;
        }
#pragma unroll
        for (int i = 0; i < C::W_PER_T; ++i) {
            rw[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_voff[i], 0, 0);
            w_voff[i] += step_w;
        }
    };
    auto store_chunk = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < C::IN_PER_T; ++i) {
            float v[4] = {__uint_as_float(rin[i].x), __uint_as_float(rin[i].y), __uint_as_float(rin[i].z), __uint_as_float(rin[i].w)};
            const int n_ok = q_m[i] & 7;
            if (unpool) {
                const int par = q_m[i] >> 8;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (int)((ridx[i] >> (8 * e)) & 0xffu) == par ? v[e] : 0.f;
            }
            if (pro) {
                const float sc = __uint_as_float(rsc[i]), sh = __uint_as_float(rsh[i]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = fmaf(v[e], sc, sh);
                    if (a.relu) v[e] = fmaxf(v[e], 0.f);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = e < n_ok ? v[e] : 0.f;          // zero padding is post-activation
            if (tid + i * 256 < C::IN_Q) *reinterpret_cast<float4*>(in_s + q_lds[i]) = make_float4(v[0], v[1], v[2], v[3]);
        }
#pragma unroll
        for (int i = 0; i < C::W_PER_T; ++i) {
            const int idx = tid + i * 256;
            if (idx < C::W_VEC) {
                const int q = idx % (COUT_T / 4);
                const int ck = idx / (COUT_T / 4);   // kk*CK + c
                *reinterpret_cast<u32x4_t*>(w_s + ck * C::COUT_P + q * 4) = rw[i];
            }
        }
    };


    // the data-gradient instances gain ~4 % from exposing three taps to the scheduler, the forward ones lose occupancy
    constexpr int KK_UNROLL = (DGRAD && COUT_T != 32) ? 3 : 1;
    const int nChunks = a.CinP / CK;
    load_chunk();
    for (int ch = 0; ch < nChunks; ++ch) {
        __syncthreads();            // previous chunk's MFMA reads are done
        store_chunk();
        __syncthreads();
        if (ch + 1 < nChunks) load_chunk();                // in flight during the MFMAs below
#pragma unroll KK_UNROLL
        for (int kk = 0; kk < C::KK; ++kk) {
            const int kh = kk / KW, kw = kk % KW;
#pragma unroll
            for (int cs = 0; cs < CK / 4; ++cs) {
                float af[C::MTW], bf[C::NTW];
#pragma unroll
                for (int m = 0; m < C::MTW; ++m)
                    af[m] = w_s[(kk * CK + cs * 4 + lq) * C::COUT_P + (wm * C::MTW + m) * 16 + lr];
#pragma unroll
                for (int n = 0; n < C::NTW; ++n) {
                    const int fl = n / C::NTT, tt = wn * C::NTT + n % C::NTT;
                    bf[n] = in_s[(cs * 4 + lq) * C::PLANE + (fl + kh) * C::ROW + tt * 16 + lr + kw + (C::HALO - PADW)];
                }
#pragma unroll
                for (int m = 0; m < C::MTW; ++m)
#pragma unroll
                    for (int n = 0; n < C::NTW; ++n) acc[m][n] = mfma16(af[m], bf[n], acc[m][n]);
            }
        }
    }

    conv_epilogue<COUT_T, FT, TT, C::MTW, C::NTT, C::WN, POOL, DGRAD>(a, acc, st_s, b, f0, t0, cout0, sl, wm, wn, lq, lr, tid);
}

template <int COUT_T, int FT, int TT, int KH, int KW, int CK, bool POOL, bool DGRAD>
static int launch_cfg(const ConvFwdArgs& a, hipStream_t s) {
    using C = ConvFwdCfg<COUT_T, FT, TT, KH, KW, CK, POOL>;
    if (a.CinP % CK || a.CoutP % COUT_T) {
        set_error("conv_fwd: packed dims CinP=%d CoutP=%d not multiples of tile (%d,%d)", a.CinP,
                  a.CoutP, CK, COUT_T);
        return PBSED_E_ARG;
    }
    const int nTt = (a.T + TT - 1) / TT, nFt = (a.F + FT - 1) / FT;
    dim3 grid(nTt * nFt * a.B, a.CoutP / COUT_T);
    const size_t lds = C::LDS_FLOATS * sizeof(float);
    auto kern = conv_fwd_kernel<COUT_T, FT, TT, KH, KW, CK, POOL, DGRAD>;
    PBSED_DYN_LDS_ONCE(kern, lds);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    return check_launch("conv_fwd");
}

// Tile selection.  cin/cout padding granularity used by pack_conv_weights must match
// (conv_tile_dims below is the single source of truth for both).
void conv_fwd_tile_dims(int KH, int KW, int Cin, int Cout, int* ck, int* cout_t) {
    if (KH == 3) {
        // measured on MI355X: a 128-wide Cout tile is slower than two 64-wide ones (accumulator registers halve the occupancy)
        *cout_t = Cout <= 16 ? 16 : Cout <= 32 ? 32 : 64;
        *ck = (Cin <= 4 && *cout_t == 16) ? 4 : 8;       // the 4-channel stage exists for the 16-wide first layer only
    } else {
        *ck = (KW == 1) ? 16 : 8;
        *cout_t = Cout <= 16 ? 16 : 128;
    }
}

int conv_fwd_launch(const ConvFwdArgs& a, int KH, int KW, int pool, int dgrad, hipStream_t s) {
    int ck, ct;
    conv_fwd_tile_dims(KH, KW, a.Cin, a.Cout, &ck, &ct);
    if (pool && (a.F % 2)) { set_error("conv_fwd: pool needs even F"); return PBSED_E_ARG; }
    // the loaders address one clip with 32-bit byte offsets (buffer loads; 2^31 marks "out of range")
    if ((size_t)a.Cin * a.F * a.T * 4 >= (1ull << 30) || (size_t)a.Cout * a.F * a.T * 4 >= (1ull << 29) ||
        (size_t)KH * KW * a.CinP * a.CoutP * 4 >= (1ull << 31)) {
        set_error("conv_fwd: one clip of the input / output must stay below 1 GiB / 512 MiB (Cin=%d Cout=%d F=%d T=%d)", a.Cin,
                  a.Cout, a.F, a.T);
        return PBSED_E_ARG;
    }
    if (dgrad && pool) { set_error("conv dgrad: pool epilogue not valid"); return PBSED_E_ARG; }
    if (dgrad && a.unpool_idx && (a.F % 2)) { set_error("conv dgrad: unpool needs even F"); return PBSED_E_ARG; }
#define CFG(CT, FT_, TT_, KH_, KW_, CK_)                                          \
    do {                                                                          \
        if (dgrad) return launch_cfg<CT, FT_, TT_, KH_, KW_, CK_, false, true>(a, s);  \
        if (pool) return launch_cfg<CT, FT_, TT_, KH_, KW_, CK_, true, false>(a, s);   \
        return launch_cfg<CT, FT_, TT_, KH_, KW_, CK_, false, false>(a, s);            \
    } while (0)
    if (KH == 3 && KW == 3) {
        if (ck == 4 && ct == 16) CFG(16, 4, 128, 3, 3, 4);
        if (ct == 16) CFG(16, 4, 128, 3, 3, 8);
        if (ct == 32) CFG(32, 4, 64, 3, 3, 8);
        if (ct == 64) CFG(64, 4, 64, 3, 3, 8);
    } else if (KH == 1 && KW == 1 && pool) {
        // 1x1 conv of a 2-D tensor followed by the (2,1) pool (the 'deep' net configuration,
        // pb_sed/experiments/weak_label_crnn/training.py:170-183): two frequency rows per tile
        if (ct == 16) return launch_cfg<16, 2, 64, 1, 1, 16, true, false>(a, s);
        return launch_cfg<128, 2, 64, 1, 1, 16, true, false>(a, s);
    } else if (KH == 1 && !pool) {
#define CFG1(CT, KW_, CK_)                                                            \
    do {                                                                              \
        if (dgrad) return launch_cfg<CT, 1, 64, 1, KW_, CK_, false, true>(a, s);      \
        return launch_cfg<CT, 1, 64, 1, KW_, CK_, false, false>(a, s);                \
    } while (0)
        if (KW == 1) {
            if (ct == 16) CFG1(16, 1, 16);
            CFG1(128, 1, 16);
        }
        if (KW == 3) {
            if (ct == 16) CFG1(16, 3, 8);
            CFG1(128, 3, 8);
        }
#undef CFG1
    }
#undef CFG
    set_error("conv_fwd: unsupported kernel %dx%d pool=%d", KH, KW, pool);
    return PBSED_E_UNSUPPORTED;
}

}  // namespace pbsed
