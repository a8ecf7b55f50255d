// Time-major projections around the GRU scans:  Y[r][n] = bias[n] + sum_i sum_k X_i[r][k] * W_i[n][k],  r = (t, b).
//
// The scans work on time-major buffers [T][B][*]; their input projections gi = W_ih x + b_ih (torch.nn.GRU, reached through
// pb_sed/models/weak_label/crnn.py:61-67) and the data gradient dx = sum over chains of dgi W_ih (autograd of the same
// product) are plain row-major GEMMs there - X rows and W rows are both contiguous along the contraction index, which is
// exactly what the bf16 MFMA wants (8 consecutive k per lane = one 16-byte LDS read) - so nothing is transposed: not the
// operands, and not the [B,C,T] <-> [T,B,C] round trips a convolution kernel on the CNN's layout needs on either side.
// Several (X_i, W_i) pairs accumulate into one output (the chains of a data gradient; the halves of a bidirectional input).
// NS = 3: exact three-way bf16 operand splits, fp32-class results (Bf3 in common.h); NS = 1: plain bf16 operands.
// Block = 128 rows x 128 outputs, 4 waves (64 x 64 each), 32 k per stage, operands staged as bf16 rows of 80 bytes
// (40 halfs: 16-byte reads of 16 consecutive rows spread over all banks), next stage's loads in flight during the MFMAs.
#include <cstdlib>

#include "common.h"
#include "pbsed_internal.h"

namespace pbsed {

constexpr int TG_MAX = 4;                 // (X, W) pairs per launch
constexpr int TG_BR = 128, TG_BN = 128, TG_KC = 32, TG_KP = 40;

struct TmGemmArgs {
    const float* x[TG_MAX];               // [R][K_i]
    const float* w[TG_MAX];               // [N][K_i]
    int k[TG_MAX];
    int row_shift[TG_MAX];                // source row = output row + row_shift (rows outside [0, R) read as zero)
    const float* bias;                    // [N] or null
    float* y;                             // [R][N]
    int n_src, R, N;
    // 1-D convolution layers in the time-major layout (a k = 3 layer = three sources with row shifts -B, 0, +B):
    const float* scale;                   // [K] prologue of every source: v = x * scale + shift, ReLU if `relu`; or null
    const float* shift;
    int relu;
    const float* rowmask;                 // [R] 1 / 0 (t < seq_len[b]) or null: with a prologue, masked source rows are zero AFTER it
                                          // (zero padding is post-activation); masked output rows do not count in the statistics
    double* stats;                        // EPI 1 / 2: [PBSED_STAT_SLOTS][N][2] sums, zeroed by the caller
    const float* bx;                      // EPI 2: the raw input [R][N] of the layer whose norm + ReLU is differentiated
    const float* bscale; const float* bshift; const float* bmean; const float* binvstd;     // [N]
    int brelu;
};

// EPI 0: y = acc + bias.  EPI 1: the same + masked per-output sums (sum y, sum y^2) for the next layer's batch norm.
// EPI 2 (data gradient): backward through mask -> ReLU -> BN-apply of the layer's prologue, dz = da [z > 0] [row counted],
// with the sums BN backward needs (sum dz, sum dz * xhat) - what conv_epilogue does for the kernels on the CNN layout.
template <int NS, int EPI>
__global__ __launch_bounds__(256, 2) void tm_gemm_kernel(TmGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned short tg_smem[];
    unsigned short* xs = tg_smem;                                  // [NS][BR][KP]
    unsigned short* ws = tg_smem + NS * TG_BR * TG_KP;             // [NS][BN][KP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane >> 4, lr = lane & 15;
    const int wr = wave >> 1, wn = wave & 1;                       // wave tile: rows wr*64.., outputs wn*64..
    const int r0 = blockIdx.x * TG_BR, n0 = blockIdx.y * TG_BN;

    // staging items: (row or output, 8-k group): 128 x 4 per operand, two of each per thread
    const int it_row = tid >> 2, it_kq = tid & 3;                  // item 0: rows 0..63, item 1: rows 64..127
    constexpr unsigned OOB = 0x80000000u;
    u32x4_t rx[2][2], rw[2][2];
    float4 psc[2], psh[2];                                         // prologue factors of this thread's 8 k of the stage
    float rm[2];                                                   // row factor of the two items: 0 = masked / outside
    const bool pro = a.scale != nullptr;
    int src = 0, k0 = 0;                                           // stage cursor: source pair, first k of the stage
    auto fetch = [&]() __attribute__((always_inline)) {
        const int K = a.k[src];
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x[src]), 0, (unsigned)((size_t)a.R * K * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w[src]), 0, (unsigned)((size_t)a.N * K * 4), 0x00020000);
        const int kk = k0 + it_kq * 8;
        if (pro) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bool kok = kk + 4 * h + 4 <= K;
                psc[h] = kok ? *reinterpret_cast<const float4*>(a.scale + kk + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
                psh[h] = kok ? *reinterpret_cast<const float4*>(a.shift + kk + 4 * h) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = r0 + it_row + 64 * i, n = n0 + it_row + 64 * i;
            const int rs = r + a.row_shift[src];
            const bool rok = r < a.R && rs >= 0 && rs < a.R;
            rm[i] = rok ? (a.rowmask ? a.rowmask[rs] : 1.f) : 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bool kok = kk + 4 * h + 4 <= K;
                rx[i][h] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (kok && rok) ? (unsigned)(((size_t)rs * K + kk + 4 * h) * 4) : OOB, 0, 0);
                rw[i][h] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (kok && n < a.N) ? (unsigned)(((size_t)n * K + kk + 4 * h) * 4) : OOB, 0, 0);
            }
        }
    };
    auto advance = [&]() __attribute__((always_inline)) {          // -> false when all sources are consumed
        k0 += TG_KC;
        if (k0 >= a.k[src]) { k0 = 0; ++src; }
        return src < a.n_src;
    };
    auto put = [&](unsigned short* base, int row, const u32x4_t (&v)[2], bool is_x, float rowf) __attribute__((always_inline)) {
        float4 lo = make_float4(__uint_as_float(v[0].x), __uint_as_float(v[0].y), __uint_as_float(v[0].z), __uint_as_float(v[0].w));
        float4 hi = make_float4(__uint_as_float(v[1].x), __uint_as_float(v[1].y), __uint_as_float(v[1].z), __uint_as_float(v[1].w));
        if (is_x && pro) {
            float e[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            const float sc[8] = {psc[0].x, psc[0].y, psc[0].z, psc[0].w, psc[1].x, psc[1].y, psc[1].z, psc[1].w};
            const float sh[8] = {psh[0].x, psh[0].y, psh[0].z, psh[0].w, psh[1].x, psh[1].y, psh[1].z, psh[1].w};
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float t = fmaf(e[q], sc[q], sh[q]);
                if (a.relu) t = fmaxf(t, 0.f);
                e[q] = rowf != 0.f ? t : 0.f;                        // masked / outside rows: zero AFTER the activation
            }
            lo = make_float4(e[0], e[1], e[2], e[3]); hi = make_float4(e[4], e[5], e[6], e[7]);
        }                                                            // no prologue: the source is taken as it is (rows outside read 0)
        u32x4_t* dst = reinterpret_cast<u32x4_t*>(base + (size_t)row * TG_KP + it_kq * 8);
        if constexpr (NS == 3) {
            const Bf3 p = split3x8(lo, hi);
            dst[0] = p.hi;
            *reinterpret_cast<u32x4_t*>(base + (size_t)(TG_BR + row) * TG_KP + it_kq * 8) = p.mid;
            *reinterpret_cast<u32x4_t*>(base + (size_t)(2 * TG_BR + row) * TG_KP + it_kq * 8) = p.lo;
        } else {
            dst[0] = u32x4_t{pack_bf16_rne(lo.x, lo.y), pack_bf16_rne(lo.z, lo.w), pack_bf16_rne(hi.x, hi.y), pack_bf16_rne(hi.z, hi.w)};
        }
    };
    static_assert(TG_BR == TG_BN, "one part stride for both operands");

    f32x4 acc[4][4];                                               // [output tile][row tile]
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[nt][rt] = f32x4{0.f, 0.f, 0.f, 0.f};

    fetch();
    bool more = true;
    while (more) {
        __syncthreads();                                           // the previous stage's fragments have been read
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            put(xs, it_row + 64 * i, rx[i], true, rm[i]);
            put(ws, it_row + 64 * i, rw[i], false, 1.f);
        }
        __syncthreads();
        more = advance();
        if (more) fetch();                                         // in flight during the MFMAs
        u32x4_t af[4][NS];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int p = 0; p < NS; ++p)
                af[nt][p] = *reinterpret_cast<const u32x4_t*>(ws + (size_t)(p * TG_BN + wn * 64 + nt * 16 + lr) * TG_KP + lq * 8);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            u32x4_t bf[NS];
#pragma unroll
            for (int p = 0; p < NS; ++p)
                bf[p] = *reinterpret_cast<const u32x4_t*>(xs + (size_t)(p * TG_BR + wr * 64 + rt * 16 + lr) * TG_KP + lq * 8);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                if constexpr (NS == 3) acc[nt][rt] = mfma_x3(Bf3{af[nt][0], af[nt][1], af[nt][2]}, Bf3{bf[0], bf[1], bf[2]}, acc[nt][rt]);
                else acc[nt][rt] = mfma_b16(af[nt][0], bf[0], acc[nt][rt]);
            }
        }
    }

    // D[output 4 lq + reg][row lr]: a lane holds 4 consecutive outputs of one row -> one 16-byte store
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (unsigned)((size_t)a.R * a.N * 4), 0x00020000);
    bool counted[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        const int r = r0 + wr * 64 + rt * 16 + lr;
        counted[rt] = r < a.R && (!a.rowmask || a.rowmask[r] != 0.f);
    }
    float s1[4][4], s2[4][4];                                      // EPI 1 / 2: this lane's partial sums per (output tile, reg)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n = n0 + wn * 64 + nt * 16 + lq * 4;
        const bool nok = n < a.N;
        float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.bias && nok) bs = *reinterpret_cast<const float4*>(a.bias + n);
        float4 bsc = bs, bsh = bs, bmu = bs, bis = bs;
        if (EPI == 2 && nok) {
            bsc = *reinterpret_cast<const float4*>(a.bscale + n); bsh = *reinterpret_cast<const float4*>(a.bshift + n);
            bmu = *reinterpret_cast<const float4*>(a.bmean + n); bis = *reinterpret_cast<const float4*>(a.binvstd + n);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) s1[nt][j] = s2[nt][j] = 0.f;
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            const int r = r0 + wr * 64 + rt * 16 + lr;
            const bool ok = r < a.R && nok;
            float v[4] = {acc[nt][rt][0] + bs.x, acc[nt][rt][1] + bs.y, acc[nt][rt][2] + bs.z, acc[nt][rt][3] + bs.w};
            if (EPI == 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float vm = counted[rt] ? v[j] : 0.f;
                    s1[nt][j] += vm; s2[nt][j] = fmaf(vm, vm, s2[nt][j]);
                }
            }
            if (EPI == 2) {
                float4 xq = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) xq = *reinterpret_cast<const float4*>(a.bx + (size_t)r * a.N + n);
                const float xv[4] = {xq.x, xq.y, xq.z, xq.w};
                const float c_sc[4] = {bsc.x, bsc.y, bsc.z, bsc.w}, c_sh[4] = {bsh.x, bsh.y, bsh.z, bsh.w};
                const float c_mu[4] = {bmu.x, bmu.y, bmu.z, bmu.w}, c_is[4] = {bis.x, bis.y, bis.z, bis.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float z = fmaf(xv[j], c_sc[j], c_sh[j]);
                    const bool keep = counted[rt] && (!a.brelu || z > 0.f);
                    v[j] = keep ? v[j] : 0.f;
                    s1[nt][j] += v[j]; s2[nt][j] = fmaf(v[j], (xv[j] - c_mu[j]) * c_is[j], s2[nt][j]);
                }
            }
            const u32x4_t o = u32x4_t{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
            __builtin_amdgcn_raw_buffer_store_b128(o, rs_y, ok ? (unsigned)(((size_t)r * a.N + n) * 4) : OOB, 0, 0);
        }
    }
    if (EPI != 0 && a.stats) {
        // rows of a wave: the 16 lanes sharing lq (DPP), then the two row halves of the block through LDS, one atomic per
        // (output, moment) into one of the slot copies
        __syncthreads();                                           // the operand tiles are dead
        float* st = reinterpret_cast<float*>(tg_smem);             // [2 row halves][128 outputs][2]
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float t1 = wave_sum16(s1[nt][j]), t2 = wave_sum16(s2[nt][j]);
                if (lr == 0) {
                    const int nl = wn * 64 + nt * 16 + lq * 4 + j;
                    st[(wr * TG_BN + nl) * 2 + 0] = t1;
                    st[(wr * TG_BN + nl) * 2 + 1] = t2;
                }
            }
        __syncthreads();
        const int nl = tid >> 1, which = tid & 1, n = n0 + nl;
        if (n < a.N) {
            const float v = st[nl * 2 + which] + st[(TG_BN + nl) * 2 + which];
            atomicAdd(&a.stats[((size_t)(blockIdx.x & (PBSED_STAT_SLOTS - 1)) * a.N + n) * 2 + which], (double)v);
        }
    }
}

}  // namespace pbsed

using namespace pbsed;

static int tm_launch(TmGemmArgs& a, int n_src, const float* const* x, const float* const* w, const int* k, int bf16, int epi,
                     void* stream) {
    const int R = a.R, N = a.N;
    if (n_src < 1 || n_src > TG_MAX || R < 1 || N < 4 || (N & 3)) {
        set_error("tm_gemm: need 1 <= n_src <= %d, N a multiple of 4 (n_src=%d R=%d N=%d)", TG_MAX, n_src, R, N);
        return PBSED_E_ARG;
    }
    for (int i = 0; i < n_src; ++i) {
        if (k[i] < 4 || (k[i] & 3)) { set_error("tm_gemm: K must be a multiple of 4 (k[%d]=%d)", i, k[i]); return PBSED_E_ARG; }
        if ((size_t)R * k[i] * 4 >= (1ull << 31) || (size_t)N * k[i] * 4 >= (1ull << 31)) {
            set_error("tm_gemm: an operand exceeds the 2 GiB the loaders address (R=%d N=%d K=%d)", R, N, k[i]);
            return PBSED_E_ARG;
        }
        if (a.scale && k[i] != k[0]) { set_error("tm_gemm: sources under one prologue need the same width"); return PBSED_E_ARG; }
        a.x[i] = x[i]; a.w[i] = w[i]; a.k[i] = k[i];
    }
    if ((size_t)R * N * 4 >= (1ull << 31)) { set_error("tm_gemm: output of %d x %d floats exceeds 2 GiB", R, N); return PBSED_E_ARG; }
    a.n_src = n_src;
    dim3 grid((R + TG_BR - 1) / TG_BR, (N + TG_BN - 1) / TG_BN);
    const int ns = bf16 ? 1 : 3;
    const size_t lds = (size_t)ns * (TG_BR + TG_BN) * TG_KP * sizeof(unsigned short);
    hipStream_t s = (hipStream_t)stream;
#define TM_GO(NS_, EPI_)                                                                  \
    do {                                                                                  \
        PBSED_DYN_LDS_ONCE((tm_gemm_kernel<NS_, EPI_>), lds);                             \
        hipLaunchKernelGGL((tm_gemm_kernel<NS_, EPI_>), grid, dim3(256), lds, s, a);      \
    } while (0)
    if (ns == 3) { if (epi == 0) TM_GO(3, 0); else if (epi == 1) TM_GO(3, 1); else TM_GO(3, 2); }
    else { if (epi == 0) TM_GO(1, 0); else if (epi == 1) TM_GO(1, 1); else TM_GO(1, 2); }
#undef TM_GO
    return check_launch("tm_gemm");
}

// y [R, N] = bias + sum_i x[i] [R, k[i]] @ w[i] [N, k[i]]^T.  bf16 != 0: plain bf16 operands; else exact bf16x3 splits.
extern "C" int pbsed_tm_gemm(int n_src, const float* const* x, const float* const* w, const int* k, const float* bias,
                             float* y, int R, int N, int bf16, void* stream) {
    TmGemmArgs a{};
    a.bias = bias; a.y = y; a.R = R; a.N = N;
    return tm_launch(a, n_src, x, w, k, bf16, 0, stream);
}

// 1-D convolution layer (kernel size = n_taps, 'same' zero padding) of a [T, B, C] tensor, forward:
//   y[t] = bias + sum_tap w[tap] @ pro(x)[t + tap - n_taps / 2],   pro(x) = rowmask * relu?(x * scale + shift)  (scale may be NULL)
// w: n_taps matrices [N, K]; stats (or NULL): masked sums of y for the next layer's batch norm.
extern "C" int pbsed_tm_conv_fwd(const float* x, int n_taps, const float* const* w, const float* bias, const float* scale,
                                 const float* shift, int relu, const float* rowmask, float* y, double* stats, int T, int B,
                                 int K, int N, int bf16, void* stream) {
    if (n_taps < 1 || n_taps > TG_MAX || !(n_taps & 1)) { set_error("tm_conv_fwd: %d taps (odd, <= %d)", n_taps, TG_MAX); return PBSED_E_ARG; }
    TmGemmArgs a{};
    const float* xs[TG_MAX];
    int ks[TG_MAX];
    for (int i = 0; i < n_taps; ++i) { xs[i] = x; ks[i] = K; a.row_shift[i] = (i - n_taps / 2) * B; }
    a.bias = bias; a.y = y; a.R = T * B; a.N = N;
    a.scale = scale; a.shift = shift; a.relu = relu; a.rowmask = rowmask; a.stats = stats;
    return tm_launch(a, n_taps, xs, w, ks, bf16, stats ? 1 : 0, stream);
}

// Its data gradient: g [T, B, N_out] -> dz [T, B, K_in], wt: n_taps matrices [K_in, N_out] (the taps arrive flipped:
// wt[tap] multiplies g[t - (tap - n_taps/2)]).  bx != NULL: backward through the layer's prologue (mask, ReLU, BN-apply of the
// raw input bx [T, B, K_in]) with the sums (sum dz, sum dz * xhat) in stats.
extern "C" int pbsed_tm_conv_bwd_data(const float* g, int n_taps, const float* const* wt, const float* rowmask, float* dz,
                                      const float* bx, const float* bscale, const float* bshift, const float* bmean,
                                      const float* binvstd, int brelu, double* stats, int T, int B, int N_out, int K_in, int bf16,
                                      void* stream) {
    if (n_taps < 1 || n_taps > TG_MAX || !(n_taps & 1)) { set_error("tm_conv_bwd_data: %d taps (odd, <= %d)", n_taps, TG_MAX); return PBSED_E_ARG; }
    if (bx && (!stats || !bscale || !bshift || !bmean || !binvstd)) { set_error("tm_conv_bwd_data: BN backward needs its factors and stats"); return PBSED_E_ARG; }
    TmGemmArgs a{};
    const float* gs[TG_MAX];
    int ks[TG_MAX];
    for (int i = 0; i < n_taps; ++i) { gs[i] = g; ks[i] = N_out; a.row_shift[i] = -(i - n_taps / 2) * B; }
    a.y = dz; a.R = T * B; a.N = K_in;
    a.rowmask = bx ? rowmask : nullptr;            // without a prologue the forward did not mask either
    a.stats = stats; a.bx = bx; a.bscale = bscale; a.bshift = bshift; a.bmean = bmean; a.binvstd = binvstd; a.brelu = brelu;
    return tm_launch(a, n_taps, gs, wt, ks, bf16, bx ? 2 : 0, stream);
}
