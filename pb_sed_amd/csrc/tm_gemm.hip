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

#ifndef TG_DBG
#define TG_DBG 0            // ablation switches of tools/kernel_ablation.sh (never set in the product build): 1 W not staged, 2 X not staged, 8 no MFMAs
#endif
constexpr int TG_MAX = 4;                 // (X, W) pairs per launch
constexpr int TG_BR = 128, TG_BN = 128, TG_KC = 32, TG_KP = 40;

struct TmGemmArgs {
    const float* x[TG_MAX];               // [R][K_i]
    const float* w[TG_MAX];               // [N][K_i]
    int k[TG_MAX];
    const float* bias;                    // [N] or null
    float* y;                             // [R][N]
    int n_src, R, N;
};

template <int NS>
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
    int src = 0, k0 = 0;                                           // stage cursor: source pair, first k of the stage
    auto fetch = [&]() __attribute__((always_inline)) {
        const int K = a.k[src];
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x[src]), 0, (unsigned)((size_t)a.R * K * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w[src]), 0, (unsigned)((size_t)a.N * K * 4), 0x00020000);
        const int kk = k0 + it_kq * 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = r0 + it_row + 64 * i, n = n0 + it_row + 64 * i;
            const bool rok = r < a.R;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const bool kok = kk + 4 * h + 4 <= K;
                rx[i][h] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (kok && rok) ? (unsigned)(((size_t)r * K + kk + 4 * h) * 4) : OOB, 0, 0);
                rw[i][h] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (kok && n < a.N) ? (unsigned)(((size_t)n * K + kk + 4 * h) * 4) : OOB, 0, 0);
            }
        }
    };
    auto advance = [&]() __attribute__((always_inline)) {          // -> false when all sources are consumed
        k0 += TG_KC;
        if (k0 >= a.k[src]) { k0 = 0; ++src; }
        return src < a.n_src;
    };
    auto put = [&](unsigned short* base, int row, const u32x4_t (&v)[2]) __attribute__((always_inline)) {
        const float4 lo = make_float4(__uint_as_float(v[0].x), __uint_as_float(v[0].y), __uint_as_float(v[0].z), __uint_as_float(v[0].w));
        const float4 hi = make_float4(__uint_as_float(v[1].x), __uint_as_float(v[1].y), __uint_as_float(v[1].z), __uint_as_float(v[1].w));
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
            if (!(TG_DBG & 2)) put(xs, it_row + 64 * i, rx[i]);
            if (!(TG_DBG & 1)) put(ws, it_row + 64 * i, rw[i]);
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
                if (TG_DBG & 8) continue;
                if constexpr (NS == 3) acc[nt][rt] = mfma_x3(Bf3{af[nt][0], af[nt][1], af[nt][2]}, Bf3{bf[0], bf[1], bf[2]}, acc[nt][rt]);
                else acc[nt][rt] = mfma_b16(af[nt][0], bf[0], acc[nt][rt]);
            }
        }
    }

    // D[output 4 lq + reg][row lr]: a lane holds 4 consecutive outputs of one row -> one 16-byte store
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (unsigned)((size_t)a.R * a.N * 4), 0x00020000);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n = n0 + wn * 64 + nt * 16 + lq * 4;
        const bool nok = n < a.N;
        float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.bias && nok) bs = *reinterpret_cast<const float4*>(a.bias + n);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            const int r = r0 + wr * 64 + rt * 16 + lr;
            const bool ok = r < a.R && nok;
            const u32x4_t o = u32x4_t{__float_as_uint(acc[nt][rt][0] + bs.x), __float_as_uint(acc[nt][rt][1] + bs.y),
                                      __float_as_uint(acc[nt][rt][2] + bs.z), __float_as_uint(acc[nt][rt][3] + bs.w)};
            __builtin_amdgcn_raw_buffer_store_b128(o, rs_y, ok ? (unsigned)(((size_t)r * a.N + n) * 4) : OOB, 0, 0);
        }
    }
}

}  // namespace pbsed

using namespace pbsed;

static int tm_launch(TmGemmArgs& a, int n_src, const float* const* x, const float* const* w, const int* k, int bf16, void* stream) {
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
        a.x[i] = x[i]; a.w[i] = w[i]; a.k[i] = k[i];
    }
    if ((size_t)R * N * 4 >= (1ull << 31)) { set_error("tm_gemm: output of %d x %d floats exceeds 2 GiB", R, N); return PBSED_E_ARG; }
    a.n_src = n_src;
    dim3 grid((R + TG_BR - 1) / TG_BR, (N + TG_BN - 1) / TG_BN);
    const int ns = bf16 ? 1 : 3;
    const size_t lds = (size_t)ns * (TG_BR + TG_BN) * TG_KP * sizeof(unsigned short);
    hipStream_t s = (hipStream_t)stream;
    if (ns == 3) {
        PBSED_DYN_LDS_ONCE(tm_gemm_kernel<3>, lds);
        hipLaunchKernelGGL(tm_gemm_kernel<3>, grid, dim3(256), lds, s, a);
    } else {
        PBSED_DYN_LDS_ONCE(tm_gemm_kernel<1>, lds);
        hipLaunchKernelGGL(tm_gemm_kernel<1>, grid, dim3(256), lds, s, a);
    }
    return check_launch("tm_gemm");
}

// y [R, N] = bias + sum_i x[i] [R, k[i]] @ w[i] [N, k[i]]^T.  bf16 != 0: plain bf16 operands; else exact bf16x3 splits.
extern "C" int pbsed_tm_gemm(int n_src, const float* const* x, const float* const* w, const int* k, const float* bias,
                             float* y, int R, int N, int bf16, void* stream) {
    TmGemmArgs a{};
    a.bias = bias; a.y = y; a.R = R; a.N = N;
    return tm_launch(a, n_src, x, w, k, bf16, stream);
}
