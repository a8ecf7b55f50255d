// Weight/bias gradients of the GRU stacks straight from the scans' time-major buffers:
//
//     dW[g][k] += sum_{t,b} dG[t][b][g] * X[t + shift][b][k]        db[g] += sum_{t,b} dG[t][b][g]
//
// which is what autograd accumulates into weight_ih / weight_hh / bias_* of torch.nn.GRU
// (pb_sed/models/base.py:64-68 -> padertorch GRU wrapper -> nn.GRU backward).  dG is the scan's dgi or dgh
// [T*B, G], X the layer input or the layer's own state sequence [T*B, K] (shift = -1 / +1 picks h_{t-1} of a
// forward / time-reversed chain; rows shifted outside [0, T) are zero).  Both operands have the reduction
// index (t, b) as the slow dimension, so rows are fetched with coalesced float4 loads, staged in LDS with a
// row stride = 16 mod 32 banks and read as fp32 MFMA fragments without any transpose.  All GEMMs of a
// backward pass (2 per chain and layer) go in one launch; the reduction is split over blocks and combined
// with float atomics into the (pre-zeroed / accumulating) gradient buffers.
#include <cstdlib>

#include "common.h"
#include "pbsed_internal.h"

namespace pbsed {

constexpr int GW_MAX = 16;     // GEMMs per launch
constexpr int GW_BM = 128;     // gate rows per block
constexpr int GW_KC = 16;      // (t,b) rows per LDS stage

struct GruWgradArgs {
    const float* dg[GW_MAX];
    const float* x[GW_MAX];
    float* dw[GW_MAX];
    float* db[GW_MAX];
    int shift_rows[GW_MAX];    // shift * B
    int TB, G, K, nsplit, rows_per_split;
};

template <int BN>
__global__ __launch_bounds__(BN * 2) void gru_wgrad_kernel(GruWgradArgs a) {
    constexpr int NT = BN * 2, WN = BN / 64;            // threads; waves along N (2 along M)
    constexpr int SA = GW_BM + 16, SB = BN + 16;        // LDS row strides: = 16 mod 32 banks
    __shared__ float As[2][GW_KC][SA];
    __shared__ float Bs[2][GW_KC][SB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane >> 4, lr = lane & 15;
    const int wm = wave / WN, wn = wave % WN;
    const int gemm = blockIdx.z / a.nsplit, split = blockIdx.z % a.nsplit;
    const int m0 = blockIdx.x * GW_BM, n0 = blockIdx.y * BN;
    const float* __restrict__ dg = a.dg[gemm];
    const float* __restrict__ x = a.x[gemm];
    const int shift = a.shift_rows[gemm];
    const int r_begin = split * a.rows_per_split, r_end = min(a.TB, r_begin + a.rows_per_split);
    if (r_begin >= r_end) return;

    // staging assignment: A stage = 16 rows x 32 float4, B stage = 16 rows x BN/4 float4
    constexpr int A_PER = GW_KC * (GW_BM / 4) / NT;     // 2 (BN=128) or 1 (BN=256)
    constexpr int B_PER = GW_KC * (BN / 4) / NT;        // 2
    float4 ra[A_PER], rb[B_PER];
    auto fetch = [&](int r0) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int f = tid + i * NT, row = f / (GW_BM / 4), c = (f % (GW_BM / 4)) * 4;
            const int r = r0 + row, g = m0 + c;
            ra[i] = (r < r_end && g < a.G) ? *reinterpret_cast<const float4*>(dg + (size_t)r * a.G + g)
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int f = tid + i * NT, row = f / (BN / 4), c = (f % (BN / 4)) * 4;
            const int r = r0 + row, rs = r + shift, k = n0 + c;
            rb[i] = (r < r_end && rs >= 0 && rs < a.TB && k < a.K) ? *reinterpret_cast<const float4*>(x + (size_t)rs * a.K + k)
                                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int f = tid + i * NT, row = f / (GW_BM / 4), c = (f % (GW_BM / 4)) * 4;
            *reinterpret_cast<float4*>(&As[buf][row][c]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int f = tid + i * NT, row = f / (BN / 4), c = (f % (BN / 4)) * 4;
            *reinterpret_cast<float4*>(&Bs[buf][row][c]) = rb[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;                                   // partial column sum of dG (N tile 0 only)

    fetch(r_begin);
    stage(0);
    __syncthreads();
    int buf = 0;
    for (int r0 = r_begin; r0 < r_end; r0 += GW_KC) {
        const bool more = r0 + GW_KC < r_end;
        if (more) fetch(r0 + GW_KC);
#pragma unroll
        for (int kk = 0; kk < GW_KC / 4; ++kk) {
            float av[4], bv[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) av[mi] = As[buf][kk * 4 + lq][wm * 64 + mi * 16 + lr];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) bv[ni] = Bs[buf][kk * 4 + lq][wn * 64 + ni * 16 + lr];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma16(av[mi], bv[ni], acc[mi][ni]);
        }
        if (blockIdx.y == 0) {                       // column sums of dG: every thread a slice of the stage's rows
            constexpr int PARTS = NT / GW_BM, ROWS = GW_KC / PARTS;
#pragma unroll
            for (int r = 0; r < ROWS; ++r) bsum += As[buf][(tid / GW_BM) * ROWS + r][tid % GW_BM];
        }
        if (more) stage(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    float* __restrict__ dw = a.dw[gemm];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int k = n0 + wn * 64 + ni * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int g = m0 + wm * 64 + mi * 16 + lq * 4 + r;
                if (g < a.G && k < a.K) unsafeAtomicAdd(dw + (size_t)g * a.K + k, acc[mi][ni][r]);
            }
        }
    if (blockIdx.y == 0 && m0 + tid % GW_BM < a.G && a.db[gemm]) unsafeAtomicAdd(a.db[gemm] + m0 + tid % GW_BM, bsum);
}

}  // namespace pbsed

using namespace pbsed;

extern "C" int pbsed_gru_wgrad(int n, const float* const* dg, const float* const* x, const int* shift, float* const* dw,
                               float* const* db, int T, int B, int G, int K, void* stream) {
    if (n < 1 || n > GW_MAX || T < 1 || B < 1 || G < 4 || K < 4 || (G & 3) || (K & 3)) {
        set_error("gru_wgrad: need 1 <= n <= %d, G and K multiples of 4 (n=%d G=%d K=%d)", GW_MAX, n, G, K);
        return PBSED_E_ARG;
    }
    GruWgradArgs a{};
    for (int i = 0; i < n; ++i) {
        a.dg[i] = dg[i]; a.x[i] = x[i]; a.dw[i] = dw[i]; a.db[i] = db ? db[i] : nullptr;
        a.shift_rows[i] = shift[i] * B;
    }
    a.TB = T * B; a.G = G; a.K = K;
    const bool wide = K > 128;
    const int bn = wide ? 256 : 128;
    dim3 grid((G + GW_BM - 1) / GW_BM, (K + bn - 1) / bn, 1);
    // split the (t,b) reduction so that the launch is one full residency round (blocks per CU from the occupancy
    // query; measured on MI355X: 768 blocks 0.48 ms, 512 blocks 0.62 ms, 1024 blocks 0.56 ms for 8 x [768 x 256 x 16000])
    const int tiles = grid.x * grid.y * n;
    static int target = 0;
    if (target == 0) {
        int occ = 0, dev = 0, n_cu = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        const hipError_t e = wide ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gru_wgrad_kernel<256>, 512, 0)
                                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gru_wgrad_kernel<128>, 256, 0);
        if (e != hipSuccess || occ < 1) occ = 2;
        target = n_cu * occ;
    }
    int nsplit = target / tiles;
    const int max_split = (a.TB + 4 * GW_KC - 1) / (4 * GW_KC);
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    a.rows_per_split = ((a.TB + nsplit - 1) / nsplit + GW_KC - 1) / GW_KC * GW_KC;
    a.nsplit = (a.TB + a.rows_per_split - 1) / a.rows_per_split;
    grid.z = n * a.nsplit;
    hipStream_t s = (hipStream_t)stream;
    if (wide) hipLaunchKernelGGL((gru_wgrad_kernel<256>), grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((gru_wgrad_kernel<128>), grid, dim3(256), 0, s, a);
    return check_launch("gru_wgrad");
}
