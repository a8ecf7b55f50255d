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

#include <type_traits>

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
    // bf16-MFMA kernel: GEMMs of different K share a launch; blockIdx.y enumerates the (GEMM, column tile) pairs
    int Ks[GW_MAX];
    unsigned char y_gemm[4 * GW_MAX], y_tile[4 * GW_MAX];
    // many-way splits of small gradients: every (GEMM, split) writes its partial tile to its own slot [G][K_max] with plain
    // stores and gru_wgrad_slot_reduce_kernel adds the slots up - instead of nsplit atomics per gradient element
    float* slots;
    int slot_kmax;
    int xcd_groups;            // producer / consumer kernel: place the row tiles of a (GEMM, column tile, split) on one XCD
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

// The same GEMMs on the bf16 MFMA (16x16x32).  NS = 3: every fp32 operand is split exactly into three bf16 parts while it
// is staged (Bf3 in common.h) and the six part products above 2^-24 are accumulated - fp32-class gradients at up to 2.6x
// the fp32-MFMA rate; NS = 1: plain bf16 operands (the bf16 training mode of BASELINE config 3), HBM-bound.
// The contraction index (t, b) is the slow dimension of both operands, the MFMA wants 8 consecutive contraction values per
// lane: a staging thread therefore fetches an 8-row x 4-column block (8 coalesced float4 rows), converts it, and writes
// each column's 8 values as ONE 16-byte LDS row.  LDS rows are stored at pos(c) = (c % 4) * (W / 4) + c / 4 of the tile's W
// columns, so that the four rows a thread writes per part are conflict-free and an MFMA tile = 16 consecutive rows
// (= columns c with the same c % 4, stride 4); the output indices follow the same bijection.
constexpr int GB_BM = 128, GB_BN = 256, GB_KC = 32, GB_KG = GB_KC / 8;

template <int NS>
__global__ __launch_bounds__(512) void gru_wgrad_b16_kernel(GruWgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32x4_t smem_b16[];
    u32x4_t* As = smem_b16;                              // [NS][KG][BM] 16-byte rows (8 bf16 along the contraction index)
    u32x4_t* Bs = smem_b16 + NS * GB_KG * GB_BM;         // [NS][KG][BN]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane >> 4, lr = lane & 15;
    const int wm = wave >> 2, wn = wave & 3;             // 2 x 4 waves, 64 x 64 outputs each
    const int gemm = a.y_gemm[blockIdx.y], split = blockIdx.z, K = a.Ks[gemm];
    const int m0 = blockIdx.x * GB_BM, n0 = a.y_tile[blockIdx.y] * GB_BN;
    const float* __restrict__ dg = a.dg[gemm];
    const float* __restrict__ x = a.x[gemm];
    const int shift = a.shift_rows[gemm];
    const int r_begin = split * a.rows_per_split, r_end = min(a.TB, r_begin + a.rows_per_split);
    if (r_begin >= r_end) return;

    // staging items: waves 0..1 (threads 0..127) one (8-row group, 4 gate columns) block of dG each, waves 2..5 one of X;
    // the role is wave-uniform, the loads are raw buffer loads whose out-of-range rows / columns read 0 without a branch
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const bool is_a = wave_u < 2, is_b = wave_u >= 2 && wave_u < 6;
    const int it = is_a ? tid : tid - GB_KG * (GB_BM / 4);
    const int wq = is_a ? GB_BM / 4 : GB_BN / 4;         // column quads of the tile
    const int kg = it / wq, jq = it % wq;
    const int col = (is_a ? m0 : n0) + 4 * jq;
    const int ld = is_a ? a.G : K;                     // row length of the operand
    const bool col_ok = (is_a || is_b) && col < ld;
    const int shift_u = is_a ? 0 : shift;
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(is_a ? dg : x), 0, (is_a || is_b) ? (unsigned)((size_t)a.TB * ld * 4) : 0u, 0x00020000);
    // two stages of operand rows are in flight in registers (rv0 / rv1): a stage's loads are issued two iterations before
    // they are converted, so a CU keeps ~100 KB of requests outstanding - the launch is bound by the fetch of its operands
    u32x4_t rv0[8], rv1[8];
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    auto fetch = [&](u32x4_t (&rv)[8], int r0) __attribute__((always_inline)) {
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int r = r0 + kg * 8 + rr, rs = r + shift_u;
            const bool ok = col_ok && r < r_end && rs >= 0 && rs < a.TB;
            rv[rr] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ok ? (unsigned)(((size_t)rs * ld + col) * 4) : OOB, 0, 0);
        }
    };
    auto stage = [&](const u32x4_t (&rv)[8]) __attribute__((always_inline)) {
        if (!(is_a || is_b)) return;
        u32x4_t* dst = (is_a ? As : Bs) + (size_t)kg * (4 * wq) + jq;
        const int part_stride = GB_KG * 4 * wq;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v[8];
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) v[rr] = __uint_as_float(i == 0 ? rv[rr].x : i == 1 ? rv[rr].y : i == 2 ? rv[rr].z : rv[rr].w);
            if (is_a) bsum[i] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            if constexpr (NS == 3) {
                const Bf3 p = split3x8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]));
                dst[i * wq] = p.hi; dst[part_stride + i * wq] = p.mid; dst[2 * part_stride + i * wq] = p.lo;
            } else {
                dst[i * wq] = u32x4_t{pack_bf16_rne(v[0], v[1]), pack_bf16_rne(v[2], v[3]), pack_bf16_rne(v[4], v[5]), pack_bf16_rne(v[6], v[7])};
            }
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&]() __attribute__((always_inline)) {
        u32x4_t af[4][NS];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int p = 0; p < NS; ++p) af[mi][p] = As[(size_t)(p * GB_KG + lq) * GB_BM + wm * 64 + mi * 16 + lr];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            u32x4_t bf[NS];
#pragma unroll
            for (int p = 0; p < NS; ++p) bf[p] = Bs[(size_t)(p * GB_KG + lq) * GB_BN + wn * 64 + ni * 16 + lr];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                if constexpr (NS == 3) {
                    acc[mi][ni] = mfma_x3(Bf3{af[mi][0], af[mi][1], af[mi][2]}, Bf3{bf[0], bf[1], bf[2]}, acc[mi][ni]);
                } else {
                    acc[mi][ni] = mfma_b16(af[mi][0], bf[0], acc[mi][ni]);
                }
            }
        }
    };

    fetch(rv0, r_begin);
    fetch(rv1, r_begin + GB_KC);                         // rows past r_end read as zeros
    for (int r0 = r_begin; r0 < r_end; r0 += 2 * GB_KC) {
        __syncthreads();                                 // the previous stage's fragments have been read
        stage(rv0);
        __syncthreads();
        fetch(rv0, r0 + 2 * GB_KC);
        compute();
        if (r0 + GB_KC >= r_end) break;
        __syncthreads();
        stage(rv1);
        __syncthreads();
        fetch(rv1, r0 + 3 * GB_KC);
        compute();
    }

    float* __restrict__ dw = a.dw[gemm];
    float* __restrict__ slot = a.slots ? a.slots + ((size_t)gemm * a.nsplit + split) * a.G * a.slot_kmax : nullptr;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int pb = wn * 64 + ni * 16 + lr;                           // LDS row of the X tile -> column k
            const int k = n0 + 4 * (pb % (GB_BN / 4)) + pb / (GB_BN / 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pa = wm * 64 + mi * 16 + lq * 4 + r;               // LDS row of the dG tile -> gate row g
                const int g = m0 + 4 * (pa % (GB_BM / 4)) + pa / (GB_BM / 4);
                if (g < a.G && k < K) {
                    if (slot) slot[(size_t)g * a.slot_kmax + k] = acc[mi][ni][r];
                    else unsafeAtomicAdd(dw + (size_t)g * K + k, acc[mi][ni][r]);
                }
            }
        }
    if (a.y_tile[blockIdx.y] == 0 && is_a && a.db[gemm]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (col + i < a.G) unsafeAtomicAdd(a.db[gemm] + col + i, bsum[i]);
    }
}

// Producer / consumer form of gru_wgrad_b16_kernel (same tile, LDS row format and output bijection): waves 0..3 only run
// MFMAs (2 x 2 waves, 64 gate rows x 128 columns = 32 accumulator tiles each, 192 MFMAs per 32-row step in the bf16x3 mode),
// waves 4..7 only fetch, split and stage - per 32-row step a producer lane converts one 8-row x 4-column block of X and one
// 8 x 2 block of dG, with the rows of the next two steps in flight in registers.  The two LDS stages alternate, one block
// barrier per step.  The non-specialised kernel above stops every wave for fetch -> convert -> barrier -> MFMA (31 % MFMA
// issue for the 8 x [768 x 256 x 16000] gradients of the FBCRNN stacks); here a SIMD's consumer wave issues MFMAs while its
// producer wave waits for loads.
template <int NS>
__global__ __launch_bounds__(512) void gru_wgrad_pc_kernel(GruWgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32x4_t smem_b16[];
    constexpr int STAGE = NS * GB_KG * (GB_BM + GB_BN);             // 16-byte rows per stage: [NS][KG][BM] of dG, [NS][KG][BN] of X
    constexpr int A_PART = GB_KG * GB_BM, B_PART = GB_KG * GB_BN, B_BASE = NS * A_PART;
    const int tid = threadIdx.x, lane = tid & 63, lq = lane >> 4, lr = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave < 4;
    const int wm = wave >> 1, wn = wave & 1;                          // consumer wave -> 64 gate rows x 128 columns
    // The gridDim.x row tiles of one (GEMM, column tile, split) read the same X rows: workgroups go round-robin over the 8
    // XCDs in linear order, so the linear id is re-read as (xcd, slot) and a whole row-tile group placed on one XCD - X comes
    // through that XCD's L2 once instead of once per row tile (6 x for the 768-row gradients)
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (a.xcd_groups) {
        const int L = bx + (int)gridDim.x * (by + (int)gridDim.y * bz);
        const int slot = L >> 3, grp = (L & 7) + 8 * (slot / (int)gridDim.x);
        bx = slot % (int)gridDim.x; by = grp % (int)gridDim.y; bz = grp / (int)gridDim.y;
    }
    const int gemm = a.y_gemm[by], split = bz, K = a.Ks[gemm];
    const int m0 = bx * GB_BM, n0 = a.y_tile[by] * GB_BN;
    const int r_begin = split * a.rows_per_split, r_end = min(a.TB, r_begin + a.rows_per_split);
    if (r_begin >= r_end) return;
    const int nSteps = (r_end - r_begin + GB_KC - 1) / GB_KC, nSteps2 = (nSteps + 1) & ~1;

    f32x4 acc[4][8];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[2] = {0.f, 0.f};
    int bcol = 0;

    if (!consumer) {
        // ================================================================ PRODUCER
        __builtin_amdgcn_s_setprio(3);                               // the consumers wait for these waves' stages, never the other way round (3 - 4 %)
        const int pt = tid - 256;
        const int kgb = pt >> 6, jqb = pt & 63;                      // X item: rows 8 kgb .. + 7, columns 4 jqb .. + 3
        const int ia = pt & 127, kga = ia >> 5, jqa = ia & 31, half = pt >> 7;     // dG item: columns 4 jqa + 2 half .. + 1
        const int colb = n0 + 4 * jqb, cola = m0 + 4 * jqa + 2 * half;
        bcol = cola;
        const bool b_ok = colb < K, a_ok = cola < a.G;
        const int shift = a.shift_rows[gemm];
        constexpr unsigned OOB = 0x80000000u;
        const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dg[gemm]), 0, (unsigned)((size_t)a.TB * a.G * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x[gemm]), 0, (unsigned)((size_t)a.TB * K * 4), 0x00020000);
        u32x4_t rb[2][8];
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        u32x2_t ra[2][8];
        // byte offsets stay below 2^31 (checked by the launcher); rows outside the split / the sequence read as zeros through
        // an out-of-range offset picked with masks (no branch per load).  The scheduling barriers keep every step's loads and
        // conversions in program order: the compiler's wait counts are exact only if the queue of outstanding loads looks the
        // same on every path into the loop (prologue and back edge), and a conversion that sinks below the next requests
        // costs a register copy of values still in flight, i.e. a full drain
        auto fetch = [&](int S, auto set_c) __attribute__((always_inline)) {
            constexpr int SET = decltype(set_c)::value;
            const int r0 = r_begin + S * GB_KC;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int r = r0 + kgb * 8 + rr, rs = r + shift;
                const unsigned ok = (unsigned)-(int)(b_ok & (r < r_end) & (rs >= 0) & (rs < a.TB));
                const unsigned off = (unsigned)(rs * K + colb) * 4u;
                rb[SET][rr] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, (off & ok) | (OOB & ~ok), 0, 0);
            }
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) {
                const int r = r0 + kga * 8 + rr;
                const unsigned ok = (unsigned)-(int)(a_ok & (r < r_end));
                const unsigned off = (unsigned)(r * a.G + cola) * 4u;
                ra[SET][rr] = __builtin_amdgcn_raw_buffer_load_b64(rs_a, (off & ok) | (OOB & ~ok), 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto put = [&](u32x4_t* dst, int part_rows, const float (&v)[8]) __attribute__((always_inline)) {
            if constexpr (NS == 3) {
                const Bf3 p = split3x8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]));
                dst[0] = p.hi; dst[part_rows] = p.mid; dst[2 * part_rows] = p.lo;
            } else {
                dst[0] = u32x4_t{pack_bf16_rne(v[0], v[1]), pack_bf16_rne(v[2], v[3]), pack_bf16_rne(v[4], v[5]), pack_bf16_rne(v[6], v[7])};
            }
        };
        auto stage = [&](int buf, auto set_c) __attribute__((always_inline)) {
            constexpr int SET = decltype(set_c)::value;
            u32x4_t* As = smem_b16 + buf * STAGE;
            u32x4_t* Bs = As + B_BASE;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v[8];
#pragma unroll
                for (int rr = 0; rr < 8; ++rr)
                    v[rr] = __uint_as_float(i == 0 ? rb[SET][rr].x : i == 1 ? rb[SET][rr].y : i == 2 ? rb[SET][rr].z : rb[SET][rr].w);
                put(Bs + kgb * GB_BN + i * (GB_BN / 4) + jqb, B_PART, v);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float v[8];
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) v[rr] = __uint_as_float(i == 0 ? ra[SET][rr].x : ra[SET][rr].y);
                bsum[i] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                put(As + kga * GB_BM + (2 * half + i) * (GB_BM / 4) + jqa, A_PART, v);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        // step S lives in register set S % 2 and LDS stage S % 2; its rows are requested two steps before they are converted
        fetch(0, I0{});
        fetch(1, I1{});                                              // rows past r_end read as zeros
        stage(0, I0{});
        fetch(2, I0{});
        __syncthreads();
        // consumers: step S; here: stage step S + 1, request step S + 3.  No conditions on the way (steps past the last one
        // fetch zeros into a stage nobody reads): where a path with and one without new loads meet, the compiler's wait
        // counts assume the fewer loads and drain the queue, which is the whole prefetch distance
        for (int S = 0; S < nSteps2; S += 2) {
            stage(1, I1{});
            fetch(S + 3, I1{});
            __syncthreads();
            stage(0, I0{});
            fetch(S + 4, I0{});
            __syncthreads();
        }
    } else {
        // ================================================================ CONSUMER
        __syncthreads();                                             // step 0 is staged
        for (int S = 0; S < nSteps2; ++S) {
            if (S >= nSteps) { __syncthreads(); continue; }          // the producers' loop runs an even number of steps
            const u32x4_t* As = smem_b16 + (S & 1) * STAGE;
            const u32x4_t* Bs = As + B_BASE;
            u32x4_t af[4][NS], bf[2][NS];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int p = 0; p < NS; ++p) af[mi][p] = As[p * A_PART + lq * GB_BM + wm * 64 + mi * 16 + lr];
#pragma unroll
            for (int p = 0; p < NS; ++p) bf[0][p] = Bs[p * B_PART + lq * GB_BN + wn * 128 + lr];
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) {
                const int cur = ni & 1;
                if (ni + 1 < 8) {
#pragma unroll
                    for (int p = 0; p < NS; ++p) bf[cur ^ 1][p] = Bs[p * B_PART + lq * GB_BN + wn * 128 + (ni + 1) * 16 + lr];
                }
                if constexpr (NS == 3) {
                    // six part products of the four row tiles round-robin, smallest first (Bf3 order: hi = 0, mid = 1, lo = 2)
#pragma unroll
                    for (int pp = 0; pp < 6; ++pp) {
                        const int pa = pp == 0 ? 2 : pp == 1 ? 0 : pp == 2 ? 1 : pp == 3 ? 1 : 0;
                        const int pb = pp == 0 ? 0 : pp == 1 ? 2 : pp == 2 ? 1 : pp == 3 ? 0 : pp == 4 ? 1 : 0;
#pragma unroll
                        for (int mi = 0; mi < 4; ++mi) acc[mi][ni] = mfma_b16(af[mi][pa], bf[cur][pb], acc[mi][ni]);
                    }
                } else {
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) acc[mi][ni] = mfma_b16(af[mi][0], bf[cur][0], acc[mi][ni]);
                }
            }
            __syncthreads();
        }
        float* __restrict__ dw = a.dw[gemm];
        float* __restrict__ slot = a.slots ? a.slots + ((size_t)gemm * a.nsplit + split) * a.G * a.slot_kmax : nullptr;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) {
                const int pb = wn * 128 + ni * 16 + lr;                          // LDS row of the X tile -> column k
                const int k = n0 + 4 * (pb % (GB_BN / 4)) + pb / (GB_BN / 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int pa = wm * 64 + mi * 16 + lq * 4 + r;               // LDS row of the dG tile -> gate row g
                    const int g = m0 + 4 * (pa % (GB_BM / 4)) + pa / (GB_BM / 4);
                    if (g < a.G && k < K) {
                        if (slot) slot[(size_t)g * a.slot_kmax + k] = acc[mi][ni][r];
                        else unsafeAtomicAdd(dw + (size_t)g * K + k, acc[mi][ni][r]);
                    }
                }
            }
    }
    if (!consumer && a.y_tile[by] == 0 && a.db[gemm]) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (bcol + i < a.G) unsafeAtomicAdd(a.db[gemm] + bcol + i, bsum[i]);
    }
}

__global__ void gru_wgrad_slot_reduce_kernel(GruWgradArgs a, int n) {
    const int gemm = blockIdx.y, K = a.Ks[gemm];
    const size_t total = (size_t)a.G * K;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int g = (int)(i / K), k = (int)(i % K);
        const float* p = a.slots + (size_t)gemm * a.nsplit * a.G * a.slot_kmax + (size_t)g * a.slot_kmax + k;
        float v = 0.f;
        for (int sp = 0; sp < a.nsplit; ++sp) v += p[(size_t)sp * a.G * a.slot_kmax];
        a.dw[gemm][(size_t)g * a.Ks[gemm] + k] += v;
    }
}

}  // namespace pbsed

using namespace pbsed;

// device scratch of the slot mode: the caller's registered buffer for (device, stream) or the library's per-device one (api.hip)
// (behind the front part that the slotted conv weight gradients keep zero)
static float* gru_wgrad_scratch(size_t floats, hipStream_t s) {
    float* base = scratch_for(s, PBSED_SCRATCH_FRONT + floats);
    return base ? base + PBSED_SCRATCH_FRONT : nullptr;
}

static int gru_wgrad_launch(int n, const float* const* dg, const float* const* x, const int* shift, float* const* dw,
                            float* const* db, int T, int B, int G, const int* Ks, int operands, void* stream) {
    if (n < 1 || n > GW_MAX || T < 1 || B < 1 || G < 4 || (G & 3)) {
        set_error("gru_wgrad: need 1 <= n <= %d, G a multiple of 4 (n=%d G=%d)", GW_MAX, n, G);
        return PBSED_E_ARG;
    }
    if (operands != 0 && operands != 1 && operands != 3) { set_error("gru_wgrad: operands %d (0 = f32, 1 = bf16, 3 = bf16x3)", operands); return PBSED_E_ARG; }
    GruWgradArgs a{};
    int kmax = 0;
    for (int i = 0; i < n; ++i) {
        if (Ks[i] < 4 || (Ks[i] & 3)) { set_error("gru_wgrad: K must be a multiple of 4 (K[%d]=%d)", i, Ks[i]); return PBSED_E_ARG; }
        if (!operands && Ks[i] != Ks[0]) { set_error("gru_wgrad: the fp32-MFMA kernel takes one K per launch"); return PBSED_E_UNSUPPORTED; }
        a.dg[i] = dg[i]; a.x[i] = x[i]; a.dw[i] = dw[i]; a.db[i] = db ? db[i] : nullptr;
        a.shift_rows[i] = shift[i] * B;
        a.Ks[i] = Ks[i];
        kmax = Ks[i] > kmax ? Ks[i] : kmax;
    }
    const int K = Ks[0];
    a.TB = T * B; a.G = G; a.K = K;
    hipStream_t s = (hipStream_t)stream;
    const int n_cu = device_cus();
    if (operands) {
        if ((size_t)a.TB * (G > kmax ? G : kmax) * 4 >= (1ull << 31)) { set_error("gru_wgrad: an operand of %d x %d floats exceeds the 2 GiB the loaders address", a.TB, G > kmax ? G : kmax); return PBSED_E_ARG; }
        int ny = 0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < (Ks[i] + GB_BN - 1) / GB_BN; ++j) {
                if (ny >= 4 * GW_MAX) { set_error("gru_wgrad: more than %d column tiles in one launch", 4 * GW_MAX); return PBSED_E_ARG; }
                a.y_gemm[ny] = (unsigned char)i; a.y_tile[ny] = (unsigned char)j; ++ny;
            }
        // one block per CU: split the (t, b) reduction to one residency round
        dim3 grid((G + GB_BM - 1) / GB_BM, ny, 1);
        const int tiles = grid.x * ny;
        int nsplit = n_cu / tiles;
        const int max_split = (a.TB + 4 * GB_KC - 1) / (4 * GB_KC);
        if (nsplit < 1) nsplit = 1;
        // one round that leaves more than 15 % of the CUs without a block (2 x 512-wide stacks: 192 tiles on 256 CUs): a few
        // rounds of shorter blocks instead - the split count with the least rounds x rows
        if (tiles * nsplit * 100 < n_cu * 85) {
            int best = nsplit;
            for (int ns = nsplit + 1; ns <= 8 && ns <= max_split; ++ns) {
                const int r_ns = (tiles * ns + n_cu - 1) / n_cu, r_b = (tiles * best + n_cu - 1) / n_cu;
                if (r_ns * best < r_b * ns) best = ns;
            }
            nsplit = best;
        }
        if (nsplit > max_split) nsplit = max_split;
        a.rows_per_split = ((a.TB + nsplit - 1) / nsplit + GB_KC - 1) / GB_KC * GB_KC;
        a.nsplit = (a.TB + a.rows_per_split - 1) / a.rows_per_split;
        grid.z = a.nsplit;
        // split reductions: every (GEMM, split) writes its partial tile to a slot with plain stores and one pass adds the slots
        // up, instead of one float atomic per split and gradient element (8 x [768 x 256], 5 splits: 7.9 M atomics on
        // addresses shared by blocks of different XCDs = 40 - 70 us of a 440 us launch; a [256 x 256] x 3-tap gradient over 16 000
        // rows with 42 splits: 0.46 ms with atomics).  PBSED_GRU_WGRAD_SLOT_MIN: smallest split count that takes the slots
        static const int slot_min = [] { const char* e = getenv("PBSED_GRU_WGRAD_SLOT_MIN"); return e ? atoi(e) : 1; }();
        a.slots = nullptr;
        if (a.nsplit > slot_min && slot_min > 0) {
            a.slot_kmax = kmax;
            const size_t need = (size_t)n * a.nsplit * G * kmax;
            if (need * sizeof(float) <= (1ull << 30)) a.slots = gru_wgrad_scratch(need, s);
        }
        const size_t lds = (size_t)operands * GB_KG * (GB_BM + GB_BN) * sizeof(u32x4_t);
        // PBSED_GRU_WGRAD_PC (default 1): the producer / consumer kernel (2: without the XCD placement); 0: the non-specialised one
        static const int pc = [] { const char* e = getenv("PBSED_GRU_WGRAD_PC"); return e ? atoi(e) : 1; }();
        a.xcd_groups = (pc == 1 && (ny * a.nsplit) % 8 == 0) ? 1 : 0;
        if (pc && operands == 3) {
            PBSED_DYN_LDS_ONCE(gru_wgrad_pc_kernel<3>, 2 * lds);
            hipLaunchKernelGGL(gru_wgrad_pc_kernel<3>, grid, dim3(512), 2 * lds, s, a);
        } else if (pc) {
            PBSED_DYN_LDS_ONCE(gru_wgrad_pc_kernel<1>, 2 * lds);
            hipLaunchKernelGGL(gru_wgrad_pc_kernel<1>, grid, dim3(512), 2 * lds, s, a);
        } else if (operands == 3) {
            PBSED_DYN_LDS_ONCE(gru_wgrad_b16_kernel<3>, lds);
            hipLaunchKernelGGL(gru_wgrad_b16_kernel<3>, grid, dim3(512), lds, s, a);
        } else {
            PBSED_DYN_LDS_ONCE(gru_wgrad_b16_kernel<1>, lds);
            hipLaunchKernelGGL(gru_wgrad_b16_kernel<1>, grid, dim3(512), lds, s, a);
        }
        if (a.slots) {
            const size_t per = (size_t)G * kmax;
            hipLaunchKernelGGL(gru_wgrad_slot_reduce_kernel, dim3((unsigned)((per + 255) / 256 < 1024 ? (per + 255) / 256 : 1024), n),
                               dim3(256), 0, s, a, n);
        }
        return check_launch("gru_wgrad");
    }
    const bool wide = K > 128;
    const int bn = wide ? 256 : 128;
    dim3 grid((G + GW_BM - 1) / GW_BM, (K + bn - 1) / bn, 1);
    // split the (t,b) reduction so that the launch is one full residency round (blocks per CU from the occupancy
    // query; measured on MI355X: 768 blocks 0.48 ms, 512 blocks 0.62 ms, 1024 blocks 0.56 ms for 8 x [768 x 256 x 16000])
    const int tiles = grid.x * grid.y * n;
    static int target = 0;
    if (target == 0) {
        int occ = 0;
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
    if (wide) hipLaunchKernelGGL((gru_wgrad_kernel<256>), grid, dim3(512), 0, s, a);
    else hipLaunchKernelGGL((gru_wgrad_kernel<128>), grid, dim3(256), 0, s, a);
    return check_launch("gru_wgrad");
}

static int x3_default() {
    static const int x3 = [] { const char* e = getenv("PBSED_GRU_WGRAD_X3"); return e ? atoi(e) : 1; }();
    return x3 ? 3 : 0;
}

// PBSED_GRU_WGRAD_X3 (default 1): the fp32 entry points run the bf16x3 kernel (fp32-class results, see gru_wgrad_b16_kernel)
extern "C" int pbsed_gru_wgrad(int n, const float* const* dg, const float* const* x, const int* shift, float* const* dw,
                               float* const* db, int T, int B, int G, int K, void* stream) {
    int Ks[GW_MAX];
    for (int i = 0; i < GW_MAX; ++i) Ks[i] = K;
    return gru_wgrad_launch(n, dg, x, shift, dw, db, T, B, G, Ks, x3_default(), stream);
}

extern "C" int pbsed_gru_wgrad_multi(int n, const float* const* dg, const float* const* x, const int* shift, float* const* dw,
                                     float* const* db, int T, int B, int G, const int* K, int bf16, void* stream) {
    if (!K) { set_error("gru_wgrad_multi: K is null"); return PBSED_E_ARG; }
    return gru_wgrad_launch(n, dg, x, shift, dw, db, T, B, G, K, bf16 ? 1 : x3_default(), stream);
}
