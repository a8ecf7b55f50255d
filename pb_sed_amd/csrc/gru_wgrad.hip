// Weight/bias gradients of the GRU stacks straight from the scans' time-major buffers:
//
//     dW[g][k] += sum_{t,b} dG[t][b][g] * X[t + shift][b][k]        db[g] += sum_{t,b} dG[t][b][g]
//
// which is what autograd accumulates into weight_ih / weight_hh / bias_* of torch.nn.GRU
// (pb_sed/models/base.py:64-68 -> padertorch GRU wrapper -> nn.GRU backward).  dG is the scan's dgi or dgh
// [T*B, G], X the layer input or the layer's own state sequence [T*B, K] (shift = -1 / +1 picks h_{t-1} of a
// forward / time-reversed chain; rows shifted outside [0, T) are zero).  Both operands have the reduction
// index (t, b) as the slow dimension, so rows are fetched with coalesced loads and converted while they are
// staged (bf16x3: exact three-way splits, fp32-class gradients; bf16: the bf16 training mode).  All GEMMs of a
// backward pass (2 per chain and layer) go in one launch; the reduction is split over blocks and combined
// through slots + one reduction pass (or float atomics for a single split).
#include <cstdlib>

#include <type_traits>

#include "common.h"
#include "pbsed_internal.h"

namespace pbsed {

constexpr int GW_MAX = 16;     // GEMMs per launch

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


// The same GEMMs on the bf16 MFMA (16x16x32).  NS = 3: every fp32 operand is split exactly into three bf16 parts while it
// is staged (Bf3 in common.h) and the six part products above 2^-24 are accumulated - fp32-class gradients at up to 2.6x
// the fp32-MFMA rate; NS = 1: plain bf16 operands (the bf16 training mode of BASELINE config 3), HBM-bound.
// The contraction index (t, b) is the slow dimension of both operands, the MFMA wants 8 consecutive contraction values per
// lane: a staging thread therefore fetches an 8-row x 4-column block (8 coalesced float4 rows), converts it, and writes
// each column's 8 values as ONE 16-byte LDS row.  LDS rows are stored at pos(c) = (c % 4) * (W / 4) + c / 4 of the tile's W
// columns, so that the four rows a thread writes per part are conflict-free and an MFMA tile = 16 consecutive rows
// (= columns c with the same c % 4, stride 4); the output indices follow the same bijection.
constexpr int GB_BM = 128, GB_BN = 256, GB_KC = 32, GB_KG = GB_KC / 8;


// Producer / consumer form (round 3; the non-specialised round-2 kernel it replaced stopped every wave for fetch -> convert ->
// barrier -> MFMA: 31 % MFMA issue): waves 0..3 only run
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
    if (operands != 1 && operands != 3) { set_error("gru_wgrad: operands %d (1 = bf16, 3 = bf16x3)", operands); return PBSED_E_ARG; }
    GruWgradArgs a{};
    int kmax = 0;
    for (int i = 0; i < n; ++i) {
        if (Ks[i] < 4 || (Ks[i] & 3)) { set_error("gru_wgrad: K must be a multiple of 4 (K[%d]=%d)", i, Ks[i]); return PBSED_E_ARG; }
        a.dg[i] = dg[i]; a.x[i] = x[i]; a.dw[i] = dw[i]; a.db[i] = db ? db[i] : nullptr;
        a.shift_rows[i] = shift[i] * B;
        a.Ks[i] = Ks[i];
        kmax = Ks[i] > kmax ? Ks[i] : kmax;
    }
    const int K = Ks[0];
    a.TB = T * B; a.G = G; a.K = K;
    hipStream_t s = (hipStream_t)stream;
    const int n_cu = launch_cus();
    {
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
        // rows with 42 splits: 0.46 ms with atomics)
        a.slots = nullptr;
        if (a.nsplit > 1) {
            a.slot_kmax = kmax;
            const size_t need = (size_t)n * a.nsplit * G * kmax;
            if (need * sizeof(float) <= (1ull << 30)) a.slots = gru_wgrad_scratch(need, s);
        }
        const size_t lds = (size_t)operands * GB_KG * (GB_BM + GB_BN) * sizeof(u32x4_t);
        a.xcd_groups = ((ny * a.nsplit) % 8 == 0) ? 1 : 0;      // row-tile groups of a (GEMM, column tile, split) on one XCD
        if (operands == 3) {
            PBSED_DYN_LDS_ONCE(gru_wgrad_pc_kernel<3>, 2 * lds);
            hipLaunchKernelGGL(gru_wgrad_pc_kernel<3>, grid, dim3(512), 2 * lds, s, a);
        } else {
            PBSED_DYN_LDS_ONCE(gru_wgrad_pc_kernel<1>, 2 * lds);
            hipLaunchKernelGGL(gru_wgrad_pc_kernel<1>, grid, dim3(512), 2 * lds, s, a);
        }
        if (a.slots) {
            const size_t per = (size_t)G * kmax;
            hipLaunchKernelGGL(gru_wgrad_slot_reduce_kernel, dim3((unsigned)((per + 255) / 256 < 1024 ? (per + 255) / 256 : 1024), n),
                               dim3(256), 0, s, a, n);
        }
        return check_launch("gru_wgrad");
    }
}

// the fp32 entry points run the bf16x3 kernel (fp32-class results)
extern "C" int pbsed_gru_wgrad(int n, const float* const* dg, const float* const* x, const int* shift, float* const* dw,
                               float* const* db, int T, int B, int G, int K, void* stream) {
    int Ks[GW_MAX];
    for (int i = 0; i < GW_MAX; ++i) Ks[i] = K;
    return gru_wgrad_launch(n, dg, x, shift, dw, db, T, B, G, Ks, 3, stream);
}

extern "C" int pbsed_gru_wgrad_multi(int n, const float* const* dg, const float* const* x, const int* shift, float* const* dw,
                                     float* const* db, int T, int B, int G, const int* K, int bf16, void* stream) {
    if (!K) { set_error("gru_wgrad_multi: K is null"); return PBSED_E_ARG; }
    return gru_wgrad_launch(n, dg, x, shift, dw, db, T, B, G, K, bf16 ? 1 : 3, stream);
}
