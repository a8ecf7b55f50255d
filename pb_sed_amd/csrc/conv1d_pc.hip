// Conv1d forward / data gradient (kernel sizes 1 and 3, zero padding, F = 1 rows) of the fp32 path on the bf16 MFMA with
// EXACT three-way bf16 operand splits (fp32-class results), PRODUCER / CONSUMER form, for gfx950.
// Op sites: the CNN1d layers and the per-frame output nets of pb_sed/models/weak_label/crnn.py:93-101 and
// pb_sed/models/strong_label/crnn.py:88-104 (padertorch CNN1d / fully_connected_stack as 1 x 1 convolutions), backward of
// pb_sed/experiments/weak_label_crnn/training.py:159-169.  Same tensors, prologue (BN-apply + ReLU + mask of the previous
// layer) and epilogue fusions (bias, masked batch statistics; data gradient: backward through mask / ReLU / BN-apply with
// the BN-backward sums) as the pipelined kernel of conv_bf16.hip that it replaces for these layers; what changes is the
// structure (conv_winox3.hip's, without the transform):
//
//   y[cout, t] = sum_{kw, cin} W[kw][cout, cin] * x[cin, t + kw - 1]          M = cout, N = t, K = 32 cin per MFMA
//
// Block = 512 threads = 4 CONSUMER + 4 PRODUCER waves (one of each per SIMD), tile = 128 cout x 128 t:
//  * consumer wave w owns 32 cout x 128 t (64 accumulator registers).  Its A operands - the pre-split weights - are not
//    staged: the pack writes them in fragment order ([chunk][kw][cout tile][part][lane] x 16 B) and the wave streams its own
//    6 KB per (chunk, tap) from L2 into a register ring of three steps (loads issued two steps = 192 MFMAs ahead).
//  * producer waves turn x into the three-part LDS image of a 32-channel chunk: lane = time position (256-byte coalesced
//    dword loads; any T), prologue, truncation split, one 8-byte store per (4 channels, position, part) into
//    [position 130][32 cin] rows of 64 bytes with XOR-swizzled 16-byte groups - a B fragment of any tap is one conflict-free
//    ds_read_b128 per part at a position offset.  Images are double-buffered: one block barrier per chunk.
// Blocks are PERSISTENT (one per CU) and walk (cout tile, time tile) items; the cout tiles of a time tile run back to back
// on one XCD, so x comes from HBM once.  The producers run ahead through the chunk stream of all of a block's tiles.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "pack_elems.h"
#include "pbsed_internal.h"

namespace pbsed {

constexpr int C1_CT = 128;                               // cout per block (32 per consumer wave)
constexpr int C1_CK = 32;                                // cin per chunk = K of one MFMA
constexpr int C1_TT = 128;                               // t per block
constexpr int C1_POS = C1_TT + 2;                        // image positions: t0 - 1 .. t0 + 128
constexpr int C1_PART = C1_POS * 64;                     // bytes of one part of an image: [position][32 cin bf16]
constexpr int C1_IMG = 3 * C1_PART;
constexpr int C1_LDS = 2 * C1_IMG;                       // 49 920
constexpr int C1_TRS = 132;                              // floats per row of a consumer wave's epilogue tile [32 cout][128 t + 4]: the
constexpr int C1_LDS_ALL = C1_LDS + 4 * 32 * C1_TRS * 4;  // accumulator-layout writes and the row reads are both conflict-free
#ifndef C1_DBG
#define C1_DBG 0            // ablation switches of tools/kernel_ablation.sh (never set in the product build)
#endif

__global__ void c1x3_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ up, int Cout, int Cin, int KW, int InP,
                                 int OutP, int dgrad) {
    const size_t total = (size_t)KW * InP * OutP;               // weights; three halfwords each
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        pack_c1x3_value(w, up, i, Cout, Cin, KW, InP, OutP, dgrad);
}

template <int KW, bool DGRAD>
__global__ __launch_bounds__(512) void conv1d_pc_kernel(ConvFwdArgs a, int nCt, int nSp, int nWork) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // uniform: scalar offsets and role branches depend on it
    const bool consumer = wave < 4;
    const int lq = lane >> 4, lr = lane & 15;

    // PERSISTENT blocks: block p works on items p, p + gridDim, ..  Item w -> XCD w % 8 (gridDim is a multiple of 8); the cout
    // tiles of a time tile follow each other on ONE XCD: x is fetched from HBM once and found in that XCD's L2 by the others.
    const int nTt = (a.T + C1_TT - 1) / C1_TT;
    auto item = [&](int k, int& ct, int& sp) __attribute__((always_inline)) {
        const int w = (int)blockIdx.x + k * (int)gridDim.x;
        const int xcd = w & 7, l = w >> 3;
        ct = l % nCt;
        sp = (l / nCt) * 8 + xcd;
        return w < nWork && sp < nSp;
    };
    int nT = 0;                                       // valid items of this block (sp grows with k: the invalid ones are at the end)
    {
        int ct_, sp_;
        while (item(nT, ct_, sp_)) ++nT;
    }
    if (nT == 0) return;
    const bool pro = a.scale != nullptr;
    const int nChunks = a.CinP / C1_CK;
    const int G = nT * nChunks;                       // the block's stream of 32-channel chunks over all of its tiles
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;

    if (!consumer) {
        // ================================================================ PRODUCER: x -> three bf16 parts in LDS
        // Thread = (producer wave pw, lane): four items (4 consecutive cin, one position) - cin group 2 pw + ((lane >> 4) & 1),
        // position (lane & 15) + 16 (lane >> 5) + 32 i of the tile; the wave's 8 channels x 128 positions.  Lanes 0..31 of a store
        // = 16 consecutive positions x the two 8-byte halves of one 16-byte k-group: all 64 banks, no conflicts.  Lanes 0..15
        // also fetch the two positions outside the tile (t0 - 1: lanes 0..7, t0 + 128: lanes 8..15; channel 8 pw + (lane & 7))
        // for the 3-tap kernels.
        const int pw = wave - 4;
        constexpr unsigned OOB = 0x80000000u;
        const unsigned clip_elems = (unsigned)(a.Cin * a.T);
        const __amdgpu_buffer_rsrc_t rs_sc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.scale), 0, pro ? (unsigned)a.Cin * 4u : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_sh = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.shift), 0, pro ? (unsigned)a.Cin * 4u : 0u, 0x00020000);
        const int half = (lane >> 4) & 1, pos0 = (lane & 15) + 16 * (lane >> 5);
        unsigned lds_i[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pos = pos0 + 32 * i + 1;
            lds_i[i] = (unsigned)(pos * 64 + (((pw ^ ((-(pos >> 2)) & 3)) & 3) * 16) + half * 8);
        }
        const int hpos = lane < 8 ? 0 : C1_POS - 1;
        const unsigned lds_h = (unsigned)(hpos * 64 + (((pw ^ ((-(hpos >> 2)) & 3)) & 3) * 16) + (lane & 7) * 2);
        const float relu_floor = (pro && a.relu) ? 0.f : -__builtin_inff();

        int ld_k = 0, ld_ch = 0, ld_t0 = 0, ld_b = 0, ld_tlim = 0;
        constexpr int NB = 2;                                  // raw register sets = chunks whose loads are in flight
        unsigned rin[NB][4][4], rhalo[NB], rsc[NB], rsh[NB];
        int fi_t0[NB], fi_tlim[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) { rhalo[n] = 0u; rsc[n] = 0u; rsh[n] = 0u; fi_t0[n] = 0; fi_tlim[n] = 0; }

        auto load_chunk = [&](auto buf_c) __attribute__((always_inline)) {
            constexpr int BUF = decltype(buf_c)::value;
            if (ld_ch == 0) {                                      // entering tile ld_k
                int ct, sp;
                item(ld_k, ct, sp);
                ld_t0 = (sp % nTt) * C1_TT;
                ld_b = sp / nTt;
                const int sl = a.seq_len ? min(a.seq_len[ld_b], a.T) : a.T;
                ld_tlim = pro ? sl : a.T;                          // Normalization re-masks its output (y * mask)
            }
            fi_t0[BUF] = ld_t0; fi_tlim[BUF] = ld_tlim;
            const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(a.x) + (size_t)ld_b * clip_elems, 0, clip_elems * 4u, 0x00020000);
            const unsigned cw = (unsigned)(ld_ch * C1_CK + 8 * pw);       // first of the wave's 8 channels
            rsc[BUF] = __builtin_amdgcn_raw_buffer_load_b32(rs_sc, (cw + (unsigned)(lane & 7)) * 4u, 0, 0);     // no prologue: empty
            rsh[BUF] = __builtin_amdgcn_raw_buffer_load_b32(rs_sh, (cw + (unsigned)(lane & 7)) * 4u, 0, 0);     // ranges, zeros
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = ld_t0 + pos0 + 32 * i;
                const unsigned ok = (unsigned)-(int)(t < a.T);
                const unsigned e0 = (cw + 4u * (unsigned)half) * (unsigned)a.T + (unsigned)t;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const unsigned off = (((e0 + (unsigned)(c * a.T)) * 4u) & ok) | (OOB & ~ok);
                    rin[BUF][i][c] = __builtin_amdgcn_raw_buffer_load_b32(rs_x, off, 0, 0);
                }
            }
            if (KW == 3) {
                const int t = lane < 8 ? ld_t0 - 1 : ld_t0 + C1_TT;
                const unsigned ok = (unsigned)-(int)(lane < 16 && t >= 0 && t < a.T);
                const unsigned e = (cw + (unsigned)(lane & 7)) * (unsigned)a.T + (unsigned)t;
                rhalo[BUF] = __builtin_amdgcn_raw_buffer_load_b32(rs_x, ((e * 4u) & ok) | (OOB & ~ok), 0, 0);
            }
            if (!(C1_DBG & 16) && ++ld_ch == nChunks) { ld_ch = 0; ++ld_k; }     // ablation bit 16: every chunk re-reads the first one
        };
        auto store_chunk = [&](int gdst, auto buf_c) __attribute__((always_inline)) {
            constexpr int BUF = decltype(buf_c)::value;
            unsigned char* img = smem_raw + (gdst & 1) * C1_IMG;
            const float vsc = pro ? __uint_as_float(rsc[BUF]) : 1.f, vsh = __uint_as_float(rsh[BUF]);   // no prologue: x * 1 + 0
            const int tl = fi_tlim[BUF] - fi_t0[BUF];               // positions of the tile inside the (masked) sequence
            float sc[4], sh[4];                                     // lane k < 8 of this wave holds channel k's scale / shift
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float sc0 = __uint_as_float(__builtin_amdgcn_readlane((int)__float_as_uint(vsc), c));
                const float sc1 = __uint_as_float(__builtin_amdgcn_readlane((int)__float_as_uint(vsc), 4 + c));
                const float sh0 = __uint_as_float(__builtin_amdgcn_readlane((int)__float_as_uint(vsh), c));
                const float sh1 = __uint_as_float(__builtin_amdgcn_readlane((int)__float_as_uint(vsh), 4 + c));
                sc[c] = half ? sc1 : sc0;
                sh[c] = half ? sh1 : sh0;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool live = pos0 + 32 * i < tl;
                float v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float u = fmaxf(fmaf(__uint_as_float(rin[BUF][i][c]), sc[c], sh[c]), relu_floor);
                    v[c] = live ? u : 0.f;                              // zero padding is post-activation
                }
                unsigned h0, m0, l0, h1, m1, l1;
                split3_pair(v[0], v[1], h0, m0, l0);
                split3_pair(v[2], v[3], h1, m1, l1);
                unsigned char* p = img + lds_i[i];
                *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(p + C1_PART) = make_uint2(m0, m1);
                *reinterpret_cast<uint2*>(p + 2 * C1_PART) = make_uint2(l0, l1);
            }
            if (KW == 3 && lane < 16) {
                const int t = lane < 8 ? fi_t0[BUF] - 1 : fi_t0[BUF] + C1_TT;
                float u = fmaxf(fmaf(__uint_as_float(rhalo[BUF]), vsc, vsh), relu_floor);   // lanes 0..7 and 8..15 hold channel lane & 7
                u = (t >= 0 && t < fi_tlim[BUF]) ? u : 0.f;
                const unsigned u0 = __float_as_uint(u);
                const float r1 = u - __uint_as_float(u0 & 0xffff0000u);
                const unsigned u1 = __float_as_uint(r1);
                const float r2 = r1 - __uint_as_float(u1 & 0xffff0000u);
                unsigned char* p = img + lds_h;
                *reinterpret_cast<unsigned short*>(p) = (unsigned short)(u0 >> 16);
                *reinterpret_cast<unsigned short*>(p + C1_PART) = (unsigned short)(u1 >> 16);
                *reinterpret_cast<unsigned short*>(p + 2 * C1_PART) = (unsigned short)(__float_as_uint(r2) >> 16);
            }
        };

        constexpr bool P_LD = !(C1_DBG & 1), P_ST = !(C1_DBG & 2);
        // chunk c of the stream lives in raw set c % 2 and is re-loaded with chunk c + 2 as soon as it is staged; image c % 2
        if (P_LD) load_chunk(I0{});
        if (G > 1 && P_LD) load_chunk(I1{});
        if (P_ST) store_chunk(0, I0{});
        if (G > 2 && P_LD) load_chunk(I0{});
        __syncthreads();
        auto step = [&](int g, auto nxt_c) __attribute__((always_inline)) {     // consumers: chunk g; stage chunk g + 1 (raw set NXT)
            using NXT = decltype(nxt_c);
            if (g + 1 < G) {
                if (P_ST) store_chunk(g + 1, NXT{});
                if (g + 3 < G && P_LD) load_chunk(NXT{});
            }
            __syncthreads();
        };
        for (int g = 0; g < G; g += 2) {
            step(g, I1{});
            if (g + 1 < G) step(g + 1, I0{});
        }
        return;
    }

    // ================================================================ CONSUMER: W from L2, x from LDS, MFMAs, epilogue
    const int MT = a.CoutP / 16;
    const size_t u_bytes = (size_t)KW * a.CinP * a.CoutP * 3 * 2;
    const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, (unsigned)u_bytes, 0x00020000);
    const unsigned voff_u = (unsigned)lane * 16u;
    const unsigned step_bytes = (unsigned)MT * 3072u;               // one (chunk, tap) of all cout tiles
    const int tile_steps = nChunks * KW;
    unsigned b_lane[KW];                                             // B fragment of tap kw: positions 16 n + lr + kw (kw = 1 for KW = 1)
#pragma unroll
    for (int kw = 0; kw < KW; ++kw) {
        const int p = lr + (KW == 1 ? 1 : kw);
        b_lane[kw] = (unsigned)(p * 64 + (((lq ^ ((-(p >> 2)) & 3)) & 3) * 16));
    }

    u32x4_t A[3][2][3];                                              // [ring slot][m][part]
    auto load_A = [&](auto slot_c, unsigned soff) __attribute__((always_inline)) {
        constexpr int SLOT = decltype(slot_c)::value;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                A[SLOT][m][p] = (C1_DBG & 4) ? u32x4_t{0u, 0u, 0u, 0u} : __builtin_amdgcn_raw_buffer_load_b128(rs_u, voff_u, soff + (unsigned)(m * 3 + p) * 1024u, 0);
    };
    // prefetch cursor: the (tile, step) whose weights are loaded next, two steps ahead of the MFMAs
    int pf_k = 0, pf_left = tile_steps;
    unsigned pf_soff;
    {
        int ct, sp;
        item(0, ct, sp);
        pf_soff = (unsigned)(ct * (C1_CT / 16) + wave * 2) * 3072u;
    }
    auto pf_advance = [&]() __attribute__((always_inline)) {
        pf_soff += step_bytes;
        if (--pf_left == 0) {
            int ct, sp;
            ++pf_k;
            const bool more = pf_k < nT && item(pf_k, ct, sp);
            pf_soff = more ? (unsigned)(ct * (C1_CT / 16) + wave * 2) * 3072u : 0xC0000000u;     // past the end: reads 0
            pf_left = more ? tile_steps : 0x7fffffff;
        }
    };
    load_A(I0{}, pf_soff); pf_advance();
    load_A(I1{}, pf_soff); pf_advance();

    f32x4 acc[2][8];
    float biasv[2][4];                                               // fetched at a tile's start: the epilogue's wait for them does not
    int k = 0, ch = 0, ct, sp;                                       // drain the weight prefetch queue
    item(0, ct, sp);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.bias), 0, a.bias ? (unsigned)a.Cout * 4u : 0u, 0x00020000);

    // BN-backward epilogue of the data gradient: the layer's raw input of the wave's 32 x 128 tile (lane = 4 consecutive t, half a
    // wave per row, 16 rows pairs) and the per-channel constants are requested at the START of a tile's last chunk, a chunk
    // step ahead of the epilogue that uses them - the epilogue no longer begins with a round trip to HBM
    const bool bnb = DGRAD && a.bx != nullptr;
    u32x4_t xq[DGRAD ? 16 : 1];
    float p_sc = 0.f, p_sh = 0.f, p_mu = 0.f, p_is = 0.f;
    auto request_bx = [&]() __attribute__((always_inline)) {
        constexpr unsigned OOB_C = 0x80000000u;
        const int t0 = (sp % nTt) * C1_TT, b = sp / nTt, cout0 = ct * C1_CT;
        const unsigned oclip = (unsigned)(a.Cout * a.T);
        const __amdgpu_buffer_rsrc_t rs_bx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.bx) + (size_t)b * oclip, 0, oclip * 4u, 0x00020000);
        const int half = lane >> 5, tq = t0 + 4 * (lane & 31);
        const bool vec = (a.T & 3) == 0;
        const int n_T = min(max(a.T - tq, 0), 4);
        const int c = cout0 + wave * 32 + (lane & 31);
        p_sc = p_sh = p_mu = p_is = 0.f;
        if (c < a.Cout) { p_sc = a.bscale[c]; p_sh = a.bshift[c]; p_mu = a.bmean[c]; p_is = a.binvstd[c]; }
#pragma unroll
        for (int i = 0; i < (DGRAD ? 16 : 1); ++i) {
            const int cout = cout0 + wave * 32 + 2 * i + half;
            const unsigned e = (unsigned)(cout * a.T + tq);
            if (vec) {
                const unsigned ok = (unsigned)-(int)(cout < a.Cout && n_T > 0);
                xq[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_bx, ((e * 4u) & ok) | (OOB_C & ~ok), 0, 0);
            } else {
                unsigned w[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned ok = (unsigned)-(int)(cout < a.Cout && k < n_T);
                    w[k] = __builtin_amdgcn_raw_buffer_load_b32(rs_bx, (((e + (unsigned)k) * 4u) & ok) | (OOB_C & ~ok), 0, 0);
                }
                xq[i] = u32x4_t{w[0], w[1], w[2], w[3]};
            }
        }
    };

    auto epilogue = [&]() __attribute__((always_inline)) {
        // Accumulator layout: t = t0 + 16 n + lr, cout = cout0 + 32 wave + 16 m + 4 lq + r - a store instruction would cover four
        // 64-byte row pieces.  The wave's 32 x 128 tile goes through its own LDS region instead and leaves as whole 512-byte rows
        // (lane = 4 consecutive t, half a wave per row: 16 stores of 16 bytes per lane instead of 64 of 4; the BN-backward
        // epilogue reads the layer's raw input the same way).  A wave owns its 32 channels alone: no cross-wave pass.
        constexpr unsigned OOB_C = 0x80000000u;
        const int t0 = (sp % nTt) * C1_TT, b = sp / nTt, cout0 = ct * C1_CT;
        const int sl = a.seq_len ? min(a.seq_len[b], a.T) : a.T;
        const unsigned oclip = (unsigned)(a.Cout * a.T);
        const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)b * oclip, 0, oclip * 4u, 0x00020000);
        const int slot = (int)(sp & (PBSED_STAT_SLOTS - 1));
        float* tr = reinterpret_cast<float*>(smem_raw + C1_LDS) + wave * (32 * C1_TRS);
        if (!DGRAD && a.stats) {
            // forward statistics (sum y, sum y^2 over the masked sequence) in the accumulator layout: one lane per channel adds
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int cout = cout0 + wave * 32 + m * 16 + lq * 4 + r;
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int n = 0; n < 8; ++n) {
                        const float v = acc[m][n][r] + biasv[m][r];
                        const float vm = t0 + n * 16 + lr < sl ? v : 0.f;
                        s1 += vm; s2 = fmaf(vm, vm, s2);
                    }
                    s1 = wave_sum16(s1);
                    s2 = wave_sum16(s2);
                    if (lr == 0 && cout < a.Cout) {
                        double* dst = a.stats + ((size_t)slot * a.Cout + cout) * 2;
                        atomicAdd(dst, (double)s1);
                        atomicAdd(dst + 1, (double)s2);
                    }
                }
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 8; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) tr[(m * 16 + lq * 4 + r) * C1_TRS + n * 16 + lr] = acc[m][n][r] + biasv[m][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int half = lane >> 5, tq = t0 + 4 * (lane & 31);            // row parity of the lane, its first t
        const bool vec = (a.T & 3) == 0;                                  // rows 16-byte aligned
        const int n_T = min(max(a.T - tq, 0), 4), n_sl = min(max(sl - tq, 0), 4);
        // (BN-backward constants: lane k < 32 holds channel k's, rows pick them up with v_readlane; xq: request_bx)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = 2 * i + half, cout = cout0 + wave * 32 + row;
            const bool cv = cout < a.Cout;
            const f32x4 v4 = *reinterpret_cast<const f32x4*>(tr + row * C1_TRS + 4 * (lane & 31));
            float v[4] = {v4[0], v4[1], v4[2], v4[3]};
            if (bnb) {
                // backward through mask -> ReLU -> BN-apply of the layer's prologue, with the BN-backward sums
                auto pick = [&](float pv) __attribute__((always_inline)) {
                    const float lo = __uint_as_float(__builtin_amdgcn_readlane((int)__float_as_uint(pv), 2 * i));
                    const float hi = __uint_as_float(__builtin_amdgcn_readlane((int)__float_as_uint(pv), 2 * i + 1));
                    return half ? hi : lo;
                };
                const float bsc = pick(p_sc), bsh = pick(p_sh), bmu = pick(p_mu), bis = pick(p_is);
                const float xv[4] = {__uint_as_float(xq[i].x), __uint_as_float(xq[i].y), __uint_as_float(xq[i].z), __uint_as_float(xq[i].w)};
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float z = fmaf(xv[k], bsc, bsh);
                    const bool keep = k < n_sl && (!a.relu || z > 0.f);
                    v[k] = keep ? v[k] : 0.f;
                    s1 += v[k]; s2 = fmaf(v[k], (xv[k] - bmu) * bis, s2);
                }
                if (a.stats) {
                    s1 = wave_sum16(s1); s2 = wave_sum16(s2);
                    s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);       // the row's other 16 lanes
                    if ((lane & 31) == 0 && cv) {
                        double* dst = a.stats + ((size_t)slot * a.Cout + cout) * 2;
                        atomicAdd(dst, (double)s1);
                        atomicAdd(dst + 1, (double)s2);
                    }
                }
            }
            const unsigned e = (unsigned)(cout * a.T + tq);
            if (vec) {
                const unsigned ok = (unsigned)-(int)(cv && n_T > 0);
                const u32x4_t q = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                __builtin_amdgcn_raw_buffer_store_b128(q, rs_y, ((e * 4u) & ok) | (OOB_C & ~ok), 0, 0);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned ok = (unsigned)-(int)(cv && k < n_T);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[k]), rs_y, (((e + (unsigned)k) * 4u) & ok) | (OOB_C & ~ok), 0, 0);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();                                  // the region is rewritten by the next tile's epilogue
    };

    // one chunk of the stream: KW steps of 96 MFMAs; G3 = (chunk index of the stream) % 3 fixes the ring slots at compile time
    auto chunk_step = [&](int g, auto g3_c) __attribute__((always_inline)) {
        constexpr int G3 = decltype(g3_c)::value;
        const unsigned char* img = smem_raw + (g & 1) * C1_IMG;
        if (ch == 0) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int n = 0; n < 8; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    biasv[m][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                        rs_b, (unsigned)(ct * C1_CT + wave * 32 + m * 16 + lq * 4 + r) * 4u, 0, 0));      // beyond Cout / no bias: 0
            }
        }
        if (DGRAD && bnb && ch == nChunks - 1) request_bx();
        u32x4_t Bf[2][2][3];                                          // [buffer][n of the pair][part]
        auto read_B = [&](int kw, int np, u32x4_t (&dst)[2][3]) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    dst[j][p] = *reinterpret_cast<const u32x4_t*>(img + p * C1_PART + (np * 2 + j) * 1024 + b_lane[kw]);
        };
        read_B(0, 0, Bf[0]);
#pragma unroll
        for (int kw = 0; kw < KW; ++kw) {
            const int s3 = (G3 * KW + kw) % 3;                            // ring slot of this step (compile time after unrolling)
            // weights of the step after next into the slot the previous step left
            if (s3 == 0) { load_A(I2{}, pf_soff); } else if (s3 == 1) { load_A(I0{}, pf_soff); } else { load_A(I1{}, pf_soff); }
            pf_advance();
            if (C1_DBG & 8) continue;
#pragma unroll
            for (int np = 0; np < 4; ++np) {
                const int cur = (kw * 4 + np) & 1;
                if (np + 1 < 4) read_B(kw, np + 1, Bf[cur ^ 1]);
                else if (kw + 1 < KW) read_B(kw + 1, 0, Bf[cur ^ 1]);
                // six part products, smallest first, round-robin over the pair's four accumulators
#pragma unroll
                for (int pp = 0; pp < 6; ++pp) {
                    const int pa = pp == 0 ? 2 : pp == 1 ? 0 : pp == 2 ? 1 : pp == 3 ? 1 : 0;     // lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
                    const int pb = pp == 0 ? 0 : pp == 1 ? 2 : pp == 2 ? 1 : pp == 3 ? 0 : pp == 4 ? 1 : 0;
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int m = 0; m < 2; ++m)
                            acc[m][np * 2 + j] = mfma_b16(A[s3][m][pa], Bf[cur][j][pb], acc[m][np * 2 + j]);
                }
            }
        }
        __syncthreads();
        if (++ch == nChunks) {                                            // the producers are already staging the next tile
            epilogue();
            ch = 0;
            ++k;
            if (k < nT) item(k, ct, sp);
        }
    };

    __syncthreads();                                                 // the first chunk is staged
    if (KW == 3) {
        for (int g = 0; g < G; ++g) chunk_step(g, I0{});
    } else {
        for (int g = 0; g < G; g += 3) {
            chunk_step(g, I0{});
            if (g + 1 < G) chunk_step(g + 1, I1{});
            if (g + 2 < G) chunk_step(g + 2, I2{});
        }
    }
}

template <int KW, bool DGRAD>
static int launch_c1(const ConvFwdArgs& a, hipStream_t s) {
    // the loaders address one clip with 32-bit byte offsets (buffer loads; 2^31 marks "out of range")
    if ((size_t)a.CinP * a.T * 4 >= (1ull << 31) || (size_t)a.CoutP * a.T * 4 >= (1ull << 31)) {
        set_error("conv1d_x3: one clip of the input / output must stay below 2 GiB (Cin=%d Cout=%d T=%d)", a.Cin, a.Cout, a.T);
        return PBSED_E_ARG;
    }
    if ((size_t)KW * a.CinP * a.CoutP * 6 >= (1ull << 31)) { set_error("conv1d_x3: packed weights exceed 2 GiB"); return PBSED_E_ARG; }
    const int nTt = (a.T + C1_TT - 1) / C1_TT;
    const int nSp = nTt * a.B, nCt = a.CoutP / C1_CT;
    const int nWork = (nSp + 7) / 8 * 8 * nCt;
    // persistent blocks, one per CU, a multiple of 8 so that an item's XCD is its block's XCD
    int blocks = device_cus() / 8 * 8;
    if (blocks < 8) blocks = 8;
    if (blocks > nWork) blocks = (nWork + 7) / 8 * 8;
    auto kern = conv1d_pc_kernel<KW, DGRAD>;
    PBSED_DYN_LDS_ONCE(kern, C1_LDS_ALL);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), C1_LDS_ALL, s, a, nCt, nSp, nWork);
    return check_launch("conv1d_x3");
}

}  // namespace pbsed

using namespace pbsed;

extern "C" {

void pbsed_conv1d_pack_dims_x3(int Cin, int Cout, int dgrad, int* InP, int* OutP) {
    const int in = dgrad ? Cout : Cin, out = dgrad ? Cin : Cout;
    *InP = (in + C1_CK - 1) / C1_CK * C1_CK;
    *OutP = (out + C1_CT - 1) / C1_CT * C1_CT;
}

// up: uint16 [InP/32][kw][OutP/16][part 3][lane 64][8]
int pbsed_pack_conv1d_weights_x3(const float* w, unsigned short* up, int Cout, int Cin, int KW, int dgrad, void* stream) {
    if (KW != 1 && KW != 3) { set_error("pack_conv1d_weights_x3: kernel size %d (1 or 3)", KW); return PBSED_E_UNSUPPORTED; }
    int InP, OutP;
    pbsed_conv1d_pack_dims_x3(Cin, Cout, dgrad, &InP, &OutP);
    const size_t total = (size_t)KW * InP * OutP;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(c1x3_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, up, Cout, Cin, KW, InP, OutP, dgrad);
    return check_launch("pack_conv1d_weights_x3");
}

int pbsed_conv1d_fwd_x3(const float* x, const unsigned short* u_packed, const float* bias, const float* scale, const float* shift,
                        int relu, const int* seq_len, float* y, double* stats, int B, int Cin, int Cout, int T, int KW,
                        void* stream) {
    ConvFwdArgs a{};
    a.x = x; a.wp = reinterpret_cast<const float*>(u_packed); a.bias = bias; a.scale = scale; a.shift = shift; a.seq_len = seq_len;
    a.y = y; a.stats = stats; a.relu = relu;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.F = 1; a.T = T;
    pbsed_conv1d_pack_dims_x3(Cin, Cout, 0, &a.CinP, &a.CoutP);
    if (KW == 3) return launch_c1<3, false>(a, (hipStream_t)stream);
    if (KW == 1) return launch_c1<1, false>(a, (hipStream_t)stream);
    set_error("conv1d_fwd_x3: kernel size %d (1 or 3)", KW);
    return PBSED_E_UNSUPPORTED;
}

int pbsed_conv1d_bwd_data_x3(const float* g, const unsigned short* ud_packed, const int* seq_len, float* dz, const float* bx,
                             const float* bmean, const float* binvstd, const float* bscale, const float* bshift, int relu,
                             double* stats, int B, int Cin, int Cout, int T, int KW, void* stream) {
    ConvFwdArgs a{};
    a.x = g; a.wp = reinterpret_cast<const float*>(ud_packed); a.seq_len = seq_len; a.y = dz;
    a.bx = bx; a.bmean = bmean; a.binvstd = binvstd; a.bscale = bscale; a.bshift = bshift;
    a.relu = relu; a.stats = bx ? stats : nullptr;
    a.B = B; a.Cin = Cout; a.Cout = Cin; a.F = 1; a.T = T;      // roles swapped
    pbsed_conv1d_pack_dims_x3(Cin, Cout, 1, &a.CinP, &a.CoutP);
    if (KW == 3) return launch_c1<3, true>(a, (hipStream_t)stream);
    if (KW == 1) return launch_c1<1, true>(a, (hipStream_t)stream);
    set_error("conv1d_bwd_data_x3: kernel size %d (1 or 3)", KW);
    return PBSED_E_UNSUPPORTED;
}

}  // extern "C"
