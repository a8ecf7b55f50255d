// 3x3 convolution forward / data gradient of the FEW-CHANNEL layers of the fp32 path (contraction over <= 16 channels into
// <= 32: the 16->16 (+ pool) and 16->32 layers of pb_sed/experiments/weak_label_crnn/training.py:161-168 and the tag-conditioned
// 11->16 first layer of the strong-label CNN) on the bf16 MFMA with EXACT three-way bf16 operand splits (fp32-class results),
// for gfx950.  These launches are HBM-bound by their tensors (16 channels x 128 rows x 500 frames x 32 clips = 131 MB each
// way); on the fp32 MFMA (conv.hip) they are bound by the pipe instead - 9.4 GFLOP at the fp32-MFMA rate, which VALU
// instructions share, is 130 - 190 us against 35 - 50 us of memory time.  Here:
//
//   y[cout, f, t] = sum_{kh, kw, cin} W[cout, cin, kh, kw] * x[cin, f + kh - 1, t + kw - 1]
//   M = cout (A = weights), N = 16 consecutive t of one row (B = input patch), K = 32 = TWO TAPS x 16 channels per MFMA:
//   a lane's 8 consecutive k are 8 channels of ONE tap, so the B fragment of a tap pair is one 16-byte LDS read per part at a
//   per-lane position offset (lanes 0..31: tap 2 s, lanes 32..63: tap 2 s + 1); the nine taps are five such steps (the tenth
//   tap has zero weights).  30 bf16 MFMAs per 16 t x 16 cout instead of 36 fp32 MFMAs at a sixteenth of the rate.
//
// Block = 256 threads, PERSISTENT over (clip, 4-row, 64-t) tiles: the pre-split weights (fragment order, 15 - 30 KB) are
// copied to LDS once per block; per tile every thread stages (4 channels, 4 positions) items - aligned 16-byte loads, lanes along t,
// prologue (BN-apply + ReLU + mask, or the un-pool of a pooled gradient), truncation split, one 8-byte LDS store per part into
// the image [part][row 6][position 66][16 channels] (32-byte positions, the two 16-byte halves swapped on every second group
// of 8 positions: fragment reads conflict-free) - and the loads of the tile after next are issued as soon as a tile is staged.
// Accumulators sit in conv_fwd_kernel's layout, so conv_epilogue.h (bias, pool + argmax byte, masked statistics, the
// BN-ReLU-backward form of the data gradient) is reused as it is.  53 - 68 KB of LDS: two to three blocks per CU.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "conv_epilogue.h"
#include "pack_elems.h"
#include "pbsed_internal.h"

namespace pbsed {

constexpr int S16_FT = 4, S16_TT = 64, S16_ROWS = S16_FT + 2, S16_POSW = S16_TT + 2;
constexpr int S16_STEPS = 5;                                  // tap pairs (0,1) (2,3) (4,5) (6,7) (8,-)
constexpr int S16_PART = S16_ROWS * S16_POSW * 32;            // bytes of one part of the image
constexpr int S16_QUADS = S16_TT / 4 + 2;                     // aligned 4-t quads of a row: t0 - 4 .. t0 + TT + 3
constexpr int S16_ITEMS = S16_ROWS * 4 * S16_QUADS;           // (row, channel group of 4, quad): 4 channels x 4 positions each
constexpr int S16_PER_T = (S16_ITEMS + 255) / 256;

__global__ void s16_pack_kernel(const float* __restrict__ w, unsigned short* __restrict__ up, int Cout, int Cin, int OutP, int dgrad) {
    const size_t total = (size_t)S16_STEPS * (OutP / 16) * 512;      // values; three halfwords each
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        pack_s16_value(w, up, i, Cout, Cin, OutP, dgrad);
}

template <int COUT_T, bool POOL, bool DGRAD>
struct S16Cfg {
    static constexpr int WM = COUT_T / 16, WN = 4 / WM;           // 16 cout: 1 x 4 waves, 32 cout: 2 x 2
    static constexpr int MTW = 1, NTT = (S16_TT / 16) / WN, NTW = S16_FT * NTT;
    static constexpr int MT = COUT_T / 16;
    static constexpr int W_BYTES = S16_STEPS * 3 * MT * 1024;
    static constexpr int FO_T = POOL ? S16_FT / 2 : S16_FT;
    static constexpr int ST_FLOATS = WN * COUT_T * FO_T * 2;
    static constexpr int LDS_BYTES = W_BYTES + 3 * S16_PART + 128 + ST_FLOATS * 4;
};

template <int COUT_T, bool POOL, bool DGRAD>
__global__ __launch_bounds__(256) void conv_s16_kernel(ConvFwdArgs a, int nTiles) {
    using C = S16Cfg<COUT_T, POOL, DGRAD>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char* w_s = smem_raw;                                  // [step][part][cout tile][lane] x 16 B
    unsigned char* img = smem_raw + C::W_BYTES;                     // [part][row][position][16 ch] bf16
    float* sc_s = reinterpret_cast<float*>(img + 3 * S16_PART);     // [16] scale, [16] shift
    float* st_s = sc_s + 32;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WN, wn = wave % C::WN;
    const int lq = lane >> 4, lr = lane & 15;
    const bool pro = a.scale != nullptr;
    const bool unpool = DGRAD && a.unpool_idx != nullptr;
    const int Fsrc = unpool ? a.F / 2 : a.F;
    const int nTt = (a.T + S16_TT - 1) / S16_TT, nFt = (a.F + S16_FT - 1) / S16_FT;
    constexpr unsigned OOB = 0x80000000u;
    const unsigned clip_elems = (unsigned)(a.Cin * Fsrc * a.T);
    const unsigned ch_step = (unsigned)(Fsrc * a.T) * 4u;          // bytes from a channel to the next

    // weights -> LDS once; scale / shift of the (<= 16) input channels too
    {
        const u32x4_t* src = reinterpret_cast<const u32x4_t*>(a.wp);
        for (int i = tid; i < C::W_BYTES / 16; i += 256) reinterpret_cast<u32x4_t*>(w_s)[i] = src[i];
        if (tid < 32) {
            const int c = tid & 15;
            float v = tid < 16 ? 1.f : 0.f;
            if (pro && c < a.Cin) v = tid < 16 ? a.scale[c] : a.shift[c];
            sc_s[tid] = v;
        }
    }
    // staging items of this thread: q = tid + 256 i -> (row r, channel group g, quad qd): 4 channels x the 4 positions
    // t0 - 4 + 4 qd .. + 3 (16-byte loads; image position p = t - (t0 - 1), kept when 0 <= p < POSW); fixed over the tiles
    int it_r[S16_PER_T], it_g[S16_PER_T], it_q[S16_PER_T];
#pragma unroll
    for (int i = 0; i < S16_PER_T; ++i) {
        const int q = tid + i * 256;
        const int rg = q / S16_QUADS;
        it_q[i] = q - rg * S16_QUADS;
        it_r[i] = rg >> 2; it_g[i] = rg & 3;
    }
    // B fragment offsets of the five steps: tap = 2 s + (lq >> 1), channels (lq & 1) * 8 ..; A fragment offset of this wave
    unsigned b_off[S16_STEPS];
#pragma unroll
    for (int s = 0; s < S16_STEPS; ++s) {
        const int tap = min(2 * s + (lq >> 1), 8), kh = tap / 3, kw = tap - 3 * kh;      // (tap 9: zero weights, any valid address)
        const int p = lr + kw;
        b_off[s] = (unsigned)((kh * S16_POSW + p) * 32 + (((lq & 1) ^ ((p >> 3) & 1)) * 16));
    }
    const unsigned a_off = (unsigned)(wm * 1024 + lane * 16);

    // two raw register sets: the loads of tile k + 2 are issued as soon as tile k is staged, i.e. they are in flight during the
    // MFMAs and the epilogue of tile k AND the whole of tile k + 1 (a tile is ~4 us per block, HBM latency under load 2 - 4 us;
    // with one set - loads issued behind the staging of the same iteration - the blocks waited for memory half of the time)
    // (32-cout blocks keep ONE set: with two they exceed 256 registers and fall to one block per CU - 96 -> 131 us at 16->32)
    constexpr int NSETS = COUT_T == 16 ? 2 : 1;
    u32x4_t rin[NSETS][S16_PER_T][4];
    unsigned ridx[NSETS][S16_PER_T][4];
    struct TileMeta { int b, f0, t0; unsigned ok; };
    TileMeta meta[NSETS];
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    auto load_tile = [&](auto set_c, int tile) __attribute__((always_inline)) {
        constexpr int SET = decltype(set_c)::value;
        TileMeta& m = meta[SET];
        m.t0 = (tile % nTt) * S16_TT;
        const int r2 = tile / nTt;
        m.f0 = (r2 % nFt) * S16_FT;
        m.b = r2 / nFt;
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.x) + (size_t)m.b * clip_elems, 0, clip_elems * 4u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc(
            unpool ? const_cast<uint8_t*>(a.unpool_idx) + (size_t)m.b * clip_elems : nullptr, 0, unpool ? clip_elems : 0u, 0x00020000);
        m.ok = 0;
#pragma unroll
        for (int i = 0; i < S16_PER_T; ++i) {
            const int f = m.f0 - 1 + it_r[i], tq = m.t0 - 4 + 4 * it_q[i];
            const bool ok = (tid + i * 256 < S16_ITEMS) && f >= 0 && f < a.F && tq >= 0 && tq < a.T;
            m.ok |= (unsigned)ok << i;
            const int c0 = it_g[i] * 4;
            const unsigned e0 = (unsigned)((c0 * Fsrc + (unpool ? (f >> 1) : f)) * a.T + tq);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool okc = ok && c0 + c < a.Cin;
                const unsigned off = okc ? (e0 * 4u + (unsigned)c * ch_step) : OOB;
                rin[SET][i][c] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0);
                if (unpool) ridx[SET][i][c] = __builtin_amdgcn_raw_buffer_load_b32(rs_i, okc ? (off >> 2) : OOB, 0, 0);
            }
        }
    };

    auto do_tile = [&](auto set_c, int tile) __attribute__((always_inline)) {
        constexpr int SET = decltype(set_c)::value;
        const int b = meta[SET].b, f0 = meta[SET].f0, t0 = meta[SET].t0;
        const unsigned ok_cur = meta[SET].ok;
        const int sl = a.seq_len ? min(a.seq_len[b], a.T) : a.T;
        const int tlim = pro ? sl : a.T;                             // Normalization re-masks its output (y * mask)
        __syncthreads();                                             // the previous tile's fragment reads (and st_s) are done
        // ---- stage: prologue / un-pool, split, 8-byte stores (4 channels of one position per store and part)
#pragma unroll
        for (int i = 0; i < S16_PER_T; ++i) {
            if (tid + i * 256 >= S16_ITEMS) continue;
            const bool inside = (ok_cur >> i) & 1u;
            const int par = (f0 - 1 + it_r[i]) & 1;
            const float4 sc = *reinterpret_cast<const float4*>(sc_s + it_g[i] * 4);
            const float4 sh = *reinterpret_cast<const float4*>(sc_s + 16 + it_g[i] * 4);
            const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
            const unsigned xv[4][4] = {{rin[SET][i][0].x, rin[SET][i][0].y, rin[SET][i][0].z, rin[SET][i][0].w},
                                       {rin[SET][i][1].x, rin[SET][i][1].y, rin[SET][i][1].z, rin[SET][i][1].w},
                                       {rin[SET][i][2].x, rin[SET][i][2].y, rin[SET][i][2].z, rin[SET][i][2].w},
                                       {rin[SET][i][3].x, rin[SET][i][3].y, rin[SET][i][3].z, rin[SET][i][3].w}};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int p = 4 * it_q[i] - 3 + e;                   // image position of t0 - 4 + 4 qd + e
                if (p < 0 || p >= S16_POSW) continue;
                const bool live = inside && (t0 - 1 + p) < tlim;
                float v[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float u = __uint_as_float(xv[c][e]);
                    if (unpool) u = (int)((ridx[SET][i][c] >> (8 * e)) & 0xffu) == par ? u : 0.f;
                    if (pro) {
                        u = fmaf(u, scv[c], shv[c]);
                        if (a.relu) u = fmaxf(u, 0.f);
                    }
                    v[c] = live ? u : 0.f;                            // zero padding is post-activation
                }
                unsigned h0, m0, l0, h1, m1, l1;
                split3_pair(v[0], v[1], h0, m0, l0);
                split3_pair(v[2], v[3], h1, m1, l1);
                unsigned char* dst = img + (unsigned)((it_r[i] * S16_POSW + p) * 32 + ((((it_g[i] >> 1) ^ (p >> 3)) & 1) * 16) + (it_g[i] & 1) * 8);
                *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
                *reinterpret_cast<uint2*>(dst + S16_PART) = make_uint2(m0, m1);
                *reinterpret_cast<uint2*>(dst + 2 * S16_PART) = make_uint2(l0, l1);
            }
        }
        __syncthreads();
        if (tile + NSETS * (int)gridDim.x < nTiles) load_tile(set_c, tile + NSETS * (int)gridDim.x);     // this set is free again
        // ---- MFMAs
        f32x4 acc[1][C::NTW];
#pragma unroll
        for (int n = 0; n < C::NTW; ++n) acc[0][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < S16_STEPS; ++s) {
            Bf3 A;
            A.hi = *reinterpret_cast<const u32x4_t*>(w_s + (s * 3 + 0) * C::MT * 1024 + a_off);
            A.mid = *reinterpret_cast<const u32x4_t*>(w_s + (s * 3 + 1) * C::MT * 1024 + a_off);
            A.lo = *reinterpret_cast<const u32x4_t*>(w_s + (s * 3 + 2) * C::MT * 1024 + a_off);
#pragma unroll
            for (int n = 0; n < C::NTW; ++n) {
                const int fl = n / C::NTT, tt = wn * C::NTT + n % C::NTT;
                const unsigned o = b_off[s] + (unsigned)((fl * S16_POSW + tt * 16) * 32);
                Bf3 Bv;
                Bv.hi = *reinterpret_cast<const u32x4_t*>(img + o);
                Bv.mid = *reinterpret_cast<const u32x4_t*>(img + S16_PART + o);
                Bv.lo = *reinterpret_cast<const u32x4_t*>(img + 2 * S16_PART + o);
                acc[0][n] = mfma_x3(A, Bv, acc[0][n]);
            }
        }
        conv_epilogue<COUT_T, S16_FT, S16_TT, 1, C::NTT, C::WN, POOL, DGRAD>(a, acc, st_s, b, f0, t0, 0, sl, wm, wn, lq, lr, tid);
    };

    const int first = blockIdx.x, stride = gridDim.x;
    if (first < nTiles) load_tile(I0{}, first);
    if constexpr (NSETS == 2) {
        if (first + stride < nTiles) load_tile(I1{}, first + stride);
        for (int tile = first; tile < nTiles; tile += 2 * stride) {
            do_tile(I0{}, tile);
            if (tile + stride < nTiles) do_tile(I1{}, tile + stride);
        }
    } else {
        for (int tile = first; tile < nTiles; tile += stride) do_tile(I0{}, tile);
    }
}

template <int COUT_T, bool POOL, bool DGRAD>
static int launch_s16(const ConvFwdArgs& a, hipStream_t s) {
    using C = S16Cfg<COUT_T, POOL, DGRAD>;
    if ((size_t)a.Cin * a.F * a.T * 4 >= (1ull << 30) || (size_t)a.Cout * a.F * a.T * 4 >= (1ull << 29)) {
        set_error("conv_s16: one clip of the input / output must stay below 1 GiB / 512 MiB (Cin=%d Cout=%d F=%d T=%d)", a.Cin, a.Cout, a.F, a.T);
        return PBSED_E_ARG;
    }
    if (a.T & 3) { set_error("conv_s16: T = %d is not a multiple of 4 (rows must be 16-byte aligned; use the direct kernel)", a.T); return PBSED_E_UNSUPPORTED; }
    const int nTt = (a.T + S16_TT - 1) / S16_TT, nFt = (a.F + S16_FT - 1) / S16_FT;
    const long long nTiles = (long long)nTt * nFt * a.B;
    if (nTiles >= (1ll << 30)) { set_error("conv_s16: too many tiles"); return PBSED_E_ARG; }
    // persistent blocks: as many as can be resident (LDS: two to three per CU), each walking tiles tile, tile + grid, ...
    static int per_cu = 0;
    auto kern = conv_s16_kernel<COUT_T, POOL, DGRAD>;
    PBSED_DYN_LDS_ONCE(kern, C::LDS_BYTES);
    if (per_cu == 0) {
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, C::LDS_BYTES) != hipSuccess || occ < 1) occ = 1;
        per_cu = occ > 3 ? 3 : occ;
    }
    long long blocks = (long long)device_cus() * per_cu;
    if (blocks > nTiles) blocks = nTiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), C::LDS_BYTES, s, a, (int)nTiles);
    return check_launch("conv_s16");
}

}  // namespace pbsed

using namespace pbsed;

extern "C" {

// Packed dims of the few-channel bf16x3 kernels: InP = 16 (the contraction channels), OutP = 16 or 32.
void pbsed_conv_pack_dims_s16(int Cin, int Cout, int dgrad, int* InP, int* OutP) {
    const int out = dgrad ? Cin : Cout;
    *InP = 16;
    *OutP = out <= 16 ? 16 : 32;
}

// up: uint16 [step 5][part 3][OutP/16][lane 64][8]: the A fragments of the five tap-pair steps, pre-split (hi + mid + lo == w)
int pbsed_pack_conv_weights_s16(const float* w, unsigned short* up, int Cout, int Cin, int dgrad, void* stream) {
    const int in = dgrad ? Cout : Cin, out = dgrad ? Cin : Cout;
    if (in > 16 || out > 32) { set_error("pack_conv_weights_s16: contraction over <= 16 channels into <= 32 (got %d -> %d)", in, out); return PBSED_E_ARG; }
    int InP, OutP;
    pbsed_conv_pack_dims_s16(Cin, Cout, dgrad, &InP, &OutP);
    const size_t total = (size_t)S16_STEPS * (OutP / 16) * 512;
    hipLaunchKernelGGL(s16_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, up, Cout, Cin, OutP, dgrad);
    return check_launch("pack_conv_weights_s16");
}

// Same contract as pbsed_conv_fwd (3x3, zero padding) for Cin <= 16, Cout <= 32; wfrag from pbsed_pack_conv_weights_s16(dgrad = 0).
int pbsed_conv_fwd_s16(const float* x, const unsigned short* wfrag, const float* bias, const float* scale, const float* shift,
                       int relu, const int* seq_len, float* y, unsigned char* pool_idx, double* stats, int stats_per_cf, int B,
                       int Cin, int Cout, int F, int T, int pool, void* stream) {
    if (Cin > 16 || Cout > 32) { set_error("conv_fwd_s16: Cin <= 16 and Cout <= 32 (got %d -> %d)", Cin, Cout); return PBSED_E_UNSUPPORTED; }
    if (pool && (F % 2)) { set_error("conv_fwd_s16: pool needs even F"); return PBSED_E_ARG; }
    ConvFwdArgs a{};
    a.x = x; a.wp = reinterpret_cast<const float*>(wfrag); a.bias = bias; a.scale = scale; a.shift = shift; a.seq_len = seq_len;
    a.y = y; a.pool_idx = pool_idx; a.stats = stats; a.stats_cf = stats_per_cf; a.relu = relu;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.F = F; a.T = T;
    pbsed_conv_pack_dims_s16(Cin, Cout, 0, &a.CinP, &a.CoutP);
    hipStream_t s = (hipStream_t)stream;
    if (a.CoutP == 16) return pool ? launch_s16<16, true, false>(a, s) : launch_s16<16, false, false>(a, s);
    return pool ? launch_s16<32, true, false>(a, s) : launch_s16<32, false, false>(a, s);
}

// Same contract as pbsed_conv_bwd_data (3x3) for layers with Cout <= 16 (the contraction) and Cin <= 32 (produced).
int pbsed_conv_bwd_data_s16(const float* g, const unsigned short* wfrag_d, const unsigned char* unpool_idx, const int* seq_len,
                            float* dz, const float* bx, const float* bmean, const float* binvstd, const float* bscale,
                            const float* bshift, int relu, double* stats, int B, int Cin, int Cout, int F, int T, void* stream) {
    if (Cout > 16 || Cin > 32) { set_error("conv_bwd_data_s16: Cout <= 16 and Cin <= 32 (got %d -> %d)", Cin, Cout); return PBSED_E_UNSUPPORTED; }
    if (unpool_idx && (F % 2)) { set_error("conv_bwd_data_s16: unpool needs even F"); return PBSED_E_ARG; }
    ConvFwdArgs a{};
    a.x = g; a.wp = reinterpret_cast<const float*>(wfrag_d); a.seq_len = seq_len; a.y = dz; a.unpool_idx = unpool_idx;
    a.bx = bx; a.bmean = bmean; a.binvstd = binvstd; a.bscale = bscale; a.bshift = bshift;
    a.relu = relu; a.stats = bx ? stats : nullptr;
    a.B = B; a.Cin = Cout; a.Cout = Cin; a.F = F; a.T = T;      // roles swapped
    pbsed_conv_pack_dims_s16(Cin, Cout, 1, &a.CinP, &a.CoutP);
    hipStream_t s = (hipStream_t)stream;
    if (a.CoutP == 16) return launch_s16<16, false, true>(a, s);
    return launch_s16<32, false, true>(a, s);
}

}  // extern "C"
