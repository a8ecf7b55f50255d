// 3x3 convolution forward / data gradient with the multiplications of the time axis done in the Winograd
// F(4,3) domain, on fp32 MFMA (gfx950).  Same op sites and contract as conv.hip (pb_sed/models/weak_label/
// crnn.py:93; 3x3 layers of training.py:159-169), same tensors, same prologue / epilogue fusions; selected for
// the layers whose direct kernel is MFMA-bound (>= 32 input and >= 64 output channels of the contraction).
//
// A 3x3 conv is three 1-D convs along t (one per kernel row kh) summed over kh and cin.  For every block of 4
// consecutive outputs along t ("tile", 6 inputs d_0..d_5 = x[4i-1 .. 4i+4]):
//     V_xi = (B^T d)_xi, xi = 0..5        U_xi[kh] = (G w[kh][0..2])_xi        M_xi = sum_{kh,cin} U_xi[kh] V_xi(row f+kh-1)
//     y_0..y_3 = A^T M
// so the MFMAs contract K = (kh, cin) for 6 transform points instead of 9 taps per output: 18 products per 4
// outputs instead of 36.  GEMM view per point: M = Cout (A = U), N = 16 tiles = 64 consecutive t of one row
// (B = V), D fragment = (cout, tile).  A wave keeps all 6 points of its (cout, row, tile) fragments, so the output
// transform is register arithmetic and leaves 4 consecutive t per lane (float4 stores), with both rows of a
// (2,1) pool window in the same lane as in conv.hip.  The input transform is applied between the global load and
// the LDS store (the raw tile is never staged): one work item = (cin, row, 4 tiles) = 18 inputs -> 6 x float4.
// Transform matrices: Lavin & Gray 2016, F(4,3); fp32 rounding error of the 1-D transform is ~1e-6 relative.
#include <cstdlib>

#include "common.h"
#include "pack_elems.h"
#include "pbsed_internal.h"

namespace pbsed {

constexpr int WN_CT = 64;          // cout per block
constexpr int WN_CK = 8;           // cin per LDS stage
constexpr int WN_FT = 4;           // output rows per block
constexpr int WN_TT = 64;          // output columns per block = 16 tiles
constexpr int WN_ROWS = WN_FT + 2;
constexpr int WN_PLANE_V = 112;    // [rows 6][tiles 16] = 96 floats padded to == 16 (mod 32)
constexpr int WN_COUT_P = 80;      // 64 padded to == 16 (mod 32)
constexpr int WN_V_FLOATS = 6 * WN_CK * WN_PLANE_V;
constexpr int WN_U_FLOATS = 18 * WN_CK * WN_COUT_P;
constexpr int WN_U_VEC = 18 * WN_CK * WN_CT / 4;       // float4 per U stage
constexpr int WN_U_PER_T = (WN_U_VEC + 255) / 256;

template <bool POOL>
struct WinoCfg {
    static constexpr int FO_T = POOL ? WN_FT / 2 : WN_FT;
    static constexpr int LDS_FLOATS = WN_V_FLOATS + WN_U_FLOATS + WN_CT * FO_T * 2;
};

__global__ void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ up, int Cout, int Cin, int InP, int OutP,
                                 int dgrad) {
    const size_t total = (size_t)18 * InP * OutP;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        up[i] = pack_wino_elem(w, i, Cout, Cin, InP, OutP, dgrad);
}

template <bool POOL, bool DGRAD>
__global__ __launch_bounds__(256, 2) void conv_wino_kernel(ConvFwdArgs a) {
    using C = WinoCfg<POOL>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* v_s = smem;                               // [6][CK][PLANE_V]
    float* u_s = smem + WN_V_FLOATS;                 // [3*6][CK][COUT_P]
    float* st_s = u_s + WN_U_FLOATS;                 // [CT][FO_T][2]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;         // 2 x 2 waves: 32 cout x 2 output rows each
    const int lq = lane >> 4, lr = lane & 15;

    const int nTt = (a.T + WN_TT - 1) / WN_TT, nFt = (a.F + WN_FT - 1) / WN_FT;
    int bx = blockIdx.x;
    const int t0 = (bx % nTt) * WN_TT; bx /= nTt;
    const int f0 = (bx % nFt) * WN_FT;
    const int b = bx / nFt;
    const int cout0 = blockIdx.y * WN_CT;
    const int sl = a.seq_len ? min(a.seq_len[b], a.T) : a.T;
    const bool pro = a.scale != nullptr;
    const bool unpool = DGRAD && a.unpool_idx != nullptr;
    const int Fsrc = unpool ? a.F / 2 : a.F;
    const bool vec = (a.T & 3) == 0;
    const int tlim = pro ? sl : a.T;                 // Normalization re-masks its output (y*mask)

    f32x4 acc[6][2][2];
#pragma unroll
    for (int x = 0; x < 6; ++x)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int fl = 0; fl < 2; ++fl) acc[x][m][fl] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- input work item of this thread: (cin ic, halo row ir, tile quad iq) -> inputs t0 + 16 iq - 1 .. + 16
    const bool loader = tid < WN_CK * WN_ROWS * 4;
    const int iq = tid & 3, ir = (tid >> 2) % WN_ROWS, ic = tid / (4 * WN_ROWS);
    const int fin = f0 - 1 + ir, tq0 = t0 + 16 * iq;
    const bool row_ok = loader && fin >= 0 && fin < a.F;
    const int par = fin & 1;
    // valid elements of the item: e in [e_lo, e_hi)  (t = tq0 - 1 + e in [0, tlim), zero padding is post-activation)
    const int e_hi = row_ok ? max(tlim - tq0 + 1, 0) : 0;
    const bool e0_ok = tq0 > 0 && e_hi > 0;

    // All loads are raw buffer loads on clip-relative resources: addresses are one 32-bit VGPR per stream, advanced by
    // a uniform step per chunk (fp32 MFMAs and VALU instructions share the SIMD's fp32 pipe on gfx950 - they do not
    // overlap - so every address instruction here is paid in MFMA time); out-of-range lanes (halo rows outside the
    // plane, padded channels, the element before the first row) read 0 without a branch.
    constexpr unsigned OOB = 0x80000000u;
    const unsigned clip_elems = (unsigned)(a.Cin * Fsrc * a.T);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x) + (size_t)b * clip_elems, 0, clip_elems * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc(
        unpool ? const_cast<uint8_t*>(a.unpool_idx) + (size_t)b * clip_elems : nullptr, 0, unpool ? clip_elems : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_u = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.wp), 0, (unsigned)(18 * a.CinP * a.CoutP) * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_sc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.scale), 0, pro ? (unsigned)a.Cin * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_sh = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.shift), 0, pro ? (unsigned)a.Cin * 4u : 0u, 0x00020000);
    const unsigned elem0 = (unsigned)((ic * Fsrc + (unpool ? (fin >> 1) : fin)) * a.T + tq0);
    unsigned voff_x = row_ok ? elem0 * 4u : OOB;     // byte offset of element e = 1 in the chunk to load next
    unsigned voff_i = row_ok ? elem0 : OOB;
    unsigned voff_c = row_ok ? (unsigned)ic * 4u : OOB;
    unsigned voff_u = (unsigned)(((tid >> 7) * a.CinP + ((tid >> 4) & 7)) * a.CoutP + cout0 + (tid & 15) * 4) * 4u;
    const unsigned step_x = (unsigned)(WN_CK * Fsrc * a.T) * 4u, step_u = (unsigned)(WN_CK * a.CoutP) * 4u;
    const unsigned stride_u = (unsigned)(2 * a.CinP * a.CoutP) * 4u;     // U load i: kx = tid/128 + 2 i

    unsigned rin[18];                                // raw inputs t = tq0 - 1 .. tq0 + 16 of the chunk in flight (bits)
    unsigned ridx[6];                                // DGRAD + unpool: pool-row bytes of the same elements
    u32x4_t ru[WN_U_PER_T];
    unsigned rsc = 0u, rsh = 0u;                     // BN-apply factors of this item's channel, fetched with the chunk

    // global loads only: everything that depends on the loaded values happens in store_chunk, after the MFMAs
    auto load_chunk = [&]() __attribute__((always_inline)) {
        if (pro) {
            rsc = __builtin_amdgcn_raw_buffer_load_b32(rs_sc, voff_c, 0, 0);
            rsh = __builtin_amdgcn_raw_buffer_load_b32(rs_sh, voff_c, 0, 0);
        }
        if (vec) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32x4_t xv = __builtin_amdgcn_raw_buffer_load_b128(rs_x, voff_x + q * 16, 0, 0);
                rin[1 + 4 * q] = xv.x; rin[2 + 4 * q] = xv.y; rin[3 + 4 * q] = xv.z; rin[4 + 4 * q] = xv.w;
            }
            rin[0] = __builtin_amdgcn_raw_buffer_load_b32(rs_x, voff_x - 4u, 0, 0);
            rin[17] = __builtin_amdgcn_raw_buffer_load_b32(rs_x, voff_x + 64u, 0, 0);
            if (unpool) {
#pragma unroll
                for (int q = 0; q < 4; ++q) ridx[q] = __builtin_amdgcn_raw_buffer_load_b32(rs_i, voff_i + q * 4, 0, 0);
                ridx[4] = __builtin_amdgcn_raw_buffer_load_b8(rs_i, voff_i - 1u, 0, 0);
                ridx[5] = __builtin_amdgcn_raw_buffer_load_b8(rs_i, voff_i + 16u, 0, 0);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 18; ++e) rin[e] = __builtin_amdgcn_raw_buffer_load_b32(rs_x, voff_x + (unsigned)(e - 1) * 4u, 0, 0);
            if (unpool) {
                // bytes gathered into the layout of the vector path: ridx[q] = elements 1 + 4q .. 4 + 4q
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned w = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        w |= (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rs_i, voff_i + (unsigned)(4 * q + k), 0, 0) << (8 * k);
                    ridx[q] = w;
                }
                ridx[4] = __builtin_amdgcn_raw_buffer_load_b8(rs_i, voff_i - 1u, 0, 0);
                ridx[5] = __builtin_amdgcn_raw_buffer_load_b8(rs_i, voff_i + 16u, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < WN_U_PER_T; ++i) ru[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_u, voff_u, i * stride_u, 0);
        voff_x += step_x; voff_i += step_x >> 2; voff_c += WN_CK * 4u; voff_u += step_u;
    };
    auto store_chunk = [&]() __attribute__((always_inline)) {
        if (loader) {
            float d[18];
            const float sc = __uint_as_float(rsc), sh = __uint_as_float(rsh);
#pragma unroll
            for (int e = 0; e < 18; ++e) {
                float u = __uint_as_float(rin[e]);
                if (unpool) {
                    const unsigned byte = e == 0 ? ridx[4] : e == 17 ? ridx[5] : (ridx[(e - 1) >> 2] >> (8 * ((e - 1) & 3))) & 0xffu;
                    u = (int)byte != par ? 0.f : u;
                }
                if (pro) {
                    u = fmaf(u, sc, sh);
                    if (a.relu) u = fmaxf(u, 0.f);
                }
                d[e] = (e == 0 ? e0_ok : e < e_hi) ? u : 0.f;
            }
            // B^T d for the 4 tiles of this item, one float4 (4 consecutive tiles) per transform point
            float vx[6][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float d0 = d[4 * i], d1 = d[4 * i + 1], d2 = d[4 * i + 2], d3 = d[4 * i + 3], d4 = d[4 * i + 4],
                            d5 = d[4 * i + 5];
                vx[0][i] = 4.f * d0 - 5.f * d2 + d4;
                vx[1][i] = -4.f * (d1 + d2) + d3 + d4;
                vx[2][i] = 4.f * (d1 - d2) - d3 + d4;
                vx[3][i] = -2.f * d1 - d2 + 2.f * d3 + d4;
                vx[4][i] = 2.f * d1 - d2 - 2.f * d3 + d4;
                vx[5][i] = 4.f * d1 - 5.f * d3 + d5;
            }
#pragma unroll
            for (int x = 0; x < 6; ++x)
                *reinterpret_cast<float4*>(v_s + (x * WN_CK + ic) * WN_PLANE_V + ir * 16 + iq * 4) =
                    make_float4(vx[x][0], vx[x][1], vx[x][2], vx[x][3]);
        }
        static_assert(WN_U_VEC % 256 == 0, "U stage = whole float4 rounds of the block");
#pragma unroll
        for (int i = 0; i < WN_U_PER_T; ++i)            // float4 idx = tid + 256 i: kc = tid/16 + 16 i, q = tid % 16
            *reinterpret_cast<u32x4_t*>(u_s + ((tid >> 4) + 16 * i) * WN_COUT_P + (tid & 15) * 4) = ru[i];
    };

    const int nChunks = a.CinP / WN_CK;
    load_chunk();
    for (int ch = 0; ch < nChunks; ++ch) {
        __syncthreads();            // previous chunk's MFMA reads are done
        __builtin_amdgcn_s_setprio(3);
        store_chunk();
        __syncthreads();
        if (ch + 1 < nChunks) load_chunk();                   // in flight during the MFMAs below
        __builtin_amdgcn_s_setprio(0);
        // operand fragments of group g + 1 are read while the MFMAs of group g issue
        constexpr int NG = 6 * (WN_CK / 4);
        float bf[2][4], af[2][3][2];
        auto read_group = [&](int g, float (&b)[4], float (&am)[3][2]) __attribute__((always_inline)) {
            const int x = g / (WN_CK / 4), cs = g % (WN_CK / 4);
            const float* vp = v_s + (x * WN_CK + cs * 4 + lq) * WN_PLANE_V + (wn * 2) * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) b[r] = vp[r * 16];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    am[kh][m] = u_s[((kh * 6 + x) * WN_CK + cs * 4 + lq) * WN_COUT_P + (wm * 2 + m) * 16 + lr];
        };
        read_group(0, bf[0], af[0]);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int x = g / (WN_CK / 4), cur = g & 1;
            if (g + 1 < NG) read_group(g + 1, bf[cur ^ 1], af[cur ^ 1]);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int fl = 0; fl < 2; ++fl)
                        acc[x][m][fl] = mfma16(af[cur][kh][m], bf[cur][fl + kh], acc[x][m][fl]);
            // pin the interleave: 5 x {1 ds_read2, 2 MFMA} + 2 MFMA (the scheduler otherwise sinks every read to its use)
            if (g + 1 < NG) {
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
            }
        }
    }

    // ---- epilogue: A^T M in registers, then bias / pool / statistics / BN-ReLU backward exactly as conv_epilogue,
    // on 4 consecutive t per lane:  t = t0 + 4*lr + e,  cout = cout0 + (wm*2+m)*16 + lq*4 + r,  f = f0 + wn*2 + fl.
    // Stores and the DGRAD re-load of the layer input are raw buffer accesses on clip-relative resources with one
    // 32-bit byte offset = channel + row + column part; a part is 2^31 / 2^29 when its index is out of range, so
    // invalid fragments drop out without branches (conv_epilogue.h).
    constexpr unsigned OOB_C = 0x80000000u, OOB_T = 0x20000000u;
    const int Fo = POOL ? a.F / 2 : a.F;
    const int tb = t0 + 4 * lr;
    const unsigned oclip = (unsigned)(a.Cout * Fo * a.T);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(a.y + (size_t)b * oclip, 0, oclip * 4u, 0x00020000);
    const bool want_idx = POOL && a.pool_idx != nullptr;
    const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(
        want_idx ? a.pool_idx + (size_t)b * oclip : nullptr, 0, want_idx ? oclip : 0u, 0x00020000);
    const bool bnb = DGRAD && a.bx != nullptr;
    const __amdgpu_buffer_rsrc_t rs_bx = __builtin_amdgcn_make_buffer_rsrc(
        bnb ? const_cast<float*>(a.bx) + (size_t)b * oclip : nullptr, 0, bnb ? oclip * 4u : 0u, 0x00020000);
    const unsigned tcol = tb < a.T ? (unsigned)tb * 4u : OOB_T;
    const int n_seq = min(max(sl - tb, 0), 4);                    // elements of the quad inside the sequence
    constexpr int NFO = POOL ? 1 : 2;
    unsigned roff[NFO];
    bool orow_ok[NFO];
#pragma unroll
    for (int fo_l = 0; fo_l < NFO; ++fo_l) {
        const int fo = POOL ? f0 / 2 + wn : f0 + wn * 2 + fo_l;
        orow_ok[fo_l] = fo < Fo;
        roff[fo_l] = orow_ok[fo_l] ? (unsigned)(fo * a.T) * 4u : OOB_T;
    }
    // DGRAD: the layer's raw forward input for every fragment this thread finishes, requested before any arithmetic
    // (inside the loops each load would wait behind the stores of the previous fragment)
    u32x4_t xq[2][4][2];
    if (bnb) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cout = cout0 + (wm * 2 + m) * 16 + lq * 4 + r;
                const unsigned coff = cout < a.Cout ? (unsigned)(cout * Fo * a.T) * 4u : OOB_C;
#pragma unroll
                for (int fl = 0; fl < NFO; ++fl) xq[m][r][fl] = __builtin_amdgcn_raw_buffer_load_b128(rs_bx, coff + roff[fl] + tcol, 0, 0);
            }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cl = (wm * 2 + m) * 16 + lq * 4 + r;
            const int cout = cout0 + cl;
            const bool cv = cout < a.Cout;
            const float bias = (a.bias && cv) ? a.bias[cout] : 0.f;
            const unsigned coff = cv ? (unsigned)(cout * Fo * a.T) * 4u : OOB_C;
            float bsc = 0.f, bsh = 0.f, bmu = 0.f, bis = 0.f;
            if (bnb && cv) { bsc = a.bscale[cout]; bsh = a.bshift[cout]; bmu = a.bmean[cout]; bis = a.binvstd[cout]; }
            float y[2][4];
#pragma unroll
            for (int fl = 0; fl < 2; ++fl) {
                const float m0 = acc[0][m][fl][r], m1 = acc[1][m][fl][r], m2 = acc[2][m][fl][r], m3 = acc[3][m][fl][r],
                            m4 = acc[4][m][fl][r], m5 = acc[5][m][fl][r];
                const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
                y[fl][0] = m0 + s12 + s34 + bias;
                y[fl][1] = d12 + 2.f * d34 + bias;
                y[fl][2] = s12 + 4.f * s34 + bias;
                y[fl][3] = d12 + 8.f * d34 + m5 + bias;
            }
            float c1 = 0.f, c2 = 0.f;               // per-channel statistics: both rows summed before the lane reduction
#pragma unroll
            for (int fo_l = 0; fo_l < NFO; ++fo_l) {
                float v[4];
                unsigned pbytes = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (POOL) {
                        const bool second = y[1][e] > y[0][e];
                        v[e] = second ? y[1][e] : y[0][e];
                        pbytes |= (unsigned)second << (8 * e);
                    } else {
                        v[e] = y[fo_l][e];
                    }
                }
                const int st_row = POOL ? wn : wn * 2 + fo_l;
                const int n_cnt = orow_ok[fo_l] ? n_seq : 0;           // padded channels produce exact zeros
                float s1 = 0.f, s2 = 0.f;
                if (DGRAD) {
                    if (bnb) {
                        // backward through mask -> ReLU -> BN-apply of the layer's prologue, with the BN-backward sums
                        const u32x4_t x4 = xq[m][r][fo_l];
                        const float xv[4] = {__uint_as_float(x4.x), __uint_as_float(x4.y), __uint_as_float(x4.z), __uint_as_float(x4.w)};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float z = fmaf(xv[e], bsc, bsh);
                            const bool keep = e < n_cnt && (!a.relu || z > 0.f);
                            v[e] = keep ? v[e] : 0.f;
                            s1 += v[e]; s2 = fmaf(v[e], (xv[e] - bmu) * bis, s2);
                        }
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float vm = e < n_cnt ? v[e] : 0.f;
                        s1 += vm; s2 = fmaf(vm, vm, s2);
                    }
                }
                const unsigned off = coff + roff[fo_l] + tcol;
                if (vec) {
                    const u32x4_t q = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                    __builtin_amdgcn_raw_buffer_store_b128(q, rs_y, off, 0, 0);
                    if (want_idx) __builtin_amdgcn_raw_buffer_store_b32(pbytes, rs_p, off >> 2, 0, 0);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned oe = tb + e < a.T ? off + 4u * e : OOB_C;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[e]), rs_y, oe, 0, 0);
                        if (want_idx) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(pbytes >> (8 * e)), rs_p, oe >> 2, 0, 0);
                    }
                }
                // every (channel, row) entry of st_s has exactly one writer: plain stores, fixed summation order
                if (a.stats && a.stats_cf) {
                    s1 = wave_sum16(s1);
                    s2 = wave_sum16(s2);
                    if (lr == 0) {
                        st_s[(cl * C::FO_T + st_row) * 2 + 0] = s1;
                        st_s[(cl * C::FO_T + st_row) * 2 + 1] = s2;
                    }
                }
                c1 += s1; c2 += s2;
            }
            if (a.stats && !a.stats_cf) {
                c1 = wave_sum16(c1);
                c2 = wave_sum16(c2);
                if (lr == 0) {
#pragma unroll
                    for (int fo_l = 0; fo_l < NFO; ++fo_l) {
                        const int st_row = POOL ? wn : wn * 2 + fo_l;
                        st_s[(cl * C::FO_T + st_row) * 2 + 0] = fo_l == 0 ? c1 : 0.f;
                        st_s[(cl * C::FO_T + st_row) * 2 + 1] = fo_l == 0 ? c2 : 0.f;
                    }
                }
            }
        }
    }
    if (a.stats) {
        __syncthreads();
        const int slot = blockIdx.x & (PBSED_STAT_SLOTS - 1);
        if (a.stats_cf) {
            for (int i = tid; i < WN_CT * C::FO_T * 2; i += 256) {
                const int which = i & 1, fo_l = (i >> 1) % C::FO_T, cl = (i >> 1) / C::FO_T;
                const int cout = cout0 + cl, fo = (POOL ? f0 / 2 : f0) + fo_l;
                if (cout < a.Cout && fo < Fo)
                    atomicAdd(&a.stats[((size_t)slot * a.Cout * Fo + cout * Fo + fo) * 2 + which], (double)st_s[i]);
            }
        } else {
            for (int i = tid; i < WN_CT * 2; i += 256) {
                const int which = i & 1, cl = i >> 1, cout = cout0 + cl;
                float v = 0.f;
#pragma unroll
                for (int fo_l = 0; fo_l < C::FO_T; ++fo_l) v += st_s[(cl * C::FO_T + fo_l) * 2 + which];
                if (cout < a.Cout) atomicAdd(&a.stats[((size_t)slot * a.Cout + cout) * 2 + which], (double)v);
            }
        }
    }
}

template <bool POOL, bool DGRAD>
static int launch_wino(const ConvFwdArgs& a, hipStream_t s) {
    using C = WinoCfg<POOL>;
    // the loaders address one clip with 32-bit byte offsets (buffer loads; 2^31 marks "out of range")
    if ((size_t)a.Cin * a.F * a.T * 4 >= (1ull << 30) || (size_t)a.Cout * a.F * a.T * 4 >= (1ull << 29)) {
        set_error("conv_wino: one clip of the input / output must stay below 1 GiB / 512 MiB (Cin=%d Cout=%d F=%d T=%d)", a.Cin,
                  a.Cout, a.F, a.T);
        return PBSED_E_ARG;
    }
    const int nTt = (a.T + WN_TT - 1) / WN_TT, nFt = (a.F + WN_FT - 1) / WN_FT;
    dim3 grid(nTt * nFt * a.B, a.CoutP / WN_CT);
    const size_t lds = C::LDS_FLOATS * sizeof(float);
    auto kern = conv_wino_kernel<POOL, DGRAD>;
    PBSED_DYN_LDS_ONCE(kern, lds);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    return check_launch("conv_wino");
}

}  // namespace pbsed

using namespace pbsed;

extern "C" {

void pbsed_conv_pack_dims_wino(int Cin, int Cout, int dgrad, int* InP, int* OutP) {
    const int in = dgrad ? Cout : Cin, out = dgrad ? Cin : Cout;
    *InP = (in + WN_CK - 1) / WN_CK * WN_CK;
    *OutP = (out + WN_CT - 1) / WN_CT * WN_CT;
}

int pbsed_pack_conv_weights_wino(const float* w, float* up, int Cout, int Cin, int dgrad, void* stream) {
    int InP, OutP;
    pbsed_conv_pack_dims_wino(Cin, Cout, dgrad, &InP, &OutP);
    const size_t total = (size_t)18 * InP * OutP;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(wino_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, up, Cout, Cin, InP, OutP, dgrad);
    return check_launch("pack_conv_weights_wino");
}

int pbsed_conv_fwd_wino(const float* x, const float* u_packed, const float* bias, const float* scale, const float* shift,
                        int relu, const int* seq_len, float* y, unsigned char* pool_idx, double* stats, int stats_per_cf,
                        int B, int Cin, int Cout, int F, int T, int pool, void* stream) {
    if (pool && (F % 2)) { set_error("conv_fwd_wino: pool needs even F"); return PBSED_E_ARG; }
    ConvFwdArgs a{};
    a.x = x; a.wp = u_packed; a.bias = bias; a.scale = scale; a.shift = shift; a.seq_len = seq_len;
    a.y = y; a.pool_idx = pool_idx; a.stats = stats; a.stats_cf = stats_per_cf; a.relu = relu;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.F = F; a.T = T;
    pbsed_conv_pack_dims_wino(Cin, Cout, 0, &a.CinP, &a.CoutP);
    return pool ? launch_wino<true, false>(a, (hipStream_t)stream) : launch_wino<false, false>(a, (hipStream_t)stream);
}

int pbsed_conv_bwd_data_wino(const float* g, const float* ud_packed, const unsigned char* unpool_idx, const int* seq_len,
                             float* dz, const float* bx, const float* bmean, const float* binvstd, const float* bscale,
                             const float* bshift, int relu, double* stats, int B, int Cin, int Cout, int F, int T,
                             void* stream) {
    if (unpool_idx && (F % 2)) { set_error("conv_bwd_data_wino: unpool needs even F"); return PBSED_E_ARG; }
    ConvFwdArgs a{};
    a.x = g; a.wp = ud_packed; a.seq_len = seq_len; a.y = dz; a.unpool_idx = unpool_idx;
    a.bx = bx; a.bmean = bmean; a.binvstd = binvstd; a.bscale = bscale; a.bshift = bshift;
    a.relu = relu; a.stats = bx ? stats : nullptr;
    a.B = B; a.Cin = Cout; a.Cout = Cin; a.F = F; a.T = T;      // roles swapped
    pbsed_conv_pack_dims_wino(Cin, Cout, 1, &a.CinP, &a.CoutP);
    return launch_wino<false, true>(a, (hipStream_t)stream);
}

}  // extern "C"
