// 3x3 weight gradient of the few-channel layers (16 input channels, 16 or 32 output channels: layers 2 and 3 of the 'shallow'
// net; the kernel itself takes 16-channel tiles of any multiple, the dispatch at the end of this file says which shapes win)
// with fp32-class operands on the bf16 MFMA; included by conv_wgrad.hip.
//
//     dW[co][ci][kf][kt] = sum_{b,f,t} dY[b,co,f,t] * a[b,ci,f+kf-1,t+kt-1]          (a = prologue(x), dY un-pooled)
//
// These layers have 4 - 19 GFLOP over 130 - 260 MB of activations: HBM-bound (25 - 33 us at 8 TB/s) if the operands can be
// formed fast enough.  The fp32-MFMA kernels above stage tiles in LDS and fetch one fp32 operand word per MFMA (16->16: 188 us,
// 35 % fp32-pipe busy).  Here nothing goes through LDS: K of the MFMA = 32 consecutive t of one (clip, row), a lane's eight k
// are eight consecutive t of its channel - what two 16-byte loads of the t-contiguous tensors deliver - and a wave walks down
// a column of rows f keeping
//   * the three-way bf16 split of three dY rows (f-1, f, f+1: A operands, 12 registers per 16 output channels and row),
//   * the split of ONE activation row with a one-element halo either side (ten values per lane, five packed pairs per part):
//     the operands of the three kt taps are registers 0..3 (t-1), the pairs re-aligned by 16 bits (t) and registers 1..4 (t+1),
// so a row costs one split of each tensor element (~6 VALU ops) against 9 taps x 6 part products on the MFMA pipe.  The fp32
// accumulators of all nine taps (x output-channel tiles) stay in registers over all units of a wave; the block's waves add
// them up in LDS and the block adds into one of the WGRAD_SLOTS partial copies.
// Units: (clip, 32-t chunk, 16-row segment); the dY rows either side of a segment are read by both neighbours (1/8 more dY).
#pragma once

struct WgradS16 {
    static constexpr int FS = 16, TT = 32, NT = 256;
};

template <int MT> struct S16Rows { Bf3 v[MT]; };

template <int MT, bool UNPOOL>
__global__ __launch_bounds__(256, 2) void conv_wgrad_s16_kernel(ConvWgradArgs a) {
    constexpr int FS = WgradS16::FS;
    extern __shared__ __attribute__((aligned(16))) float smem[];          // [16 * MT cout][16 cin * 9]
    const int tid = threadIdx.x, lane = tid & 63, kg = lane >> 4, lr = lane & 15;
    // wave-uniform on purpose: the units (hence the clip's buffer resources) follow from it, and resources the compiler cannot
    // prove uniform turn every load into a waterfall loop (first version: 8 600 clocks per row instead of ~1 500)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ci = blockIdx.y * 16 + lr, cout0 = blockIdx.z * 16 * MT;
    const bool pro = a.scale != nullptr;
    const bool ci_ok = ci < a.Cin;                        // fewer than 16 input channels (the tag-conditioned 11 -> 16 layer): the
                                                          // missing lanes read zeros and their columns are not written
    const float sc = !ci_ok ? 0.f : pro ? a.scale[ci] : 1.f, sh = (pro && ci_ok) ? a.shift[ci] : 0.f;
    const float floor_v = (pro && a.relu) ? 0.f : -__builtin_inff();
    const bool do_bias = a.db != nullptr && blockIdx.y == 0;
    const int Fg = UNPOOL ? a.F / 2 : a.F;
    const int nTc = (a.T + 31) / 32, nFs = (a.F + FS - 1) / FS, nUnits = a.B * nTc * nFs;
    const unsigned gclip = (unsigned)(a.Cout * Fg * a.T), xclip = (unsigned)(a.Cin * a.F * a.T);

    f32x4 acc[9][MT];
    float accb[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        accb[m] = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[k][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    for (int unit = blockIdx.x * 4 + wave; unit < nUnits; unit += gridDim.x * 4) {
        const int tc = unit % nTc, fs = (unit / nTc) % nFs, b = unit / (nFs * nTc);
        const int t0 = tc * 32, f0 = fs * FS;
        const int sl = a.seq_len ? min(a.seq_len[b], a.T) : a.T;
        const int tlim = pro ? sl : a.T;                                   // activations are zero from the clip's length on
        const bool edge = t0 == 0 || t0 + 33 > tlim;                       // some t-1 .. t+1 of the chunk is outside [0, tlim)
        const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.g) + (size_t)b * gclip, 0, gclip * 4u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc(
            UNPOOL ? const_cast<uint8_t*>(a.unpool_idx) + (size_t)b * gclip : nullptr, 0, UNPOOL ? gclip : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.x) + (size_t)b * xclip, 0, xclip * 4u, 0x00020000);
        const int pbase = t0 + kg * 8;                                      // t of this lane's first dY element
        const unsigned xo = (unsigned)(ci * a.F * a.T + pbase);
        unsigned go[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) go[m] = (unsigned)((cout0 + m * 16 + lr) * Fg * a.T + pbase);
        constexpr unsigned OOB = 0x20000000u;

        // raw rows in flight, three rows ahead of their use
        struct RawX { u32x4_t q0, q1; unsigned l, r; };
        struct RawG { u32x4_t q0[MT], q1[MT]; unsigned i0[MT], i1[MT]; };
        RawX rx[3];
        RawG rg[3];
        auto load_x = [&](int f, RawX& o) __attribute__((always_inline)) {
            const unsigned off = (f >= 0 && f < a.F && ci_ok) ? xo + (unsigned)(f * a.T) : OOB;
            o.q0 = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off * 4u, 0, 0);
            o.q1 = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off * 4u + 16u, 0, 0);
            o.l = __builtin_amdgcn_raw_buffer_load_b32(rs_x, off * 4u - 4u, 0, 0);
            o.r = __builtin_amdgcn_raw_buffer_load_b32(rs_x, off * 4u + 32u, 0, 0);
        };
        auto load_g = [&](int f, RawG& o) __attribute__((always_inline)) {
            const bool ok = f >= 0 && f < a.F;
            const int fg = UNPOOL ? (f >> 1) : f;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const unsigned off = ok ? go[m] + (unsigned)(fg * a.T) : OOB;
                o.q0[m] = __builtin_amdgcn_raw_buffer_load_b128(rs_g, off * 4u, 0, 0);
                o.q1[m] = __builtin_amdgcn_raw_buffer_load_b128(rs_g, off * 4u + 16u, 0, 0);
                if (UNPOOL) {
                    o.i0[m] = __builtin_amdgcn_raw_buffer_load_b32(rs_i, off, 0, 0);
                    o.i1[m] = __builtin_amdgcn_raw_buffer_load_b32(rs_i, off + 4u, 0, 0);
                }
            }
        };
        // dY row in flight -> split A operands (un-pooled through the argmax byte; the bias gradient counts a row once:
        // in the segment that owns it)
        auto make_g = [&](int f, const RawG& in, S16Rows<MT>& out) __attribute__((always_inline)) {
            const bool own = do_bias && f >= f0 && f < f0 + FS;           // (rows past F read zeros)
            const unsigned par = (unsigned)(f & 1);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float v[8] = {__uint_as_float(in.q0[m].x), __uint_as_float(in.q0[m].y), __uint_as_float(in.q0[m].z),
                              __uint_as_float(in.q0[m].w), __uint_as_float(in.q1[m].x), __uint_as_float(in.q1[m].y),
                              __uint_as_float(in.q1[m].z), __uint_as_float(in.q1[m].w)};
                if (UNPOOL) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = ((in.i0[m] >> (8 * e)) & 0xffu) == par ? v[e] : 0.f;
                        v[4 + e] = ((in.i1[m] >> (8 * e)) & 0xffu) == par ? v[4 + e] : 0.f;
                    }
                }
                if (edge) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = pbase + e < a.T ? v[e] : 0.f;
                }
                if (own) accb[m] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                out.v[m] = split3x8(make_float4(v[0], v[1], v[2], v[3]), make_float4(v[4], v[5], v[6], v[7]));
            }
        };
        // activation row in flight -> prologue -> the B operands of the three kt taps
        Bf3 xb[3];
        auto make_x = [&](const RawX& in) __attribute__((always_inline)) {
            float v[10] = {__uint_as_float(in.l), __uint_as_float(in.q0.x), __uint_as_float(in.q0.y), __uint_as_float(in.q0.z),
                           __uint_as_float(in.q0.w), __uint_as_float(in.q1.x), __uint_as_float(in.q1.y), __uint_as_float(in.q1.z),
                           __uint_as_float(in.q1.w), __uint_as_float(in.r)};
#pragma unroll
            for (int i = 0; i < 10; ++i) v[i] = fmaxf(fmaf(v[i], sc, sh), floor_v);
            if (edge) {
#pragma unroll
                for (int i = 0; i < 10; ++i) v[i] = (unsigned)(pbase - 1 + i) < (unsigned)tlim ? v[i] : 0.f;
            }
            unsigned ph[5], pm[5], pl[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) split3_pair(v[2 * j], v[2 * j + 1], ph[j], pm[j], pl[j]);
            xb[0] = Bf3{u32x4_t{ph[0], ph[1], ph[2], ph[3]}, u32x4_t{pm[0], pm[1], pm[2], pm[3]}, u32x4_t{pl[0], pl[1], pl[2], pl[3]}};
            xb[2] = Bf3{u32x4_t{ph[1], ph[2], ph[3], ph[4]}, u32x4_t{pm[1], pm[2], pm[3], pm[4]}, u32x4_t{pl[1], pl[2], pl[3], pl[4]}};
#define PBSED_S16_MID(p) u32x4_t{__builtin_amdgcn_alignbit(p[1], p[0], 16), __builtin_amdgcn_alignbit(p[2], p[1], 16), \
                                 __builtin_amdgcn_alignbit(p[3], p[2], 16), __builtin_amdgcn_alignbit(p[4], p[3], 16)}
            xb[1] = Bf3{PBSED_S16_MID(ph), PBSED_S16_MID(pm), PBSED_S16_MID(pl)};
#undef PBSED_S16_MID
        };
        const int nrow = min(FS, a.F - f0);                                 // the last segment of a column may be short
        // one activation row f = f0 + i (raw stage sx) against dY rows f+1 (kf = 0; raw stage sg), f (kf = 1), f-1 (kf = 2);
        // the two raw stages are refilled with the rows three further down
        auto row = [&](int i, RawX& sx, RawG& sg, S16Rows<MT>& g_prev, S16Rows<MT>& g_cur, S16Rows<MT>& g_next)
                       __attribute__((always_inline)) {
            const int f = f0 + i;
            make_g(f + 1, sg, g_next);
            make_x(sx);
            if (i + 3 < nrow) load_x(f + 3, sx);
            if (i + 3 <= nrow) load_g(f + 4, sg);
#pragma unroll
            for (int kt = 0; kt < 3; ++kt)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    acc[0 + kt][m] = mfma_x3(g_next.v[m], xb[kt], acc[0 + kt][m]);
                    acc[3 + kt][m] = mfma_x3(g_cur.v[m], xb[kt], acc[3 + kt][m]);
                    acc[6 + kt][m] = mfma_x3(g_prev.v[m], xb[kt], acc[6 + kt][m]);
                }
        };

        S16Rows<MT> gw[3];
        load_g(f0 - 1, rg[0]);
        load_g(f0, rg[2]);
        load_x(f0, rx[0]);
        load_g(f0 + 1, rg[1]);
        load_x(f0 + 1, rx[1]);
        make_g(f0 - 1, rg[0], gw[0]);
        make_g(f0, rg[2], gw[1]);
        load_g(f0 + 2, rg[2]);
        load_x(f0 + 2, rx[2]);
        load_g(f0 + 3, rg[0]);
        for (int i = 0; i < nrow; i += 3) {                                  // raw stages: x row i -> i % 3, dY row i + 1 -> (i + 1) % 3
            row(i, rx[0], rg[1], gw[0], gw[1], gw[2]);
            if (i + 1 < nrow) row(i + 1, rx[1], rg[2], gw[1], gw[2], gw[0]);
            if (i + 2 < nrow) row(i + 2, rx[2], rg[0], gw[2], gw[0], gw[1]);
        }
    }

    // ---- the block's four partial sums -> LDS -> one slot of the gradient
    constexpr int OUT_ROW = 16 * 9;
    for (int i = tid; i < 16 * MT * OUT_ROW; i += 256) smem[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(smem + (m * 16 + kg * 4 + r) * OUT_ROW + lr * 9 + k, acc[k][m][r]);
    __syncthreads();
    const int slot = a.nslots > 1 ? (int)(blockIdx.x % a.nslots) : 0;
    float* dwp = a.dw + (size_t)slot * a.slot_w;
    for (int i = tid; i < 16 * MT * OUT_ROW; i += 256) {
        const int co = cout0 + i / OUT_ROW, col = i % OUT_ROW;
        if ((int)blockIdx.y * 16 + col / 9 < a.Cin) atomicAdd(dwp + ((size_t)co * a.Cin + blockIdx.y * 16) * 9 + col, smem[i]);
    }
    if (do_bias) {
        // a lane's partial bias sum covers its channel's eight t of every row: the four k groups of a wave, then the waves
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float v = accb[m];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (kg == 0) atomicAdd(&a.db[(size_t)slot * a.slot_b + cout0 + m * 16 + lr], v);
        }
    }
}

template <int MT, bool UNPOOL>
static int launch_wgrad_s16_t(const ConvWgradArgs& a_in, hipStream_t s) {
    ConvWgradArgs a = a_in;
    auto kern = conv_wgrad_s16_kernel<MT, UNPOOL>;
    const int nUnits = a.B * ((a.T + 31) / 32) * ((a.F + WgradS16::FS - 1) / WgradS16::FS);
    const int gy = (a.Cin + 15) / 16, gz = a.Cout / (16 * MT);
    // two blocks per CU over all (input tile, output tile) pairs; a multiple of 8 columns keeps the pairs of one unit range on
    // one XCD (block id = x + gx * (y + gy * z), XCD = id % 8): they read the same rows, from one L2
    int gx = (2 * launch_cus()) / (gy * gz) / 8 * 8;
    if (gx < 8) gx = 8;
    if (gx * 4 > nUnits) gx = (nUnits + 3) / 4;
    const size_t lds = (size_t)16 * MT * 16 * 9 * sizeof(float);
    const int nw = a.Cout * a.Cin * 9, nb = a.Cout;
    float* scratch = (nw + nb <= WGRAD_SLOT_MAX && gx >= 4 * WGRAD_SLOTS / (gy * gz)) ? wgrad_slot_scratch(s) : nullptr;
    if (scratch) {
        const int stride = (nw + nb + 63) / 64 * 64;
        a.dw = scratch; a.db = a_in.db ? scratch + nw : nullptr;
        a.nslots = WGRAD_SLOTS; a.slot_w = stride; a.slot_b = stride;
    }
    hipLaunchKernelGGL(kern, dim3(gx, gy, gz), dim3(256), lds, s, a);
    if (scratch)
        hipLaunchKernelGGL(wgrad_slot_reduce_kernel, dim3((nw + nb + 255) / 256), dim3(256), 0, s, scratch, a_in.dw, a_in.db, nw,
                           nb, a.slot_w);
    return check_launch("conv_wgrad_s16");
}

// 3x3, fp32 path, 2 .. 16 input channels (11: the tag-conditioned first layer of the BiCRNN), 16 or 32 output channels, T % 4 == 0
// (16-byte loads).  Measured at B = 32, T = 500
// (tools/gpu_conv_bench.py): 16->16 F = 128 under a pool 187 -> 116 us, 16->32 F = 64 162 -> 118 us; 32->32 F = 64 (four tile
// pairs) 185 us against 153 us of conv_wgrad_pc_kernel, which keeps it.  A row is ~260 VALU instructions (splits 110, un-pool
// 30, prologue 20, addresses) and 54 MFMAs, and the two do not overlap on a SIMD: ~1 900 clocks per row and wave.
static bool wgrad_s16_takes(const ConvWgradArgs& a, int KH, int KW) {
    return !a.bf16 && KH == 3 && KW == 3 && a.Cin >= 2 && a.Cin <= 16 && (a.Cout == 16 || a.Cout == 32) && (a.T & 3) == 0;
}

// one (16 cout, 16 cin) tile pair per wave: two output tiles per wave need 256 registers with one row in flight, and the
// second split of the activation row that the pairs cost is hidden behind the MFMAs
static int launch_wgrad_s16(const ConvWgradArgs& a, hipStream_t s) {
    return a.unpool_idx ? launch_wgrad_s16_t<1, true>(a, s) : launch_wgrad_s16_t<1, false>(a, s);
}
