// Convolution weight gradient on fp32 MFMA for gfx950:  dW[co][ci][kh][kw] += sum_{b,f,t}
// dY[b,co,f,t] * a[b,ci,f+kh-P,t+kw-P] with a = prologue(x) (BN-apply + ReLU + seq mask recomputed
// while staging, never materialised) and dY the un-pooled output gradient (gathered through the
// forward's pool argmax).  This is the backward twin of the op sites listed in conv.hip
// (torch autograd of Conv2d/Conv1d in the reference: pb_sed/models/weak_label/crnn.py:93).
//
// GEMM view: M = Cout (A = dY), N = (cin, tap) (B = shifted a), K = spatial (b,f,t) split over
// blocks.  Both operands are K-major in HBM (t contiguous), so tiles are staged with aligned 16-byte
// loads into LDS planes padded to == 2 (mod 32) dwords, which makes the transposed ds_read_b32
// operand fetches conflict-free.  The next spatial chunk is prefetched into registers during the
// MFMAs.  Per-block partial sums are transposed through LDS and reduced with row-contiguous fp32
// atomics into the [Cout,Cin,KH,KW] gradient (the bias gradient is one extra MFMA against ones).
#include <cstdlib>

#include <type_traits>

#include "common.h"
#include "pbsed_internal.h"

namespace pbsed {

constexpr int pad2mod32(int n) { return ((n + 29) / 32) * 32 + 2; }
constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <int KWAVES>
__device__ __forceinline__ void lds_acc(float* p, float v) {
    if (KWAVES > 1) atomicAdd(p, v); else *p = v;
}

template <int KH, int KW, int WAVES, int MT, int NCG, bool TAPN, int KWAVES, int FT_>
struct WgradCfg {
    // WAVES waves tile Cout (MT M-tiles each); KWAVES groups of them split the spatial (K) dimension;
    // FT_ > 0 overrides the number of frequency rows per chunk (more bytes in flight for few-channel layers)
    static constexpr int FT = FT_ ? FT_ : ((KH == 3) ? 2 : 1), TT = 64;
    static constexpr int KK = KH * KW, NT = WAVES * KWAVES * 64;
    static constexpr int TQ_PER = (TT / 4) / KWAVES;
    static constexpr int COUT_T = WAVES * MT * 16, CIN_T = TAPN ? 1 : NCG * 16;   // TAPN: Cin == 1, taps on N
    static constexpr int HALO = (KW > 1) ? 4 : 0;
    static constexpr int ROWS = FT + KH - 1, ROW = TT + 2 * HALO, QR = ROW / 4;
    static constexpr int PLANE_Y = pad2mod32(FT * TT), PLANE_A = pad2mod32(ROWS * ROW);
    static constexpr int YQ = COUT_T * FT * (TT / 4), AQ = CIN_T * ROWS * QR;
    static constexpr int Y_PER_T = (YQ + NT - 1) / NT, A_PER_T = (AQ + NT - 1) / NT;
    static constexpr int OUT_ROW = CIN_T * KK;
    static constexpr int NACC = TAPN ? 1 : NCG * KK;
    static constexpr int LDS_FLOATS = cmax(COUT_T * PLANE_Y + CIN_T * PLANE_A, COUT_T * OUT_ROW);
};

template <int KH, int KW, int WAVES, int MT, int NCG, bool TAPN, int KWAVES, int FT_>
__global__ __launch_bounds__(WAVES * KWAVES * 64, (KH == 3 && WAVES == 4 && NCG == 1 && !TAPN) ? 3 : 1)
void conv_wgrad_kernel(ConvWgradArgs a) {
    using C = WgradCfg<KH, KW, WAVES, MT, NCG, TAPN, KWAVES, FT_>;
    constexpr int FT = C::FT, TT = C::TT, KK = C::KK, NT = C::NT;
    constexpr int PADH = (KH - 1) / 2, PADW = (KW - 1) / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dy_s = smem;                              // [COUT_T][PLANE_Y]
    float* a_s = smem + C::COUT_T * C::PLANE_Y;      // [CIN_T][PLANE_A]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = (tid >> 6) % WAVES, kwv = (tid >> 6) / WAVES;     // Cout tile / spatial slice of this wave
    const int lq = lane >> 4, lr = lane & 15;
    const int cin0 = blockIdx.y * C::CIN_T, cout0 = blockIdx.z * C::COUT_T;
    const int nTt = (a.T + TT - 1) / TT, nFt = (a.F + FT - 1) / FT;
    const int nChunks = a.B * nFt * nTt;
    const bool pro = a.scale != nullptr;
    const bool do_bias = (a.db != nullptr) && (blockIdx.y == 0);
    const bool unpool = a.unpool_idx != nullptr;
    const int Fg = unpool ? a.F / 2 : a.F;
    const bool vec = (a.T & 3) == 0;

    f32x4 acc[MT][C::NACC];
    f32x4 accb[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        accb[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < C::NACC; ++i) acc[m][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // Loads are raw buffer loads on clip-relative resources (channels past the end read 0 without a branch) with
    // per-quad element offsets hoisted out of the chunk loop; the chunk cursor advances with carries instead of
    // divisions; prologue and masks run in store_chunk (fp32 MFMAs and VALU share the SIMD's fp32 pipe on gfx950,
    // so every address / predicate instruction is paid in MFMA time, and load_chunk must not wait on its loads).
    u32x4_t ry[C::Y_PER_T], ra[C::A_PER_T];
    unsigned ryi[C::Y_PER_T];                        // pool-row bytes of the dY quads (unpool)
    int y_thr[C::Y_PER_T], y_fq[C::Y_PER_T];         // element offset inside the clip; (fl << 8) | qc
    int a_thr[C::A_PER_T], a_rq[C::A_PER_T];         // element offset inside the clip; (r << 8) | qc
    float a_sc[C::A_PER_T], a_sh[C::A_PER_T];
    int r_ym[C::Y_PER_T], r_am[C::A_PER_T];          // per chunk in flight: valid elements of the quad (| parity << 8)
    constexpr unsigned OOB = 0x20000000u;            // element offset beyond every clip (x 4 = 2^31 bytes)
    const unsigned gclip = (unsigned)(a.Cout * Fg * a.T), xclip = (unsigned)(a.Cin * a.F * a.T);
#pragma unroll
    for (int i = 0; i < C::Y_PER_T; ++i) {
        const int q = tid + i * NT;
        const int cl = q / (FT * (TT / 4)), rem = q % (FT * (TT / 4));
        const int fl = rem / (TT / 4), qc = rem % (TT / 4);
        y_thr[i] = q < C::YQ ? ((cout0 + cl) * Fg + (unpool ? (fl >> 1) : fl)) * a.T + 4 * qc : (int)OOB;
        y_fq[i] = (fl << 8) | qc;
    }
#pragma unroll
    for (int i = 0; i < C::A_PER_T; ++i) {
        const int q = tid + i * NT;
        const int cl = q / (C::ROWS * C::QR), rem = q % (C::ROWS * C::QR);
        const int r = rem / C::QR, qc = rem % C::QR;
        const int cin = cin0 + cl;
        const bool ok = q < C::AQ && cin < a.Cin;
        a_thr[i] = ok ? (cin * a.F + r - PADH) * a.T + 4 * qc - C::HALO : (int)OOB;
        a_rq[i] = (r << 8) | qc;
        a_sc[i] = (pro && ok) ? a.scale[cin] : 0.f;
        a_sh[i] = (pro && ok) ? a.shift[cin] : 0.f;
    }
    int cb, cf, ct, sb, sf, st;                      // chunk to load next and the stride of gridDim.x chunks
    { int c = blockIdx.x; ct = c % nTt; c /= nTt; cf = c % nFt; cb = c / nFt; }
    { int c = gridDim.x; st = c % nTt; c /= nTt; sf = c % nFt; sb = c / nFt; }

    auto load_chunk = [&]() __attribute__((always_inline)) {
        const int t0 = ct * TT, f0 = cf * FT, b = cb;
        const int sl = a.seq_len ? min(a.seq_len[b], a.T) : a.T;
        const int tlim = pro ? sl : a.T;
        const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.g) + (size_t)b * gclip, 0, gclip * 4u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc(
            unpool ? const_cast<uint8_t*>(a.unpool_idx) + (size_t)b * gclip : nullptr, 0, unpool ? gclip : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.x) + (size_t)b * xclip, 0, xclip * 4u, 0x00020000);
        const int gch = (unpool ? (f0 >> 1) : f0) * a.T + t0, xch = f0 * a.T + t0;
        // ---- dY tile (un-pooled through the argmax byte in store_chunk)
#pragma unroll
        for (int i = 0; i < C::Y_PER_T; ++i) {
            const int fl = y_fq[i] >> 8, tq = t0 + 4 * (y_fq[i] & 0xff);
            const bool ok = f0 + fl < a.F && tq < a.T;
            const unsigned off = ok ? (unsigned)(y_thr[i] + gch) : OOB;
            ry[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_g, off * 4u, 0, 0);
            if (unpool) {
                if (vec) {
                    ryi[i] = __builtin_amdgcn_raw_buffer_load_b32(rs_i, off, 0, 0);
                } else {                             // a dword straddling the end of the clip would read 0 as a whole
                    unsigned w = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) w |= (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rs_i, off + e, 0, 0) << (8 * e);
                    ryi[i] = w;
                }
            }
            r_ym[i] = (ok ? min(a.T - tq, 4) : 0) | (((f0 + fl) & 1) << 8);
        }
        // ---- a tile = x with halo; prologue, zero padding (post-activation) and masks in store_chunk
#pragma unroll
        for (int i = 0; i < C::A_PER_T; ++i) {
            const int f = f0 - PADH + (a_rq[i] >> 8), tq = t0 - C::HALO + 4 * (a_rq[i] & 0xff);
            const bool ok = f >= 0 && f < a.F && tq >= 0 && tq < a.T;
            const unsigned off = ok ? (unsigned)(a_thr[i] + xch) : OOB;
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off * 4u, 0, 0);
            r_am[i] = ok ? min(max(tlim - tq, 0), 4) : 0;
        }
        ct += st; if (ct >= nTt) { ct -= nTt; ++cf; }
        cf += sf; if (cf >= nFt) { cf -= nFt; ++cb; }
        cb += sb;
    };
    auto store_chunk = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < C::Y_PER_T; ++i) {
            const int q = tid + i * NT;
            if (q < C::YQ) {
                const int cl = q / (FT * (TT / 4)), rem = q % (FT * (TT / 4));
                float v[4] = {__uint_as_float(ry[i].x), __uint_as_float(ry[i].y), __uint_as_float(ry[i].z), __uint_as_float(ry[i].w)};
                const int n_ok = r_ym[i] & 7;
                if (!vec) {
#pragma unroll
                    for (int e = 1; e < 4; ++e) v[e] = e < n_ok ? v[e] : 0.f;
                }
                if (unpool) {
                    const int par = r_ym[i] >> 8;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (int)((ryi[i] >> (8 * e)) & 0xffu) == par ? v[e] : 0.f;
                }
                float* d = dy_s + cl * C::PLANE_Y + rem * 4;
                d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
            }
        }
#pragma unroll
        for (int i = 0; i < C::A_PER_T; ++i) {
            const int q = tid + i * NT;
            if (q < C::AQ) {
                const int cl = q / (C::ROWS * C::QR), rem = q % (C::ROWS * C::QR);
                float v[4] = {__uint_as_float(ra[i].x), __uint_as_float(ra[i].y), __uint_as_float(ra[i].z), __uint_as_float(ra[i].w)};
                if (pro) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = fmaf(v[e], a_sc[i], a_sh[i]);
                        if (a.relu) v[e] = fmaxf(v[e], 0.f);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = e < r_am[i] ? v[e] : 0.f;
                float* d = a_s + cl * C::PLANE_A + rem * 4;
                d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
            }
        }
    };

    int chunk = blockIdx.x;
    if (chunk < nChunks) load_chunk();
    for (; chunk < nChunks; chunk += gridDim.x) {
        __syncthreads();
        store_chunk();
        __syncthreads();
        if (chunk + (int)gridDim.x < nChunks) load_chunk();
#pragma unroll
        for (int fl = 0; fl < FT; ++fl) {
#pragma unroll 2
            for (int tq = kwv * C::TQ_PER; tq < (kwv + 1) * C::TQ_PER; ++tq) {
                float af[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    af[m] = dy_s[((wave * MT + m) * 16 + lr) * C::PLANE_Y + fl * TT + tq * 4 + lq];
                if (TAPN) {
                    const int kh = lr / KW, kw = lr % KW;     // N index = tap
                    const float bf = (lr < KK) ? a_s[(fl + kh) * C::ROW + tq * 4 + lq + kw + (C::HALO - PADW)] : 0.f;
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m][0] = mfma16(af[m], bf, acc[m][0]);
                } else {
#pragma unroll
                    for (int g = 0; g < NCG; ++g)
#pragma unroll
                        for (int kk = 0; kk < KK; ++kk) {
                            const int kh = kk / KW, kw = kk % KW;
                            const float bf = a_s[(g * 16 + lr) * C::PLANE_A + (fl + kh) * C::ROW + tq * 4 + lq +
                                                 kw + (C::HALO - PADW)];
#pragma unroll
                            for (int m = 0; m < MT; ++m)
                                acc[m][g * KK + kk] = mfma16(af[m], bf, acc[m][g * KK + kk]);
                        }
                }
                if (do_bias) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) accb[m] = mfma16(af[m], 1.0f, accb[m]);
                }
            }
        }
    }

    // ---- reduce: transpose the block's partial dW through LDS, then row-contiguous atomics
    __syncthreads();
    float* out_s = smem;                             // [COUT_T][OUT_ROW]
    if (KWAVES > 1) {
        for (int i = tid; i < C::COUT_T * C::OUT_ROW; i += NT) out_s[i] = 0.f;
        __syncthreads();
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        if (TAPN) {
            if (lr < KK)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    lds_acc<KWAVES>(out_s + ((wave * MT + m) * 16 + lq * 4 + r) * C::OUT_ROW + lr, acc[m][0][r]);
        } else {
#pragma unroll
            for (int g = 0; g < NCG; ++g)
#pragma unroll
                for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        lds_acc<KWAVES>(out_s + ((wave * MT + m) * 16 + lq * 4 + r) * C::OUT_ROW + (g * 16 + lr) * KK + kk,
                                        acc[m][g * KK + kk][r]);
        }
    }
    __syncthreads();
    const int ncol = min(C::CIN_T, a.Cin - cin0) * KK;      // valid, contiguous part of each row
    const int slot = a.nslots > 1 ? (int)(blockIdx.x % a.nslots) : 0;
    float* dwp = a.dw + (size_t)slot * a.slot_w;
    for (int row = tid >> 6; row < C::COUT_T; row += NT / 64) {
        const int cout = cout0 + row;
        if (cout >= a.Cout) break;
        float* dst = dwp + ((size_t)cout * a.Cin + cin0) * KK;
        for (int col = lane; col < ncol; col += 64) atomicAdd(dst + col, out_s[row * C::OUT_ROW + col]);
    }
    if (do_bias && lr == 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cout = cout0 + (wave * MT + m) * 16 + lq * 4 + r;
                if (cout < a.Cout) atomicAdd(&a.db[(size_t)slot * a.slot_b + cout], accb[m][r]);
            }
    }
}

// Few-channel layers have so few gradient words that >1000 blocks adding into them serialise on a handful of L2
// lines; their blocks add into WGRAD_SLOTS partial copies instead, summed into the gradient by this kernel.
constexpr int WGRAD_SLOTS = 32, WGRAD_SLOT_MAX = 16384 + 64;      // floats per slot (weights + bias)

// ... and leaves the slots zero behind it: they are filled once per scratch registration, not once per launch
__global__ void wgrad_slot_reduce_kernel(float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db,
                                         int nw, int nb, int stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nw + nb) return;
    float v = 0.f;
#pragma unroll 8
    for (int sl = 0; sl < WGRAD_SLOTS; ++sl) {
        v += part[(size_t)sl * stride + i];
        part[(size_t)sl * stride + i] = 0.f;
    }
    if (i < nw) dw[i] += v; else if (db) db[i - nw] += v;
}

static_assert(PBSED_SCRATCH_FRONT == (size_t)WGRAD_SLOTS * WGRAD_SLOT_MAX, "common.h");
static float* wgrad_slot_scratch(hipStream_t s) { return scratch_zeroed_front(s, PBSED_SCRATCH_FRONT, PBSED_SCRATCH_FRONT); }

#include "conv_wgrad_s16.h"

// Shared launcher: K-split so that the grid is (close to) an integer number of full residency rounds (blocks per CU
// from the occupancy query, n_cu * occ resident slots, one or two rounds depending on how much K there is), slotted
// accumulation for small gradients.  C supplies TT, FT, CIN_T, COUT_T, KK, NT, LDS_FLOATS.
template <class C, class = void> struct wgrad_columns : std::false_type {};
template <class C> struct wgrad_columns<C, std::enable_if_t<C::COLUMNS>> : std::true_type {};

template <class C, class Kern>
static int launch_wgrad_cfg(Kern kern, const ConvWgradArgs& a_in, hipStream_t s) {
    ConvWgradArgs a = a_in;
    const int nTt = (a.T + C::TT - 1) / C::TT, nFt = (a.F + C::FT - 1) / C::FT;
    const int nChunks = a.B * nFt * nTt;
    const int gy = (a.Cin + C::CIN_T - 1) / C::CIN_T;
    const int gz = (a.Cout + C::COUT_T - 1) / C::COUT_T;
    const size_t lds = C::LDS_FLOATS * sizeof(float);
    PBSED_DYN_LDS_ONCE(kern, lds);
    static int occ_dev[64] = {0};                     // blocks of this kernel a CU holds, per device ordinal
    int dev = 0;
    PBSED_HIP_TRY(hipGetDevice(&dev), "hipGetDevice");
    int& occ = occ_dev[dev & 63];
    if (occ == 0 && (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, C::NT, lds) != hipSuccess || occ < 1)) occ = 2;
    const int slots = launch_cus() * occ;             // resident block slots of this kernel (on the caller's CU budget, if it set one)
    int split = slots / (gy * gz);
    if (split >= 8) split &= ~7;       // multiple of 8: blocks sharing a spatial chunk share an XCD's L2
    if (split < 1) split = 1;
    // one residency round: a second round (twice the blocks, half the chunks each) doubles the atomics of the final
    // reduction for nothing when the chunks divide evenly (128->128: 0.465 -> 0.452 ms)
    const int nw = a.Cout * a.Cin * C::KK, nb = a.Cout;
    const bool slot_ok = nw + nb <= WGRAD_SLOT_MAX;
    const int cap = (slot_ok ? 4096 : 1024) / (gy * gz);                              // atomics-per-address cap
    if (split > cap) split = cap > 0 ? cap : 1;
    if (split > nChunks) split = nChunks;
    if constexpr (wgrad_columns<C>::value) {                          // column-walking kernels split over (clip, column) units
        const int units = a.B * nTt * (C::KK <= 3 ? a.F : 1);       // the Conv1d form takes (clip, row, column) units, the 3x3 form walks the rows
        if (split > units) split = units;
        // PBSED_WGRAD_XCD_COLS=1 (off by default: built while the GPU pool was closed to the build, NOT measured): block x of the 3x3
        // column walk takes list position (x % 8) * (split / 8) + x / 8 instead of x, i.e. the blocks of ONE XCD walk split / 8
        // ADJACENT columns (t ranges of one clip) at the same time.  A column's rows are 128-byte segments (+ 16 bytes either
        // side for dY) of 2 000-byte tensor rows: every segment straddles cache lines it shares with the neighbouring columns,
        // which the default map deals to eight different L2s (profiles/r05_pmc_hbm_traffic.csv: 512 MB fetched per 128->128
        // launch for 213 MB of tensors).
        const char* xcd_cols = getenv("PBSED_WGRAD_XCD_COLS");          // read per launch: an A/B inside one process
        a.xcd_cols = (xcd_cols && xcd_cols[0] == '1' && C::KK == 9 && split >= 16 && split % 8 == 0) ? 1 : 0;
    }
    dim3 grid(split, gy, gz);
    float* scratch = (slot_ok && split >= 4 * WGRAD_SLOTS) ? wgrad_slot_scratch(s) : nullptr;
    if (scratch) {
        const int stride = (nw + nb + 63) / 64 * 64;
        a.dw = scratch; a.db = a_in.db ? scratch + nw : nullptr;
        a.nslots = WGRAD_SLOTS; a.slot_w = stride; a.slot_b = stride;
        hipLaunchKernelGGL(kern, grid, dim3(C::NT), lds, s, a);
        hipLaunchKernelGGL(wgrad_slot_reduce_kernel, dim3((nw + nb + 255) / 256), dim3(256), 0, s, scratch, a_in.dw, a_in.db,
                           nw, nb, stride);
        return check_launch("conv_wgrad(slotted)");
    }
    hipLaunchKernelGGL(kern, grid, dim3(C::NT), lds, s, a);
    return check_launch("conv_wgrad");
}

template <int KH, int KW, int WAVES, int MT, int NCG, bool TAPN = false, int KWAVES = 1, int FT_ = 0>
static int launch_wgrad(const ConvWgradArgs& a, hipStream_t s) {
    return launch_wgrad_cfg<WgradCfg<KH, KW, WAVES, MT, NCG, TAPN, KWAVES, FT_>>(
        conv_wgrad_kernel<KH, KW, WAVES, MT, NCG, TAPN, KWAVES, FT_>, a, s);
}

// ============================================================================================
// 3x3 weight gradient with the time axis in the Winograd F(4,3) domain (the bilinear algorithm of conv_wino.hip
// transposed): per tile of 4 outputs   dU_xi[kh] += (A dy)_xi * (B^T d)_xi(row f+kh-1),   dw[kh][0..2] = G^T dU[kh].
// 18 products per 4 output positions instead of 36.  GEMM per point and kernel row: M = Cout (A = transformed dY),
// N = 16 cin (B = transformed prologue(x)), K = tiles; both transforms are applied between the global load and the
// LDS store, the G^T transform on the accumulators before the common LDS-transposed atomic reduction.  The bias
// gradient is the (A dy)_1 = dy0+dy1+dy2+dy3 plane against a ones-vector.
// ============================================================================================
struct WinoWgradCfg {
    static constexpr int FT = 2, TT = 64, KK = 9, NT = 256;
    static constexpr int COUT_T = 64, CIN_T = 16, TILES = TT / 4, ROWS = FT + 2;
    static constexpr int PLANE_P = pad2mod32(FT * TILES), PLANE_V = pad2mod32(ROWS * TILES);
    static constexpr int P_FLOATS = 6 * COUT_T * PLANE_P, V_FLOATS = 6 * CIN_T * PLANE_V;
    static constexpr int YQ_PER_T = COUT_T * FT * TILES / NT;        // dY tiles (float4) per thread
    static constexpr int OUT_ROW = CIN_T * KK;
    static constexpr int LDS_FLOATS = cmax(P_FLOATS + V_FLOATS, COUT_T * OUT_ROW);
};

__global__ __launch_bounds__(256, 2) void conv_wgrad_wino_kernel(ConvWgradArgs a) {
    using C = WinoWgradCfg;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* p_s = smem;                               // [6][COUT_T][PLANE_P]   (row fl, tile) along the plane
    float* v_s = smem + C::P_FLOATS;                 // [6][CIN_T][PLANE_V]    (halo row, tile)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane >> 4, lr = lane & 15;
    const int cin0 = blockIdx.y * C::CIN_T, cout0 = blockIdx.z * C::COUT_T;
    const int nTt = (a.T + C::TT - 1) / C::TT, nFt = (a.F + C::FT - 1) / C::FT;
    const int nChunks = a.B * nFt * nTt;
    const bool pro = a.scale != nullptr;
    const bool do_bias = (a.db != nullptr) && (blockIdx.y == 0);
    const bool unpool = a.unpool_idx != nullptr;
    const int Fg = unpool ? a.F / 2 : a.F;
    const bool vec = (a.T & 3) == 0;

    f32x4 acc[3][6];
    f32x4 accb = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int x = 0; x < 6; ++x) acc[kh][x] = f32x4{0.f, 0.f, 0.f, 0.f};

    // this thread's x item: (cin ic, halo row ir, tile quad iq) and its 8 dY tiles q = tid + i*256 -> (cout, row, tile)
    const int iq = tid & 3, ir = (tid >> 2) & 3, ic = tid >> 4;
    u32x4_t ry[C::YQ_PER_T];
    unsigned ryi[C::YQ_PER_T];                       // pool-row bytes of the same tiles (unpool)
    unsigned rin[18];
    int r_ehi = 0;                                   // valid elements of the x item in flight: e < r_ehi (and e = 0: r_e0)
    bool r_e0 = false;
    int r_ylim = 4;                                  // valid elements of the dY tiles in flight (T % 4 != 0 only)

    // Chunk cursor (clip, row block, time block) of the chunk to load next, advanced by gridDim.x chunks per step with
    // carries instead of divisions; every load is a raw buffer load on a clip-relative resource, so channels past the
    // end, halo rows outside the plane and tiles past the row end read 0 without branches and the per-chunk address
    // arithmetic is a handful of adds (VALU instructions are paid in fp32-MFMA time on gfx950).
    int cb, cf, ct;
    { int c = blockIdx.x; ct = c % nTt; c /= nTt; cf = c % nFt; cb = c / nFt; }
    int sb, sf, st;
    { int c = gridDim.x; st = c % nTt; c /= nTt; sf = c % nFt; sb = c / nFt; }
    constexpr unsigned OOB = 0x80000000u;
    const unsigned gclip = (unsigned)(a.Cout * Fg * a.T), xclip = (unsigned)(a.Cin * a.F * a.T);
    const int ytile = tid & 15, yfl = (tid >> 4) & 1;
    const unsigned gthr = (unsigned)(((cout0 + (tid >> 5)) * Fg + (unpool ? 0 : yfl)) * a.T + 4 * ytile);
    const unsigned gstride = (unsigned)(8 * Fg * a.T);                       // cout + 8 per dY tile i
    const int xthr = (((cin0 + ic) * a.F + ir - 1) * a.T + 16 * iq);
    float sc = 1.f, sh = 0.f;
    if (pro && cin0 + ic < a.Cin) { sc = a.scale[cin0 + ic]; sh = a.shift[cin0 + ic]; }

    auto load_chunk = [&]() __attribute__((always_inline)) {
        const int t0 = ct * C::TT, f0 = cf * C::FT, b = cb;
        const int sl = a.seq_len ? min(a.seq_len[b], a.T) : a.T;
        const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.g) + (size_t)b * gclip, 0, gclip * 4u, 0x00020000);
        const unsigned gch = (unsigned)((unpool ? (f0 >> 1) : f0) * a.T + t0);
        const unsigned goff = (t0 + 4 * ytile < a.T) ? gthr + gch : OOB / 4u;  // element offset of dY tile 0 (x 4 below)
#pragma unroll
        for (int i = 0; i < C::YQ_PER_T; ++i)
            ry[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_g, (goff + i * gstride) * 4u, 0, 0);
        if (unpool) {
            const __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<uint8_t*>(a.unpool_idx) + (size_t)b * gclip, 0, gclip, 0x00020000);
#pragma unroll
            for (int i = 0; i < C::YQ_PER_T; ++i) {
                if (vec) {
                    ryi[i] = __builtin_amdgcn_raw_buffer_load_b32(rs_i, goff + i * gstride, 0, 0);
                } else {                             // a dword straddling the end of the clip would read 0 as a whole
                    unsigned w = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        w |= (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rs_i, goff + i * gstride + e, 0, 0) << (8 * e);
                    ryi[i] = w;
                }
            }
        }
        // raw inputs t = tq0 - 1 .. tq0 + 16 of (cin, halo row); prologue / mask / transform happen in store_chunk
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.x) + (size_t)b * xclip, 0, xclip * 4u, 0x00020000);
        const int f = f0 - 1 + ir, tq0 = t0 + 16 * iq;
        const bool row_ok = f >= 0 && f < a.F;
        const unsigned xoff = row_ok ? (unsigned)(xthr + f0 * a.T + t0) * 4u : OOB;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u32x4_t xv = __builtin_amdgcn_raw_buffer_load_b128(rs_x, xoff + q * 16, 0, 0);
            rin[1 + 4 * q] = xv.x; rin[2 + 4 * q] = xv.y; rin[3 + 4 * q] = xv.z; rin[4 + 4 * q] = xv.w;
        }
        rin[0] = __builtin_amdgcn_raw_buffer_load_b32(rs_x, xoff - 4u, 0, 0);
        rin[17] = __builtin_amdgcn_raw_buffer_load_b32(rs_x, xoff + 64u, 0, 0);
        r_ehi = row_ok ? max((pro ? sl : a.T) - tq0 + 1, 0) : 0;
        r_e0 = tq0 > 0 && r_ehi > 0;
        r_ylim = a.T - (t0 + 4 * ytile);
        ct += st; if (ct >= nTt) { ct -= nTt; ++cf; }
        cf += sf; if (cf >= nFt) { cf -= nFt; ++cb; }
        cb += sb;
    };
    auto store_chunk = [&]() __attribute__((always_inline)) {
        // A dy: one tile (float4) -> 6 points
#pragma unroll
        for (int i = 0; i < C::YQ_PER_T; ++i) {
            const int q = tid + i * C::NT;
            const int pos = q & 31, cl = q >> 5;            // pos = fl*16 + tile
            float y0 = __uint_as_float(ry[i].x), y1 = __uint_as_float(ry[i].y), y2 = __uint_as_float(ry[i].z),
                  y3 = __uint_as_float(ry[i].w);
            if (!vec) { y1 = r_ylim > 1 ? y1 : 0.f; y2 = r_ylim > 2 ? y2 : 0.f; y3 = r_ylim > 3 ? y3 : 0.f; }
            if (unpool) {                            // tile row f0 + yfl: keep the elements whose window maximum was this row
                const unsigned iv = ryi[i];
                y0 = (int)(iv & 0xffu) == yfl ? y0 : 0.f; y1 = (int)((iv >> 8) & 0xffu) == yfl ? y1 : 0.f;
                y2 = (int)((iv >> 16) & 0xffu) == yfl ? y2 : 0.f; y3 = (int)(iv >> 24) == yfl ? y3 : 0.f;
            }
            const float e02 = y0 + y2, o13 = y1 + y3, e04 = y0 + 4.f * y2, o28 = 2.f * y1 + 8.f * y3;
            float* d = p_s + cl * C::PLANE_P + pos;
            d[0 * C::COUT_T * C::PLANE_P] = y0;
            d[1 * C::COUT_T * C::PLANE_P] = e02 + o13;
            d[2 * C::COUT_T * C::PLANE_P] = e02 - o13;
            d[3 * C::COUT_T * C::PLANE_P] = e04 + o28;
            d[4 * C::COUT_T * C::PLANE_P] = e04 - o28;
            d[5 * C::COUT_T * C::PLANE_P] = y3;
        }
        // B^T d of prologue(x)
        float dv[18];
#pragma unroll
        for (int e = 0; e < 18; ++e) {
            float u = __uint_as_float(rin[e]);
            if (pro) {
                u = fmaf(u, sc, sh);
                if (a.relu) u = fmaxf(u, 0.f);
            }
            dv[e] = (e == 0 ? r_e0 : e < r_ehi) ? u : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float d0 = dv[4 * i], d1 = dv[4 * i + 1], d2 = dv[4 * i + 2], d3 = dv[4 * i + 3], d4 = dv[4 * i + 4],
                        d5 = dv[4 * i + 5];
            float* d = v_s + ic * C::PLANE_V + ir * 16 + iq * 4 + i;
            d[0 * C::CIN_T * C::PLANE_V] = 4.f * d0 - 5.f * d2 + d4;
            d[1 * C::CIN_T * C::PLANE_V] = -4.f * (d1 + d2) + d3 + d4;
            d[2 * C::CIN_T * C::PLANE_V] = 4.f * (d1 - d2) - d3 + d4;
            d[3 * C::CIN_T * C::PLANE_V] = -2.f * d1 - d2 + 2.f * d3 + d4;
            d[4 * C::CIN_T * C::PLANE_V] = 2.f * d1 - d2 - 2.f * d3 + d4;
            d[5 * C::CIN_T * C::PLANE_V] = 4.f * d1 - 5.f * d3 + d5;
        }
    };

    int chunk = blockIdx.x;
    if (chunk < nChunks) load_chunk();
    for (; chunk < nChunks; chunk += gridDim.x) {
        __syncthreads();
        store_chunk();
        __syncthreads();
        if (chunk + (int)gridDim.x < nChunks) load_chunk();
#pragma unroll
        for (int kk = 0; kk < C::TILES / 4; ++kk) {
#pragma unroll
            for (int x = 0; x < 6; ++x) {
                float af[2], bf[4];
                const float* pp = p_s + (x * C::COUT_T + wave * 16 + lr) * C::PLANE_P + kk * 4 + lq;
                af[0] = pp[0]; af[1] = pp[16];
                const float* vp = v_s + (x * C::CIN_T + lr) * C::PLANE_V + kk * 4 + lq;
#pragma unroll
                for (int r = 0; r < 4; ++r) bf[r] = vp[r * 16];
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    acc[kh][x] = mfma16(af[0], bf[kh], acc[kh][x]);
                    acc[kh][x] = mfma16(af[1], bf[kh + 1], acc[kh][x]);
                }
                if (x == 1 && do_bias) {
                    accb = mfma16(af[0], 1.0f, accb);
                    accb = mfma16(af[1], 1.0f, accb);
                }
            }
        }
    }

    // ---- G^T on the accumulators, then transpose the block's partial dW through LDS and reduce with row-contiguous atomics
    __syncthreads();
    float* out_s = smem;                             // [COUT_T][OUT_ROW]
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float u0 = acc[kh][0][r], u1 = acc[kh][1][r], u2 = acc[kh][2][r], u3 = acc[kh][3][r], u4 = acc[kh][4][r],
                        u5 = acc[kh][5][r];
            const float s12 = u1 + u2, s34 = u3 + u4;
            float* o = out_s + (wave * 16 + lq * 4 + r) * C::OUT_ROW + lr * 9 + kh * 3;
            o[0] = .25f * u0 - s12 * (1.f / 6.f) + s34 * (1.f / 24.f);
            o[1] = (u2 - u1) * (1.f / 6.f) + (u3 - u4) * (1.f / 12.f);
            o[2] = (s34 - s12) * (1.f / 6.f) + u5;
        }
    __syncthreads();
    const int ncol = min(C::CIN_T, a.Cin - cin0) * 9;      // valid, contiguous part of each row
    const int slot = a.nslots > 1 ? (int)(blockIdx.x % a.nslots) : 0;
    float* dwp = a.dw + (size_t)slot * a.slot_w;
    for (int row = tid >> 6; row < C::COUT_T; row += C::NT / 64) {
        const int cout = cout0 + row;
        if (cout >= a.Cout) break;
        float* dst = dwp + ((size_t)cout * a.Cin + cin0) * 9;
        for (int col = lane; col < ncol; col += 64) atomicAdd(dst + col, out_s[row * C::OUT_ROW + col]);
    }
    if (do_bias && lr == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cout = cout0 + wave * 16 + lq * 4 + r;
            if (cout < a.Cout) atomicAdd(&a.db[(size_t)slot * a.slot_b + cout], accb[r]);
        }
    }
}


// ============================================================================================
// Weight gradient on bf16 MFMA (v_mfma_f32_16x16x32_bf16) - the compute dtype of BASELINE.json config 3.
// dW[cout][cin][kh][kw] = sum over (b, f, t) of dY[b][cout][f][t] * prologue(x)[b][cin][f+kh-1][t+kw-1]:
// M = cout (A = dY), N = cin (B = x), K = 32 CONSECUTIVE t of one row per MFMA.  With time as the contraction index
// both operands are read the way they lie in memory (T innermost) - no transposition anywhere: the tiles are staged as
// bf16 [channel][row][t] images and a lane's 8 k-values are one aligned 16-byte LDS read.  The kw = 0 / 2 operands are
// the centre operand shifted by one element: built in registers from the centre read plus the two neighbouring
// elements (five v_alignbit per row), so the x tile is stored once.  Block = 64 cout x 32 cin, wave = (16 cin, 32 cout),
// all KH*KW taps in its accumulators; loader / chunk walk / slotted LDS-transposed atomic reduction as conv_wgrad_kernel
// (fp32 accumulation, fp32 gradients).
// ============================================================================================
typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wg_bf16x2 __attribute__((ext_vector_type(2)));
typedef float wg_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned wg_pack(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(wg_f32x2{lo, hi}, wg_bf16x2));
}
__device__ __forceinline__ f32x4 wg_mfma(u32x4_t a, u32x4_t b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wg_bf16x8, a), __builtin_bit_cast(wg_bf16x8, b), c, 0, 0, 0);
}
constexpr int pad8mod64(int n) { return ((n + 55) / 64) * 64 + 8; }      // halfs: plane stride = 16 bytes mod 128

// NS = 1: plain bf16 operands; NS = 3: exact three-way splits of dY and x (Bf3 in common.h), six part products per product:
// fp32-class gradients - the Conv1d layers of the fp32 path.
template <int KH, int KW, int MH, int NS = 1, int FT_ = 0>
struct WgradB16Cfg {
    // MH groups of 32 output channels per block (two waves each: the two 16-channel halves of the 32-channel cin tile).
    // 3x3: MH = 4 (128 cout x 32 cin, 512 threads): the kernel is bound by the bytes a block pulls per MFMA, and the
    // wider cout tile halves the re-reads of x (measured at 64: no faster than the fp32 Winograd kernel).
    static constexpr int FT = FT_ ? FT_ : (KH == 3) ? 2 : 1, TT = (KH == 3) ? 64 : 128;
    static constexpr int KK = KH * KW, NT = MH * 128;
    static constexpr int COUT_T = MH * 32, CIN_T = 32;
    static constexpr int HALO = (KW > 1) ? 8 : 0;                          // elements; keeps the centre read 16-byte aligned
    static constexpr int ROWS = FT + KH - 1, ROW = TT + 2 * HALO, QR = ROW / 4;
    static constexpr int PLANE_Y = pad8mod64(FT * TT), PLANE_A = pad8mod64(ROWS * ROW);    // halfs
    static constexpr int YQ = COUT_T * FT * (TT / 4), AQ = CIN_T * ROWS * QR;
    static constexpr int Y_PER_T = (YQ + NT - 1) / NT, A_PER_T = (AQ + NT - 1) / NT;
    static constexpr int OUT_ROW = CIN_T * KK + 1;
    static constexpr int OUT_ROWS = 64;                                    // the LDS transpose of the result runs 64 cout rows at a time
    static constexpr int LDS_FLOATS = cmax(NS * (COUT_T * PLANE_Y + CIN_T * PLANE_A) / 2 + 2 * CIN_T, OUT_ROWS * OUT_ROW);
};

template <int KH, int KW, int MH, int NS = 1, int FT_ = 0>
__global__ __launch_bounds__(MH * 128) void conv_wgrad_bf16_kernel(ConvWgradArgs a) {
    using C = WgradB16Cfg<KH, KW, MH, NS, FT_>;
    constexpr int Y_PART = C::COUT_T * C::PLANE_Y, A_PART = C::CIN_T * C::PLANE_A;       // halfs per operand part
    constexpr int FT = C::FT, TT = C::TT, KK = C::KK, NT = C::NT;
    constexpr int PADH = (KH - 1) / 2, PADW = (KW - 1) / 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned short* dy_s = reinterpret_cast<unsigned short*>(smem);              // [NS][COUT_T][PLANE_Y]
    unsigned short* a_s = dy_s + NS * Y_PART;                                      // [NS][CIN_T][PLANE_A]
    float* sc_s = reinterpret_cast<float*>(a_s + NS * A_PART);                     // [CIN_T] scale, [CIN_T] shift

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = wave & 1, mh = wave >> 1;                                        // cin sub-tile / cout half of this wave
    const int lq = lane >> 4, lr = lane & 15;
    const int cin0 = blockIdx.y * C::CIN_T, cout0 = blockIdx.z * C::COUT_T;
    const int nTt = (a.T + TT - 1) / TT, nFt = (a.F + FT - 1) / FT;
    const int nChunks = a.B * nFt * nTt;
    const bool pro = a.scale != nullptr;
    const bool do_bias = (a.db != nullptr) && (blockIdx.y == 0) && g == 0;
    const bool unpool = a.unpool_idx != nullptr;
    const int Fg = unpool ? a.F / 2 : a.F;
    const bool vec = (a.T & 3) == 0;
    if (tid < C::CIN_T) {
        const int cin = cin0 + tid;
        sc_s[tid] = (pro && cin < a.Cin) ? a.scale[cin] : 0.f;
        sc_s[C::CIN_T + tid] = (pro && cin < a.Cin) ? a.shift[cin] : 0.f;
    }

    f32x4 acc[2][KK], accb[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        accb[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < KK; ++i) acc[m][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    u32x4_t ry[C::Y_PER_T], ra[C::A_PER_T];
    unsigned ryi[C::Y_PER_T];
    int r_ym[C::Y_PER_T], r_am[C::A_PER_T];
    constexpr unsigned OOB = 0x20000000u;            // element offset beyond every clip (x 4 = 2^31 bytes)
    const unsigned gclip = (unsigned)(a.Cout * Fg * a.T), xclip = (unsigned)(a.Cin * a.F * a.T);

    auto load_chunk = [&](int chunk) __attribute__((always_inline)) {
        const int ct = chunk % nTt, c1 = chunk / nTt, cf = c1 % nFt, b = c1 / nFt;
        const int t0 = ct * TT, f0 = cf * FT;
        const int sl = a.seq_len ? min(a.seq_len[b], a.T) : a.T;
        const int tlim = pro ? sl : a.T;
        const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.g) + (size_t)b * gclip, 0, gclip * 4u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc(
            unpool ? const_cast<uint8_t*>(a.unpool_idx) + (size_t)b * gclip : nullptr, 0, unpool ? gclip : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.x) + (size_t)b * xclip, 0, xclip * 4u, 0x00020000);
#pragma unroll
        for (int i = 0; i < C::Y_PER_T; ++i) {
            const int q = tid + i * NT;
            const int cl = q / (FT * (TT / 4)), rem = q % (FT * (TT / 4));
            const int fl = rem / (TT / 4), qc = rem % (TT / 4);
            const int f = f0 + fl, tq = t0 + 4 * qc, cout = cout0 + cl;
            const bool ok = q < C::YQ && cout < a.Cout && f < a.F && tq < a.T;
            const unsigned off = ok ? (unsigned)((cout * Fg + (unpool ? (f >> 1) : f)) * a.T + tq) : OOB;
            ry[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_g, off * 4u, 0, 0);
            if (unpool) {
                if (vec) {
                    ryi[i] = __builtin_amdgcn_raw_buffer_load_b32(rs_i, off, 0, 0);
                } else {
                    unsigned w = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) w |= (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rs_i, off + e, 0, 0) << (8 * e);
                    ryi[i] = w;
                }
            }
            r_ym[i] = (ok ? min(a.T - tq, 4) : 0) | ((f & 1) << 8);
        }
#pragma unroll
        for (int i = 0; i < C::A_PER_T; ++i) {
            const int q = tid + i * NT;
            const int cl = q / (C::ROWS * C::QR), rem = q % (C::ROWS * C::QR);
            const int r = rem / C::QR, qc = rem % C::QR;
            const int f = f0 - PADH + r, tq = t0 - C::HALO + 4 * qc, cin = cin0 + cl;
            const bool ok = q < C::AQ && cin < a.Cin && f >= 0 && f < a.F && tq >= 0 && tq < a.T;
            const unsigned off = ok ? (unsigned)((cin * a.F + f) * a.T + tq) : OOB;
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off * 4u, 0, 0);
            r_am[i] = ok ? min(max(tlim - tq, 0), 4) : 0;
        }
    };
    auto store_chunk = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < C::Y_PER_T; ++i) {
            const int q = tid + i * NT;
            if (q < C::YQ) {
                const int cl = q / (FT * (TT / 4)), rem = q % (FT * (TT / 4));
                float v[4] = {__uint_as_float(ry[i].x), __uint_as_float(ry[i].y), __uint_as_float(ry[i].z), __uint_as_float(ry[i].w)};
                const int n_ok = r_ym[i] & 7;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = e < n_ok ? v[e] : 0.f;
                if (unpool) {
                    const int par = r_ym[i] >> 8;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (int)((ryi[i] >> (8 * e)) & 0xffu) == par ? v[e] : 0.f;
                }
                if constexpr (NS == 3) {
                    unsigned h0, m0, l0, h1, m1, l1;
                    split3_pair(v[0], v[1], h0, m0, l0);
                    split3_pair(v[2], v[3], h1, m1, l1);
                    *reinterpret_cast<uint2*>(dy_s + cl * C::PLANE_Y + rem * 4) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2*>(dy_s + Y_PART + cl * C::PLANE_Y + rem * 4) = make_uint2(m0, m1);
                    *reinterpret_cast<uint2*>(dy_s + 2 * Y_PART + cl * C::PLANE_Y + rem * 4) = make_uint2(l0, l1);
                } else {
                    uint2 o;
                    o.x = wg_pack(v[0], v[1]); o.y = wg_pack(v[2], v[3]);
                    *reinterpret_cast<uint2*>(dy_s + cl * C::PLANE_Y + rem * 4) = o;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < C::A_PER_T; ++i) {
            const int q = tid + i * NT;
            if (q < C::AQ) {
                const int cl = q / (C::ROWS * C::QR), rem = q % (C::ROWS * C::QR);
                float v[4] = {__uint_as_float(ra[i].x), __uint_as_float(ra[i].y), __uint_as_float(ra[i].z), __uint_as_float(ra[i].w)};
                if (pro) {
                    const float sc = sc_s[cl], sh = sc_s[C::CIN_T + cl];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = fmaf(v[e], sc, sh);
                        if (a.relu) v[e] = fmaxf(v[e], 0.f);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = e < r_am[i] ? v[e] : 0.f;        // zero padding is post-activation
                if constexpr (NS == 3) {
                    unsigned h0, m0, l0, h1, m1, l1;
                    split3_pair(v[0], v[1], h0, m0, l0);
                    split3_pair(v[2], v[3], h1, m1, l1);
                    *reinterpret_cast<uint2*>(a_s + cl * C::PLANE_A + rem * 4) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2*>(a_s + A_PART + cl * C::PLANE_A + rem * 4) = make_uint2(m0, m1);
                    *reinterpret_cast<uint2*>(a_s + 2 * A_PART + cl * C::PLANE_A + rem * 4) = make_uint2(l0, l1);
                } else {
                    uint2 o;
                    o.x = wg_pack(v[0], v[1]); o.y = wg_pack(v[2], v[3]);
                    *reinterpret_cast<uint2*>(a_s + cl * C::PLANE_A + rem * 4) = o;
                }
            }
        }
    };

    const u32x4_t ones = u32x4_t{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    int chunk = blockIdx.x;
    if (chunk < nChunks) load_chunk(chunk);
    for (; chunk < nChunks; chunk += gridDim.x) {
        __syncthreads();
        store_chunk();
        __syncthreads();
        if (chunk + (int)gridDim.x < nChunks) load_chunk(chunk + gridDim.x);      // in flight during the MFMAs below
#pragma unroll
        for (int fl = 0; fl < FT; ++fl) {
#pragma unroll
            for (int ks = 0; ks < TT / 32; ++ks) {
                u32x4_t af[2][NS];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int p = 0; p < NS; ++p)
                        af[m][p] = *reinterpret_cast<const u32x4_t*>(dy_s + p * Y_PART + ((mh * 2 + m) * 16 + lr) * C::PLANE_Y + fl * TT + ks * 32 + lq * 8);
                auto prod = [&](const u32x4_t (&A)[NS], const u32x4_t (&Bq)[NS], f32x4 c) __attribute__((always_inline)) {
                    if constexpr (NS == 3) return mfma_x3(Bf3{A[0], A[1], A[2]}, Bf3{Bq[0], Bq[1], Bq[2]}, c);
                    else return wg_mfma(A[0], Bq[0], c);
                };
#pragma unroll
                for (int kh = 0; kh < KH; ++kh) {
                    u32x4_t c[NS], left[NS], right[NS];
#pragma unroll
                    for (int p = 0; p < NS; ++p) {
                        const unsigned short* row = a_s + p * A_PART + (g * 16 + lr) * C::PLANE_A + (fl + kh) * C::ROW + C::HALO + ks * 32 + lq * 8;
                        c[p] = *reinterpret_cast<const u32x4_t*>(row);
                        if (KW > 1) {
                            const unsigned xm1 = row[-1], x8 = row[8];
                            const unsigned s1 = __builtin_amdgcn_alignbit(c[p].y, c[p].x, 16), s2 = __builtin_amdgcn_alignbit(c[p].z, c[p].y, 16),
                                           s3 = __builtin_amdgcn_alignbit(c[p].w, c[p].z, 16);
                            left[p] = u32x4_t{(c[p].x << 16) | xm1, s1, s2, s3};                    // x[t-1 .. t+6]
                            right[p] = u32x4_t{s1, s2, s3, (x8 << 16) | (c[p].w >> 16)};           // x[t+1 .. t+8]
                        }
                    }
                    if (KW == 1) {
#pragma unroll
                        for (int m = 0; m < 2; ++m) acc[m][kh] = prod(af[m], c, acc[m][kh]);
                    } else if constexpr (NS == 3) {
                        // the six part products of the six (m, kw) accumulators of this kernel row round-robin, smallest
                        // first: consecutive MFMAs never wait for each other's accumulator
#pragma unroll
                        for (int pp = 0; pp < 6; ++pp) {
                            const int pa = pp == 0 ? 2 : pp == 1 ? 0 : pp == 2 ? 1 : pp == 3 ? 1 : 0;     // lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
                            const int pb = pp == 0 ? 0 : pp == 1 ? 2 : pp == 2 ? 1 : pp == 3 ? 0 : pp == 4 ? 1 : 0;
#pragma unroll
                            for (int m = 0; m < 2; ++m) {
                                acc[m][kh * KW + 0] = wg_mfma(af[m][pa], left[pb], acc[m][kh * KW + 0]);
                                acc[m][kh * KW + 1] = wg_mfma(af[m][pa], c[pb], acc[m][kh * KW + 1]);
                                acc[m][kh * KW + 2] = wg_mfma(af[m][pa], right[pb], acc[m][kh * KW + 2]);
                            }
                        }
                    } else {
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            acc[m][kh * KW + 0] = prod(af[m], left, acc[m][kh * KW + 0]);
                            acc[m][kh * KW + 1] = prod(af[m], c, acc[m][kh * KW + 1]);
                            acc[m][kh * KW + 2] = prod(af[m], right, acc[m][kh * KW + 2]);
                        }
                    }
                }
                if (do_bias) {
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int p = 0; p < NS; ++p) accb[m] = wg_mfma(af[m][NS - 1 - p], ones, accb[m]);
                }
            }
        }
    }

    // ---- reduce: transpose the block's partial dW through LDS, then row-contiguous atomics (as conv_wgrad_kernel)
    __syncthreads();
    float* out_s = smem;                             // [OUT_ROWS][OUT_ROW]
    const int ncol = min(C::CIN_T, a.Cin - cin0) * KK;
    const int slot = a.nslots > 1 ? (int)(blockIdx.x % a.nslots) : 0;
    float* dwp = a.dw + (size_t)slot * a.slot_w;
#pragma unroll 1
    for (int part = 0; part < C::COUT_T / C::OUT_ROWS; ++part) {
        if (part > 0) __syncthreads();
        if ((mh * 32) / C::OUT_ROWS == part) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        out_s[((mh * 32) % C::OUT_ROWS + m * 16 + lq * 4 + r) * C::OUT_ROW + (g * 16 + lr) * KK + kk] = acc[m][kk][r];
        }
        __syncthreads();
        for (int row = wave; row < C::OUT_ROWS; row += NT / 64) {
            const int cout = cout0 + part * C::OUT_ROWS + row;
            if (cout >= a.Cout) break;
            float* dst = dwp + ((size_t)cout * a.Cin + cin0) * KK;
            for (int col = lane; col < ncol; col += 64) atomicAdd(dst + col, out_s[row * C::OUT_ROW + col]);
        }
    }
    if (do_bias && lr == 0) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cout = cout0 + (mh * 2 + m) * 16 + lq * 4 + r;
                if (cout < a.Cout) atomicAdd(&a.db[(size_t)slot * a.slot_b + cout], accb[m][r]);
            }
    }
}

// ============================================================================================
// 3x3 weight gradient of the fp32 path on the bf16 MFMA with exact three-way operand splits, PRODUCER / CONSUMER form.
// Same arithmetic as conv_wgrad_bf16_kernel<3, 3, *, 3> (M = cout, N = cin, K = 32 consecutive t of one row; six part products
// per product, fp32 accumulation: fp32-class gradients), but the staging no longer alternates with the MFMAs: four producer
// waves build the three-part LDS images one step ahead (global loads two steps ahead; un-pool / BN-apply + ReLU + mask;
// splits; 8-byte LDS stores) while four consumer waves - one per SIMD, 32 cout x 32 cin x 9 taps = 144 accumulator registers
// each - multiply; one block barrier per step.
// A block walks COLUMNS: for a fixed (clip, 32-t range) the steps run down the rows f = 0 .. F-1, so an x row is loaded and
// split ONCE and used by the three kernel rows of three consecutive steps (ring of four row slots + one all-zero slot for
// the rows above / below the plane); dY of a step is double-buffered.  What a step pulls from L2 is then 128 dY rows + 32 x
// rows of 128 / 192 bytes instead of 128 + 96: these scattered 16-byte-per-lane loads cost the CU's texture path ~60 clocks
// each whatever they hit (an ablation that re-reads one cached chunk is no faster), which is what bounded the
// chunk-at-a-time form of this kernel.
// Operand images are t-innermost as in the bf16 kernel: dY rows are 64 bytes with XOR-swizzled 16-byte groups, x rows
// [cin][32 t] with an 80-byte channel stride (an odd multiple of 16 bytes), so every fragment is one conflict-free
// ds_read_b128 per part.  The kw = 0 / 2 taps use the dY operand shifted by one element against the SAME x fragment
// (dW[kw] = sum_t dY[t] x[t + kw - 1] = sum_u dY[u - kw + 1] x[u]): the shifted forms are built once per step in registers and
// serve all three kernel rows and cin tiles; the two elements a shift pulls in from outside the lane's 8 (dY[8j - 1],
// dY[8j + 8]) sit packed in one dword of a small side array with an odd dword stride per row (read straight from the row
// image they are 8-way bank conflicts: every row starts on the same 4 banks), the row's outer elements come with two extra
// quads per dY row.  (The first form shifted x instead - per kernel row and cin tile: 220 VALU instructions per 222 MFMAs in
// the consumer, and VALU instructions are paid in MFMA time.)
// Why not the Winograd form here: its operands are 6/4 as large per part and are read once per transform point and kernel
// row - at bf16x3 rates that kernel is bound by LDS bytes, this one by the MFMA pipe.
// ============================================================================================
#ifndef WGPC_DBG
#define WGPC_DBG 0          // ablation switches of tools/kernel_ablation.sh (never set in the product build)
#endif
// Consumer waves: WM x WN wave tiles of (MTL x 16 cout) x (NTL x 16 cin), times KS waves that SPLIT THE STEP'S TIME RANGE (wave
// ks contracts over t = 32 ks .. + 31 of a 32 KS wide step; the partial sums meet in the final LDS reduction): the layers with
// 16 / 32 channels have one or two MFMA tiles per side, and four waves only find work along t.  WM * WN * KS = 4.
template <int WM, int WN, int KS = 1, int MTL = 2, int NTL = 2, bool GT = false>
struct WgradPcCfg {
    static_assert(WM * WN * KS == 4, "four consumer waves");
    static constexpr bool COLUMNS = true;                                // launch_wgrad_cfg: the split is over (clip, column) units
    static constexpr int FT = 1, TT = 32 * KS, KK = 9, NT = 512, NJ = TT / 8;
    static constexpr int COUT_T = WM * MTL * 16, CIN_T = WN * NTL * 16;
    // bytes: channel stride of an x row slot (TT t, then padded to an odd multiple of 16 bytes: the 16 rows of a fragment read
    // cover all banks)
    static constexpr int XCH = (TT * 2 / 16) % 2 ? TT * 2 : TT * 2 + 16;
    static constexpr int DY_PART = KS * COUT_T * 64, X_PART = CIN_T * XCH;   // bytes per operand part; dY: [ks][cout][32 t]
    static constexpr int YB_CH = (NJ + 1) * 4, YB_PART = COUT_T * YB_CH; // boundary words: per cout row NJ dwords {dY[8j - 1], dY[8j + 8]} (+ 1 pad:
    static constexpr int DY_STAGE = 3 * (DY_PART + YB_PART), X_SLOT = 3 * X_PART;   // odd dword stride = conflict-free 4-byte reads)
    static constexpr int X_BASE = 2 * DY_STAGE;                         // LDS: dY stage 0, dY stage 1, x slots 0..3, zero slot
    // GT (the BN-backward loader, BNG kernels): two fp32 tiles [cout][TT t] of the formed dY on their way to global memory
    static constexpr int G_BASE = X_BASE + 5 * X_SLOT, G_STAGE = COUT_T * TT * 4;
    static constexpr int LDS_MAIN = G_BASE + (GT ? 2 * G_STAGE : 0);
    static constexpr int XQ = TT / 4, YQ = TT / 4 + 2;                   // 4-element quads per x row / per dY row (one more on either side)
    static constexpr int DY_ITEMS = COUT_T * YQ, X_ITEMS = CIN_T * XQ;
    static constexpr int DY_PER_T = (DY_ITEMS + 255) / 256, X_PER_T = (X_ITEMS + 255) / 256;
    // steps whose global loads are in flight: the steps of the few-channel configurations are short (a few hundred clocks), two
    // of them do not cover the memory latency
    static constexpr int NB = (KS > 1) ? 4 : 2;
    static constexpr int OUT_ROW = CIN_T * KK + 1, OUT_ROWS = COUT_T < 64 ? COUT_T : 64;
    static constexpr int LDS_FLOATS = cmax((LDS_MAIN + 3) / 4, OUT_ROWS * OUT_ROW);
    static_assert(YB_CH / 4 % 2 == 1 && (XCH / 16) % 2 == 1, "odd strides");
};

// BNG: the dY loader applies the BN backward of the next layer's input norm (ConvWgradArgs::gx / gcoef / gseq / gout): one more
// 16-byte load and three coefficient loads per dY quad, two FMAs per element, and - in the blocks of the first cin tile - a
// 16-byte store of dY for the layer's data gradient; the stand-alone elementwise pass over the tensor disappears.
// (BNG = 1: coefficients per channel, per-thread constants; 2: per (channel, output row), loaded with every step.)
template <int WM, int WN, int KS = 1, int MTL = 2, int NTL = 2, int BNG = 0>
__global__ __launch_bounds__(512) void conv_wgrad_pc_kernel(ConvWgradArgs a) {
    using C = WgradPcCfg<WM, WN, KS, MTL, NTL, BNG != 0>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned char* lds = reinterpret_cast<unsigned char*>(smem);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave < 4;
    const int lq = lane >> 4, lr = lane & 15;
    const int cin0 = blockIdx.y * C::CIN_T, cout0 = blockIdx.z * C::COUT_T;
    const int nTt = (a.T + C::TT - 1) / C::TT;
    const int nCols = a.B * nTt;                                         // columns = (clip, TT-wide t range); the launcher's chunks = columns x rows
    const bool pro = a.scale != nullptr;
    const bool unpool = a.unpool_idx != nullptr;
    const int Fg = unpool ? a.F / 2 : a.F;
    // consumer wave -> (cout group, cin group, 32-t slice of the step)
    const int wks = wave % KS, wni = (wave / KS) % WN, wmi = wave / (KS * WN);
    const bool do_bias = (a.db != nullptr) && (blockIdx.y == 0) && wni == 0;
    // columns of this block: bx, + gridDim.x, ..; its steps run through them row by row.  bx = blockIdx.x, or (xcd_cols) the
    // XCD-contiguous position (x % 8) * (gridDim.x / 8) + x / 8: a permutation of 0 .. gridDim.x - 1 (gridDim.x % 8 == 0)
    const int bx = a.xcd_cols ? (int)(blockIdx.x & 7u) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    int nColMine = 0;
    if (bx < nCols) nColMine = (nCols - 1 - bx) / (int)gridDim.x + 1;
    const int nSteps = nColMine * a.F;
    constexpr int NBU = C::NB;
    const int nStepsR = (nSteps + NBU - 1) / NBU * NBU;                  // the producers' loop body covers NB steps without conditions

    f32x4 acc[MTL][NTL][9], accb[MTL];
#pragma unroll
    for (int m = 0; m < MTL; ++m) {
        accb[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < NTL; ++n)
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[m][n][k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // the all-zero x slot (rows above / below the plane)
    for (int i = tid; i < C::X_SLOT / 16; i += 512)
        reinterpret_cast<u32x4_t*>(lds + C::X_BASE + 4 * C::X_SLOT)[i] = u32x4_t{0u, 0u, 0u, 0u};

    if (!consumer && nSteps > 0) {
        // ================================================================ PRODUCER
        const int pt = tid - 256;
        constexpr unsigned OOB = 0x20000000u;            // element offset beyond every clip (x 4 = 2^31 bytes)
        const unsigned gclip = (unsigned)(a.Cout * Fg * a.T), xclip = (unsigned)(a.Cin * a.F * a.T);
        const __amdgpu_buffer_rsrc_t rs_sc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.scale), 0, pro ? (unsigned)a.Cin * 4u : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_sh = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.shift), 0, pro ? (unsigned)a.Cin * 4u : 0u, 0x00020000);
        // per-thread item constants (computed once: the step loop only adds a uniform offset and selects, no divisions and no
        // divergent branches): dY item i = (cout row, quad -1 .. TT / 4 of the step's range: the outer two only feed the
        // boundary words), x item i = (cin, quad of XQ)
        int y_q[C::DY_PER_T], y_valid[C::DY_PER_T], y_item[C::DY_PER_T];
        unsigned y_lds[C::DY_PER_T], y_bnd[C::DY_PER_T];
        int y_base[C::DY_PER_T];
#pragma unroll
        for (int i = 0; i < C::DY_PER_T; ++i) {
            const int it = pt + i * 256, cl = it / C::YQ;
            y_q[i] = it % C::YQ - 1;
            const int qq = y_q[i] & (C::TT / 4 - 1);                             // (the outer quads have no place in the tile)
            const int row = cl & 15, q8 = qq & 7, ks = qq >> 3;                  // 32-t slice ks, quad q8 of its 8
            y_lds[i] = (unsigned)((ks * C::COUT_T + cl) * 64 + ((((q8 >> 1) ^ ((-(row >> 2)) & 3)) & 3) * 16) + (q8 & 1) * 8);
            y_bnd[i] = (unsigned)(cl * C::YB_CH);
            y_item[i] = it < C::DY_ITEMS;
            y_valid[i] = y_item[i] & (cout0 + cl < a.Cout);
            y_base[i] = (cout0 + cl) * Fg * a.T + 4 * y_q[i];
        }
        // BNG: coefficient row of the item's channel and whether this thread writes the item's dY (inner quads, first cin tile)
        const int cfS = BNG == 2 ? Fg : 1;
        const unsigned nCoef = (unsigned)(a.Cout * cfS);
        const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.gcoef), 0, BNG ? 3u * nCoef * 4u : 0u, 0x00020000);
        unsigned y_c[BNG ? C::DY_PER_T : 1];
        int y_wr[BNG ? C::DY_PER_T : 1];
        unsigned y_g[BNG ? C::DY_PER_T : 1];
        float y_k[BNG == 1 ? C::DY_PER_T : 1][3];
        if constexpr (BNG != 0) {
#pragma unroll
            for (int i = 0; i < C::DY_PER_T; ++i) {
                const int it = pt + i * 256, cl = it / C::YQ;
                y_c[i] = (unsigned)((cout0 + cl) * cfS);
                y_wr[i] = y_item[i] & (y_q[i] >= 0) & (y_q[i] < C::TT / 4);      // an inner quad: has a place in the fp32 tile
                y_g[i] = (unsigned)(cl * C::TT * 4 + (y_q[i] & (C::TT / 4 - 1)) * 16);
                if constexpr (BNG == 1) {
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        y_k[i][k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_k, y_valid[i] ? y_c[i] * 4u : 0x80000000u, k * nCoef * 4u, 0));
                }
            }
        }
        int x_q[C::X_PER_T], x_valid[C::X_PER_T], x_item[C::X_PER_T];
        unsigned x_lds[C::X_PER_T];
        int x_base[C::X_PER_T];
        float x_sc[C::X_PER_T], x_sh[C::X_PER_T];
#pragma unroll
        for (int i = 0; i < C::X_PER_T; ++i) {
            const int it = pt + i * 256, cl = it / C::XQ;
            x_q[i] = it % C::XQ;
            x_lds[i] = (unsigned)(cl * C::XCH + x_q[i] * 8);
            x_item[i] = it < C::X_ITEMS;
            x_valid[i] = x_item[i] & (cin0 + cl < a.Cin);
            x_base[i] = (cin0 + cl) * a.F * a.T + 4 * x_q[i];
            const bool ok = x_valid[i] && pro;
            x_sc[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_sc, ok ? (unsigned)(cin0 + cl) * 4u : 0x80000000u, 0, 0));
            x_sh[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_sh, ok ? (unsigned)(cin0 + cl) * 4u : 0x80000000u, 0, 0));
            if (!pro) x_sc[i] = 1.f;
        }
        const float relu_floor = (pro && a.relu) ? 0.f : -__builtin_inff();
        constexpr int NB = C::NB;                          // raw register sets = steps whose loads are in flight
        u32x4_t ry[NB][C::DY_PER_T], rx[NB][C::X_PER_T];
        unsigned ryi[NB][C::DY_PER_T];
        int ry_n[NB][C::DY_PER_T], rx_n[NB][C::X_PER_T], r_par[NB];
        u32x4_t rgx[BNG ? NB : 1][C::DY_PER_T];                 // BNG: the raw conv output under the norm, the item's three
        unsigned rk[BNG == 2 ? NB : 1][C::DY_PER_T][3];         // coefficients (per step when they depend on the row)

        // step S of the block = (column S / F of its list, row S % F).  Raw set S % 2 holds what is staged DURING step S - 1 ..
        // the dY tile of step S and the x row of step S + 1 (x row g lives in ring slot g % 4; rows 0 and 1 of the stream are
        // staged in the prologue): load_pair(S) issues the loads of {dY of step S, x row of step S + 1}.
        auto locate = [&](int S, int& b, int& f, int& t0) __attribute__((always_inline)) {
            const int col = bx + (S / a.F) * (int)gridDim.x;
            f = S % a.F; b = col / nTt; t0 = (col % nTt) * C::TT;
        };
        auto load_dy = [&](int S, auto buf_c) __attribute__((always_inline)) {
            constexpr int BUF = decltype(buf_c)::value;
            int b, f, t0;
            locate(S, b, f, t0);
            const int live = S < nSteps;                   // steps past the block's last one: every offset out of range, zeros staged
            const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(a.g) + (size_t)b * gclip, 0, gclip * 4u, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc(
                unpool ? const_cast<uint8_t*>(a.unpool_idx) + (size_t)b * gclip : nullptr, 0, unpool ? gclip : 0u, 0x00020000);
            r_par[BUF] = f & 1;
            const int fg = unpool ? (f >> 1) : f;
            const unsigned y_row = (unsigned)(fg * a.T + t0);
            int tlim_g = a.T;
            __amdgpu_buffer_rsrc_t rs_gx = rs_g;
            if constexpr (BNG != 0) {
                const int bc = min(b, a.B - 1);
                tlim_g = a.gseq ? min(a.gseq[bc], a.T) : a.T;
                rs_gx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.gx) + (size_t)b * gclip, 0, gclip * 4u, 0x00020000);
            }
#pragma unroll
            for (int i = 0; i < C::DY_PER_T; ++i) {
                const int tq = t0 + 4 * y_q[i];
                // branch-free selects (as ?: the compiler forks into two load sites that must wait for each other)
                const unsigned ok = (unsigned)-(y_valid[i] & (tq >= 0) & (tq < a.T) & live);
                const unsigned off = ((unsigned)(y_base[i] + (int)y_row) & ok) | (OOB & ~ok);
                ry[BUF][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_g, off * 4u, 0, 0);
                if (unpool) ryi[BUF][i] = __builtin_amdgcn_raw_buffer_load_b32(rs_i, off, 0, 0);
                if constexpr (BNG != 0) {
                    rgx[BUF][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_gx, off * 4u, 0, 0);
                    if constexpr (BNG == 2) {
                        const unsigned coff = (((y_c[i] + (unsigned)fg) * 4u) & ok) | (0x80000000u & ~ok);
#pragma unroll
                        for (int k = 0; k < 3; ++k) rk[BUF][i][k] = __builtin_amdgcn_raw_buffer_load_b32(rs_k, coff, k * nCoef * 4u, 0);
                    }
                    ry_n[BUF][i] = (int)((unsigned)min(max(tlim_g - tq, 0), 4) & ok);
                } else {
                    ry_n[BUF][i] = (int)((unsigned)min(a.T - tq, 4) & ok);
                }
            }
        };
        auto load_x = [&](int S, auto buf_c) __attribute__((always_inline)) {       // the x row of step S (its centre row f)
            constexpr int BUF = decltype(buf_c)::value;
            int b, f, t0;
            locate(S, b, f, t0);
            const int live = S < nSteps;
            b = min(b, a.B - 1);
            const int sl = a.seq_len ? min(a.seq_len[b], a.T) : a.T;
            const int tlim = pro ? sl : a.T;
            const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(a.x) + (size_t)b * xclip, 0, xclip * 4u, 0x00020000);
            const int x_row = f * a.T + t0;
#pragma unroll
            for (int i = 0; i < C::X_PER_T; ++i) {
                const int tq = t0 + 4 * x_q[i];
                const unsigned ok = (unsigned)-(x_valid[i] & (tq < a.T) & live);
                const unsigned off = ((unsigned)(x_base[i] + x_row) & ok) | (OOB & ~ok);
                rx[BUF][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, off * 4u, 0, 0);
                rx_n[BUF][i] = (int)((unsigned)min(max(tlim - tq, 0), 4) & ok);
            }
        };
        auto put = [&](unsigned char* p, int part_bytes, const float (&v)[4]) __attribute__((always_inline)) {
            unsigned h0, m0, l0, h1, m1, l1;
            split3_pair(v[0], v[1], h0, m0, l0);
            split3_pair(v[2], v[3], h1, m1, l1);
            *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(p + part_bytes) = make_uint2(m0, m1);
            *reinterpret_cast<uint2*>(p + 2 * part_bytes) = make_uint2(l0, l1);
        };
        auto store_dy = [&](int stage, auto buf_c) __attribute__((always_inline)) {
            constexpr int BUF = decltype(buf_c)::value;
            unsigned char* dy_s = lds + stage * C::DY_STAGE;
            unsigned char* g_s = lds + C::G_BASE + stage * C::G_STAGE;
#pragma unroll
            for (int i = 0; i < C::DY_PER_T; ++i) {
                if (y_item[i]) {
                    const u32x4_t r = ry[BUF][i];
                    float v[4] = {__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w)};
                    if constexpr (BNG != 0) {
                        const u32x4_t gr = rgx[BUF][i];
                        const float gxv[4] = {__uint_as_float(gr.x), __uint_as_float(gr.y), __uint_as_float(gr.z), __uint_as_float(gr.w)};
                        constexpr int KB = BNG == 2 ? BUF : 0, KI = BNG == 1 ? 1 : 0;
                        const float k1 = BNG == 2 ? __uint_as_float(rk[KB][i][0]) : y_k[i * KI][0], k2 = BNG == 2 ? __uint_as_float(rk[KB][i][1]) : y_k[i * KI][1],
                                    k3 = BNG == 2 ? __uint_as_float(rk[KB][i][2]) : y_k[i * KI][2];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = e < ry_n[BUF][i] ? fmaf(v[e], k1, fmaf(gxv[e], k2, k3)) : 0.f;
                        const u32x4_t q4 = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
#if !(WGPC_DBG & 64)
                        // dY leaves through LDS: the CONSUMER waves copy the step's fp32 tile to global memory.  A buffer store in
                        // this (loading) wave turns every later wait for a load into s_waitcnt vmcnt(0) - on gfx9 a wave's loads and
                        // stores count on ONE vmcnt, in issue order: a load behind a store waits for the store - and the two-step
                        // load prefetch collapses (measured: +30 % on the launch, also with the store hidden in inline assembly).
                        if (y_wr[i]) *reinterpret_cast<u32x4_t*>(g_s + y_g[i]) = q4;
#endif
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        bool keep = e < ry_n[BUF][i];
                        if (unpool) keep = keep && (int)((ryi[BUF][i] >> (8 * e)) & 0xffu) == r_par[BUF];
                        v[e] = keep ? v[e] : 0.f;
                    }
                    const int q = y_q[i];
                    if (q >= 0 && q < C::TT / 4) put(dy_s + y_lds[i], C::DY_PART, v);
                    // boundary words of the row's 8-element groups: quad 2 j - 1 ends with dY[8 j - 1] (low half of word j), quad
                    // 2 j + 2 starts with dY[8 j + 8] (high half of word j)
                    const bool lo_w = (q & 1) && q <= 2 * C::NJ - 3, hi_w = !(q & 1) && q >= 2;
                    if (lo_w || hi_w) {
                        const float bv = lo_w ? v[3] : v[0];
                        const int j = lo_w ? (q + 1) >> 1 : (q - 2) >> 1;
                        const unsigned u0 = __float_as_uint(bv);
                        const float r1 = bv - __uint_as_float(u0 & 0xffff0000u);
                        const unsigned u1 = __float_as_uint(r1);
                        const float r2 = r1 - __uint_as_float(u1 & 0xffff0000u);
                        unsigned char* pb = dy_s + 3 * C::DY_PART + y_bnd[i] + j * 4 + (hi_w ? 2 : 0);
                        *reinterpret_cast<unsigned short*>(pb) = (unsigned short)(u0 >> 16);
                        *reinterpret_cast<unsigned short*>(pb + C::YB_PART) = (unsigned short)(u1 >> 16);
                        *reinterpret_cast<unsigned short*>(pb + 2 * C::YB_PART) = (unsigned short)(__float_as_uint(r2) >> 16);
                    }
                }
            }
        };
        auto store_x = [&](int slot, auto buf_c) __attribute__((always_inline)) {
            constexpr int BUF = decltype(buf_c)::value;
            unsigned char* x_s = lds + C::X_BASE + slot * C::X_SLOT;
#pragma unroll
            for (int i = 0; i < C::X_PER_T; ++i) {
                if (x_item[i]) {
                    const u32x4_t r = rx[BUF][i];
                    float v[4] = {__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w)};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float u = fmaxf(fmaf(v[e], x_sc[i], x_sh[i]), relu_floor);      // no prologue: x * 1 + 0, floor -inf
                        v[e] = e < rx_n[BUF][i] ? u : 0.f;               // zero padding is post-activation
                    }
                    put(x_s + x_lds[i], C::X_PART, v);
                }
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2 % NB>;
        using I3 = std::integral_constant<int, 3 % NB>;
        constexpr bool P_LD = !(WGPC_DBG & 1), P_ST = !(WGPC_DBG & 2);
        // Raw set k % NB holds the pair {dY of step k, x row k + 1}.  Prologue: x rows 0 and 1 and dY of step 0 staged, the pairs
        // of steps 1 .. NB in flight.
        // No conditions between here and the end of the loop (loads of steps past the last one are masked out of range and
        // stage zeros into images nobody reads): where paths with different numbers of outstanding loads meet, the compiler's
        // wait counts assume the fewest and a step's loads are waited for right after they were requested - the prefetch
        // distance collapses to one step.  The scheduling barriers keep the load / convert groups in program order, so the queue
        // of outstanding loads looks the same on every path into the loop.
        if (P_LD) { load_dy(0, I0{}); load_x(0, I0{}); load_x(1, I1{}); }
        __builtin_amdgcn_sched_barrier(0);
        if (P_ST) { store_dy(0, I0{}); store_x(0, I0{}); store_x(1, I1{}); }
        __builtin_amdgcn_sched_barrier(0);
        auto load_pair = [&](int k, auto buf_c) __attribute__((always_inline)) {
            if (P_LD) { load_dy(k, buf_c); load_x(k + 1, buf_c); }
            __builtin_amdgcn_sched_barrier(0);
        };
        load_pair(1, I1{});
        if (NB == 2) {
            load_pair(2, I0{});
        } else {
            load_pair(2, I2{}); load_pair(3, I3{}); load_pair(4, I0{});
        }
        __syncthreads();
        // during step S (consumers: dY stage S % 2, x rows S - 1, S, S + 1): stage the pair of step S + 1 (dY of step S + 1, x row
        // S + 2), then re-load its raw set with the pair of step S + 1 + NB
        auto stage_step = [&](int S, auto buf_c) __attribute__((always_inline)) {
            using BUF = decltype(buf_c);
            if (P_ST) { store_dy((S + 1) & 1, BUF{}); store_x((S + 2) & 3, BUF{}); }
            __builtin_amdgcn_sched_barrier(0);
            load_pair(S + 1 + NB, BUF{});
            __syncthreads();
        };
        if (NB == 2) {
            for (int S = 0; S < nStepsR; S += 2) {
                stage_step(S, I1{});
                stage_step(S + 1, I0{});
            }
        } else {
            for (int S = 0; S < nStepsR; S += 4) {
                stage_step(S, I1{});
                stage_step(S + 1, I2{});
                stage_step(S + 2, I3{});
                stage_step(S + 3, I0{});
            }
        }
    } else if (nSteps > 0) {
        // ================================================================ CONSUMER
        const u32x4_t ones = u32x4_t{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
        // dY fragment: slice wks, row lr, t group lq, and its boundary word;  x fragment: cin lr, t = 32 wks + 8 lq .. + 7
        const unsigned a_lane = (unsigned)(wks * C::COUT_T * 64 + lr * 64 + (((lq ^ ((-(lr >> 2)) & 3)) & 3) * 16));
        const unsigned bnd_lane = (unsigned)(lr * C::YB_CH + (wks * 4 + lq) * 4);
        const unsigned b_lane = (unsigned)(lr * C::XCH + (wks * 4 + lq) * 16);
        const bool g_writer = BNG != 0 && blockIdx.y == 0 && a.gout != nullptr;     // the first cin tile's blocks write dY out
        auto step = [&](int S) __attribute__((always_inline)) {
            const int f = S % a.F;
            const unsigned char* dy_s = lds + (S & 1) * C::DY_STAGE;
            if constexpr (BNG != 0) {
                // the fp32 tile of the formed dY of this step (staged by the producers) -> global memory; a pooled row is staged
                // in two steps (both parities), written in the even one
                if (g_writer && (!unpool || !(f & 1))) {
                    const int col = bx + (S / a.F) * (int)gridDim.x;
                    const int b = col / nTt, t0 = (col % nTt) * C::TT, fg = unpool ? (f >> 1) : f;
                    const unsigned gclip = (unsigned)(a.Cout * Fg * a.T);
                    const __amdgpu_buffer_rsrc_t rs_go = __builtin_amdgcn_make_buffer_rsrc(a.gout + (size_t)b * gclip, 0, gclip * 4u, 0x00020000);
                    const unsigned char* g_s = lds + C::G_BASE + (S & 1) * C::G_STAGE;
#pragma unroll
                    for (int k = 0; k < C::COUT_T * C::TT / 4 / 256; ++k) {
                        const int ch = wave * (C::COUT_T * C::TT / 4 / 4) + k * 64 + lane;        // 16-byte chunk of the tile
                        const int row = ch / (C::TT / 4), tq = t0 + 4 * (ch % (C::TT / 4));
                        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(g_s + ch * 16);
                        const bool ok = cout0 + row < a.Cout && tq < a.T;
                        const unsigned off = ok ? (unsigned)(((cout0 + row) * Fg + fg) * a.T + tq) * 4u : 0x80000000u;
                        __builtin_amdgcn_raw_buffer_store_b128(v, rs_go, off, 0, 0);
                    }
                }
            }
            // x rows f - 1, f, f + 1 of this column: ring slots (S - 1, S, S + 1) % 4, the zero slot outside the plane
            unsigned xoff[3];
            xoff[0] = (unsigned)(C::X_BASE + (f > 0 ? ((S - 1) & 3) : 4) * C::X_SLOT);
            xoff[1] = (unsigned)(C::X_BASE + (S & 3) * C::X_SLOT);
            xoff[2] = (unsigned)(C::X_BASE + (f + 1 < a.F ? ((S + 1) & 3) : 4) * C::X_SLOT);
            // dW[kw] = sum_t dY[t] x[t + kw - 1] = sum_u dY[u - kw + 1] x[u]: the kw = 0 / 2 taps take the dY operand shifted by one
            // element (built ONCE per step in registers from the centre fragment and its boundary word, shared by the three kernel
            // rows and the cin tiles) against the SAME x fragment - the x fragments need no arithmetic at all.  (Shifting x, as
            // the first form of this kernel did, costs the shifts per (kernel row, cin tile): 220 VALU instructions per 222 MFMAs.)
            u32x4_t af[MTL][3], aL[MTL][3], aR[MTL][3];             // dY[t], dY[t - 1] (tap kw = 2), dY[t + 1] (tap kw = 0)
#pragma unroll
            for (int m = 0; m < MTL; ++m)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const u32x4_t c = *reinterpret_cast<const u32x4_t*>(dy_s + p * C::DY_PART + ((wmi * MTL + m) * 16) * 64 + a_lane);
                    // {dY[t - 1] (low half), dY[t + 8] (high half)} of this lane's 8-element group
                    const unsigned w = *reinterpret_cast<const unsigned*>(dy_s + 3 * C::DY_PART + p * C::YB_PART +
                                                                             ((wmi * MTL + m) * 16) * C::YB_CH + bnd_lane);
                    const unsigned s1 = __builtin_amdgcn_alignbit(c.y, c.x, 16), s2 = __builtin_amdgcn_alignbit(c.z, c.y, 16),
                                   s3 = __builtin_amdgcn_alignbit(c.w, c.z, 16);
                    af[m][p] = c;
                    aL[m][p] = u32x4_t{__builtin_amdgcn_perm(c.x, w, 0x05040100u), s1, s2, s3};     // dY[t-1 .. t+6]
                    aR[m][p] = u32x4_t{s1, s2, s3, __builtin_amdgcn_perm(w, c.w, 0x07060302u)};    // dY[t+1 .. t+8]
                }
            // x fragments of the (kernel row, cin tile) pairs one pair ahead of the MFMAs that use them
            u32x4_t cx[2][3];
            auto read_x = [&](int j, u32x4_t (&cr)[3]) __attribute__((always_inline)) {
                const int kh = j / NTL, n = j % NTL;
                const unsigned char* x_s = lds + xoff[kh];
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    cr[p] = *reinterpret_cast<const u32x4_t*>(x_s + p * C::X_PART + ((wni * NTL + n) * 16) * C::XCH + b_lane);
            };
            read_x(0, cx[0]);
#pragma unroll
            for (int j = 0; j < 3 * NTL; ++j) {
                const int kh = j / NTL, n = j % NTL, cur = j & 1;
                if (j + 1 < 3 * NTL) read_x(j + 1, cx[cur ^ 1]);
                // six part products of the (m, kw) accumulators round-robin, smallest first
#pragma unroll
                for (int pp = 0; pp < 6; ++pp) {
                    const int pa = pp == 0 ? 2 : pp == 1 ? 0 : pp == 2 ? 1 : pp == 3 ? 1 : 0;     // lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
                    const int pb = pp == 0 ? 0 : pp == 1 ? 2 : pp == 2 ? 1 : pp == 3 ? 0 : pp == 4 ? 1 : 0;
#pragma unroll
                    for (int m = 0; m < MTL; ++m) {
                        acc[m][n][kh * 3 + 0] = wg_mfma(aR[m][pa], cx[cur][pb], acc[m][n][kh * 3 + 0]);
                        acc[m][n][kh * 3 + 1] = wg_mfma(af[m][pa], cx[cur][pb], acc[m][n][kh * 3 + 1]);
                        acc[m][n][kh * 3 + 2] = wg_mfma(aL[m][pa], cx[cur][pb], acc[m][n][kh * 3 + 2]);
                    }
                }
            }
            if (do_bias) {
#pragma unroll
                for (int m = 0; m < MTL; ++m)
#pragma unroll
                    for (int p = 0; p < 3; ++p) accb[m] = wg_mfma(af[m][2 - p], ones, accb[m]);
            }
        };
        __syncthreads();                                                 // the prologue's images are in place
        for (int S = 0; S < nStepsR; ++S) {
            if (!(WGPC_DBG & 8) && S < nSteps) step(S);
            __syncthreads();
        }
    }

    // ---- reduce: transpose the block's partial dW through LDS (the KS time slices add up there, one after the other), then
    // row-contiguous atomics (as conv_wgrad_bf16_kernel)
    __syncthreads();
    float* out_s = smem;                             // [OUT_ROWS][OUT_ROW]
    const int ncol = min(C::CIN_T, a.Cin - cin0) * 9;
    const int slot = a.nslots > 1 ? (int)(blockIdx.x % a.nslots) : 0;
    float* dwp = a.dw + (size_t)slot * a.slot_w;
    constexpr int WROWS = MTL * 16;                  // cout rows of a consumer wave
#pragma unroll 1
    for (int part = 0; part < (C::COUT_T + C::OUT_ROWS - 1) / C::OUT_ROWS; ++part) {
#pragma unroll 1
        for (int ks = 0; ks < KS; ++ks) {
            if (part > 0 || ks > 0) __syncthreads();
            if (consumer && wks == ks && (wmi * WROWS) / C::OUT_ROWS == part) {
#pragma unroll
                for (int m = 0; m < MTL; ++m)
#pragma unroll
                    for (int n = 0; n < NTL; ++n)
#pragma unroll
                        for (int kk = 0; kk < 9; ++kk)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                float* o = &out_s[((wmi * WROWS) % C::OUT_ROWS + m * 16 + lq * 4 + r) * C::OUT_ROW + ((wni * NTL + n) * 16 + lr) * 9 + kk];
                                *o = ks == 0 ? acc[m][n][kk][r] : *o + acc[m][n][kk][r];
                            }
            }
        }
        __syncthreads();
        for (int row = wave; row < C::OUT_ROWS; row += 8) {
            const int cout = cout0 + part * C::OUT_ROWS + row;
            if (cout >= a.Cout || part * C::OUT_ROWS + row >= C::COUT_T) break;
            float* dst = dwp + ((size_t)cout * a.Cin + cin0) * 9;
            for (int col = lane; col < ncol; col += 64) atomicAdd(dst + col, out_s[row * C::OUT_ROW + col]);
        }
    }
    if (consumer && do_bias && lr == 0) {
#pragma unroll
        for (int m = 0; m < MTL; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cout = cout0 + (wmi * MTL + m) * 16 + lq * 4 + r;
                if (cout < a.Cout) atomicAdd(&a.db[(size_t)slot * a.slot_b + cout], accb[m][r]);
            }
    }
}

// ============================================================================================
// Conv1d (F = 1 rows; kernel sizes 1 and 3) weight gradient of the fp32 path, PRODUCER / CONSUMER form of the bf16x3 kernel:
// conv_wgrad_pc_kernel without the column walk - a step is a (clip, 32-t range) unit, every operand element is staged once
// per step (the 3-tap kernel stages x[t0 - 8 .. t0 + 40) and builds the kw = 0 / 2 operands by shifting the centre one in
// registers, boundary elements from the side array, as there).  What makes these layers different is the ratio of staged
// elements to products: without kernel rows sharing an x row a step has 1 / 3 of the MFMAs per staged element, so the blocks
// are as large as the accumulators allow: consumer wave = 64 cout x 32 cin x 3 taps (k = 3; block 128 x 64) or
// 64 cout x 64 cin (k = 1; block 128 x 128).  The producers' VALU instructions and the consumers' MFMAs share the SIMDs' issue
// cycles (conv_winox3.hip), which is what bounds this kernel.
// ============================================================================================
template <int KW>
struct Wgrad1dPcCfg {
    static constexpr bool COLUMNS = true;
    static constexpr int FT = 1, TT = 32, KK = KW, NT = 512;
    static constexpr int MTL = 4, NTL = KW == 3 ? 2 : 4;                 // 16-row tiles per consumer wave: cout, cin
    static constexpr int COUT_T = 2 * MTL * 16, CIN_T = 2 * NTL * 16;    // 2 x 2 consumer waves
    static constexpr int XQ = KW == 3 ? 12 : 8;                          // 4-element quads staged per x row
    static constexpr int XCH = KW == 3 ? 112 : 64;                       // bytes per x row (k = 3: 48 t + 16, 16 bytes mod 128)
    static constexpr int DY_PART = COUT_T * 64, X_PART = CIN_T * XCH;
    static constexpr int XB_CH = 20, XB_PART = KW == 3 ? CIN_T * XB_CH : 0;
    static constexpr int DY_STAGE = 3 * DY_PART, X_STAGE = 3 * (X_PART + XB_PART);
    static constexpr int X_BASE = 2 * DY_STAGE;
    static constexpr int LDS_MAIN = X_BASE + 2 * X_STAGE;
    static constexpr int DY_ITEMS = COUT_T * 8, X_ITEMS = CIN_T * XQ;
    static constexpr int DY_PER_T = DY_ITEMS / 256, X_PER_T = X_ITEMS / 256;
    static constexpr int OUT_ROW = CIN_T * KK + 1, OUT_ROWS = 64;
    static constexpr int LDS_FLOATS = cmax((LDS_MAIN + 3) / 4, OUT_ROWS * OUT_ROW);
    static_assert(DY_ITEMS % 256 == 0 && X_ITEMS % 256 == 0, "whole items per producer thread");
};

template <int KW>
__global__ __launch_bounds__(512) void conv1d_wgrad_pc_kernel(ConvWgradArgs a) {
    using C = Wgrad1dPcCfg<KW>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned char* lds = reinterpret_cast<unsigned char*>(smem);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave < 4;
    const int lq = lane >> 4, lr = lane & 15;
    const int cin0 = blockIdx.y * C::CIN_T, cout0 = blockIdx.z * C::COUT_T;
    const int nTt = (a.T + C::TT - 1) / C::TT;
    const int nUnits = a.B * a.F * nTt;                                  // steps = (clip, row, 32-t range) units (rows: 1x1 conv2d layers)
    const bool pro = a.scale != nullptr;
    const int wmi = wave >> 1, wni = wave & 1;                           // consumer wave -> (64-cout group, cin group)
    const bool do_bias = (a.db != nullptr) && (blockIdx.y == 0) && wni == 0;
    int nSteps = 0;
    if ((int)blockIdx.x < nUnits) nSteps = (nUnits - 1 - (int)blockIdx.x) / (int)gridDim.x + 1;

    f32x4 acc[C::MTL][C::NTL][KW], accb[C::MTL];
#pragma unroll
    for (int m = 0; m < C::MTL; ++m) {
        accb[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < C::NTL; ++n)
#pragma unroll
            for (int k = 0; k < KW; ++k) acc[m][n][k] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    if (!consumer && nSteps > 0) {
        // ================================================================ PRODUCER
        const int pt = tid - 256;
        constexpr unsigned OOB = 0x20000000u;            // element offset beyond every clip (x 4 = 2^31 bytes)
        const int plane = a.F * a.T;                       // elements per channel of a clip
        const unsigned gclip = (unsigned)(a.Cout * plane), xclip = (unsigned)(a.Cin * plane);
        const __amdgpu_buffer_rsrc_t rs_sc = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.scale), 0, pro ? (unsigned)a.Cin * 4u : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_sh = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(a.shift), 0, pro ? (unsigned)a.Cin * 4u : 0u, 0x00020000);
        // per-thread item constants: dY item = (cout row, quad of 8), x item = (cin row, quad of XQ)
        int y_q[C::DY_PER_T], y_valid[C::DY_PER_T];
        unsigned y_lds[C::DY_PER_T], y_base[C::DY_PER_T];
#pragma unroll
        for (int i = 0; i < C::DY_PER_T; ++i) {
            const int it = pt + i * 256, cl = it >> 3;
            y_q[i] = it & 7;
            const int row = cl & 15;
            y_lds[i] = (unsigned)(cl * 64 + ((((y_q[i] >> 1) ^ ((-(row >> 2)) & 3)) & 3) * 16) + (y_q[i] & 1) * 8);
            y_valid[i] = cout0 + cl < a.Cout;
            y_base[i] = (unsigned)((cout0 + cl) * plane + 4 * y_q[i]);
        }
        constexpr int XOFF = KW == 3 ? 8 : 0;               // the x window starts XOFF elements before the step's range
        int x_q[C::X_PER_T], x_valid[C::X_PER_T];
        unsigned x_lds[C::X_PER_T], x_bnd[C::X_PER_T];
        int x_base[C::X_PER_T];
        float x_sc[C::X_PER_T], x_sh[C::X_PER_T];
#pragma unroll
        for (int i = 0; i < C::X_PER_T; ++i) {
            const int it = pt + i * 256, cl = it / C::XQ;
            x_q[i] = it % C::XQ;
            if (KW == 3) {
                x_lds[i] = (unsigned)(cl * C::XCH + x_q[i] * 8);
            } else {
                const int row = cl & 15;
                x_lds[i] = (unsigned)(cl * 64 + ((((x_q[i] >> 1) ^ ((-(row >> 2)) & 3)) & 3) * 16) + (x_q[i] & 1) * 8);
            }
            x_bnd[i] = (unsigned)(cl * C::XB_CH);
            x_valid[i] = cin0 + cl < a.Cin;
            x_base[i] = (cin0 + cl) * plane + 4 * x_q[i] - XOFF;
            const bool ok = x_valid[i] && pro;
            x_sc[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_sc, ok ? (unsigned)(cin0 + cl) * 4u : 0x80000000u, 0, 0));
            x_sh[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_sh, ok ? (unsigned)(cin0 + cl) * 4u : 0x80000000u, 0, 0));
            if (!pro) x_sc[i] = 1.f;
        }
        const float relu_floor = (pro && a.relu) ? 0.f : -__builtin_inff();
        constexpr int NB = 2;                              // raw register sets = steps whose loads are in flight
        u32x4_t ry[NB][C::DY_PER_T], rx[NB][C::X_PER_T];
        int ry_n[NB][C::DY_PER_T], rx_lo[NB][C::X_PER_T], rx_hi[NB][C::X_PER_T];
        const bool vec = (a.T & 3) == 0;                   // rows 16-byte aligned: one load per quad

        auto locate = [&](int S, int& b, int& t0, int& row0) __attribute__((always_inline)) {
            const int u = (int)blockIdx.x + S * (int)gridDim.x;
            const int bf = u / nTt;
            t0 = (u % nTt) * C::TT;
            b = bf / a.F; row0 = (bf % a.F) * a.T;
        };
        auto load_quad = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned e, int lo, int hi) __attribute__((always_inline)) -> u32x4_t {
            // elements lo <= k < hi of the quad at element offset e are inside the row (branch-free selects)
            if (vec) {
                const unsigned ok = (unsigned)-(int)(hi > lo);
                return __builtin_amdgcn_raw_buffer_load_b128(rs, ((e * 4u) & ok) | ((OOB * 4u) & ~ok), 0, 0);
            }
            unsigned w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned ok = (unsigned)-(int)(k >= lo && k < hi);
                w[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, (((e + (unsigned)k) * 4u) & ok) | ((OOB * 4u) & ~ok), 0, 0);
            }
            return u32x4_t{w[0], w[1], w[2], w[3]};
        };
        auto load_step = [&](int S, auto buf_c) __attribute__((always_inline)) {
            constexpr int BUF = decltype(buf_c)::value;
            int b, t0, row0;
            locate(S, b, t0, row0);
            const int sl = a.seq_len ? min(a.seq_len[b], a.T) : a.T;
            const int tlim = pro ? sl : a.T;
            const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(a.g) + (size_t)b * gclip, 0, gclip * 4u, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(a.x) + (size_t)b * xclip, 0, xclip * 4u, 0x00020000);
#pragma unroll
            for (int i = 0; i < C::DY_PER_T; ++i) {
                const int tq = t0 + 4 * y_q[i];
                const int hi = y_valid[i] ? min(max(a.T - tq, 0), 4) : 0;
                ry[BUF][i] = load_quad(rs_g, y_base[i] + (unsigned)(row0 + t0), 0, hi);
                ry_n[BUF][i] = hi;
            }
#pragma unroll
            for (int i = 0; i < C::X_PER_T; ++i) {
                const int tq = t0 - XOFF + 4 * x_q[i];
                const int lo = x_valid[i] ? min(max(-tq, 0), 4) : 4;          // elements before the row's start
                const int hi_row = x_valid[i] ? min(max(a.T - tq, 0), 4) : 0;
                rx[BUF][i] = load_quad(rs_x, (unsigned)(x_base[i] + row0 + t0), vec ? 0 : lo, vec ? (tq >= 0 ? hi_row : 0) : hi_row);
                rx_lo[BUF][i] = lo;
                rx_hi[BUF][i] = x_valid[i] ? min(max(tlim - tq, 0), 4) : 0;   // zero padding is post-activation
            }
        };
        auto put = [&](unsigned char* p, int part_bytes, const float (&v)[4]) __attribute__((always_inline)) {
            unsigned h0, m0, l0, h1, m1, l1;
            split3_pair(v[0], v[1], h0, m0, l0);
            split3_pair(v[2], v[3], h1, m1, l1);
            *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(p + part_bytes) = make_uint2(m0, m1);
            *reinterpret_cast<uint2*>(p + 2 * part_bytes) = make_uint2(l0, l1);
        };
        auto store_step = [&](int stage, auto buf_c) __attribute__((always_inline)) {
            constexpr int BUF = decltype(buf_c)::value;
            unsigned char* dy_s = lds + stage * C::DY_STAGE;
            unsigned char* x_s = lds + C::X_BASE + stage * C::X_STAGE;
#pragma unroll
            for (int i = 0; i < C::DY_PER_T; ++i) {
                const u32x4_t r = ry[BUF][i];
                float v[4] = {__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w)};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = e < ry_n[BUF][i] ? v[e] : 0.f;
                put(dy_s + y_lds[i], C::DY_PART, v);
            }
#pragma unroll
            for (int i = 0; i < C::X_PER_T; ++i) {
                const u32x4_t r = rx[BUF][i];
                float v[4] = {__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w)};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float u = fmaxf(fmaf(v[e], x_sc[i], x_sh[i]), relu_floor);
                    v[e] = (e >= rx_lo[BUF][i] && e < rx_hi[BUF][i]) ? u : 0.f;
                }
                put(x_s + x_lds[i], C::X_PART, v);
                if (KW == 3) {
                    // boundary words: quad 1 + 2 j ends with x[8 j - 1] (low half of word j), quad 4 + 2 j starts with x[8 j + 8]
                    // (high half of word j); relative to the row image the centre starts at element 8
                    const int q = x_q[i];
                    const bool lo_w = (q & 1) && q <= 7, hi_w = !(q & 1) && q >= 4 && q <= 10;
                    if (lo_w || hi_w) {
                        const float bv = lo_w ? v[3] : v[0];
                        const int j = lo_w ? (q - 1) >> 1 : (q - 4) >> 1;
                        const unsigned u0 = __float_as_uint(bv);
                        const float r1 = bv - __uint_as_float(u0 & 0xffff0000u);
                        const unsigned u1 = __float_as_uint(r1);
                        const float r2 = r1 - __uint_as_float(u1 & 0xffff0000u);
                        unsigned char* pb = x_s + 3 * C::X_PART + x_bnd[i] + j * 4 + (hi_w ? 2 : 0);
                        *reinterpret_cast<unsigned short*>(pb) = (unsigned short)(u0 >> 16);
                        *reinterpret_cast<unsigned short*>(pb + C::XB_PART) = (unsigned short)(u1 >> 16);
                        *reinterpret_cast<unsigned short*>(pb + 2 * C::XB_PART) = (unsigned short)(__float_as_uint(r2) >> 16);
                    }
                }
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        // step S lives in raw set S % 2 and LDS stage S % 2; its loads are issued two steps ahead
        load_step(0, I0{});
        if (nSteps > 1) load_step(1, I1{});
        store_step(0, I0{});
        if (nSteps > 2) load_step(2, I0{});
        __syncthreads();
        auto stage_step = [&](int S, auto nxt_c) __attribute__((always_inline)) {     // consumers: step S; stage step S + 1
            using NXT = decltype(nxt_c);
            if (S + 1 < nSteps) {
                store_step((S + 1) & 1, NXT{});
                if (S + 3 < nSteps) load_step(S + 3, NXT{});
            }
            __syncthreads();
        };
        for (int S = 0; S < nSteps; S += 2) {
            stage_step(S, I1{});
            if (S + 1 < nSteps) stage_step(S + 1, I0{});
        }
    } else if (nSteps > 0) {
        // ================================================================ CONSUMER
        const u32x4_t ones = u32x4_t{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
        const unsigned a_lane = (unsigned)(lr * 64 + (((lq ^ ((-(lr >> 2)) & 3)) & 3) * 16));     // dY fragment: row lr, t group lq
        const unsigned b_lane = KW == 3 ? (unsigned)(lr * C::XCH + 16 + lq * 16) : a_lane;          // x fragment: cin lr, t = 8 lq .. + 7
        const unsigned bnd_lane = (unsigned)(lr * C::XB_CH + lq * 4);                              // its boundary word
        __syncthreads();                                                 // the prologue's images are in place
        for (int S = 0; S < nSteps; ++S) {
            const unsigned char* dy_s = lds + (S & 1) * C::DY_STAGE;
            const unsigned char* x_s = lds + C::X_BASE + (S & 1) * C::X_STAGE;
            u32x4_t af[C::MTL][3];
#pragma unroll
            for (int m = 0; m < C::MTL; ++m)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    af[m][p] = *reinterpret_cast<const u32x4_t*>(dy_s + p * C::DY_PART + ((wmi * C::MTL + m) * 16) * 64 + a_lane);
            // x fragments one cin tile ahead of the MFMAs that use them
            u32x4_t craw[2][3];
            unsigned wraw[2][3];
            auto read_x = [&](int n, u32x4_t (&cr)[3], unsigned (&wr)[3]) __attribute__((always_inline)) {
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    cr[p] = *reinterpret_cast<const u32x4_t*>(x_s + p * C::X_PART + ((wni * C::NTL + n) * 16) * C::XCH + b_lane);
                    // {x[t - 1] (low half), x[t + 8] (high half)} of this lane's 8-element group
                    if (KW == 3)
                        wr[p] = *reinterpret_cast<const unsigned*>(x_s + 3 * C::X_PART + p * C::XB_PART + ((wni * C::NTL + n) * 16) * C::XB_CH + bnd_lane);
                }
            };
            read_x(0, craw[0], wraw[0]);
#pragma unroll
            for (int n = 0; n < C::NTL; ++n) {
                const int cur = n & 1;
                if (n + 1 < C::NTL) read_x(n + 1, craw[cur ^ 1], wraw[cur ^ 1]);
                u32x4_t c[3], left[3], right[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    c[p] = craw[cur][p];
                    if (KW == 3) {
                        const unsigned w = wraw[cur][p];
                        const unsigned s1 = __builtin_amdgcn_alignbit(c[p].y, c[p].x, 16), s2 = __builtin_amdgcn_alignbit(c[p].z, c[p].y, 16),
                                       s3 = __builtin_amdgcn_alignbit(c[p].w, c[p].z, 16);
                        left[p] = u32x4_t{__builtin_amdgcn_perm(c[p].x, w, 0x05040100u), s1, s2, s3};     // x[t-1 .. t+6]
                        right[p] = u32x4_t{s1, s2, s3, __builtin_amdgcn_perm(w, c[p].w, 0x07060302u)};    // x[t+1 .. t+8]
                    }
                }
                // six part products of the (m, kw) accumulators round-robin, smallest first
#pragma unroll
                for (int pp = 0; pp < 6; ++pp) {
                    const int pa = pp == 0 ? 2 : pp == 1 ? 0 : pp == 2 ? 1 : pp == 3 ? 1 : 0;     // lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
                    const int pb = pp == 0 ? 0 : pp == 1 ? 2 : pp == 2 ? 1 : pp == 3 ? 0 : pp == 4 ? 1 : 0;
#pragma unroll
                    for (int m = 0; m < C::MTL; ++m) {
                        if (KW == 3) {
                            acc[m][n][0] = wg_mfma(af[m][pa], left[pb], acc[m][n][0]);
                            acc[m][n][1] = wg_mfma(af[m][pa], c[pb], acc[m][n][1]);
                            acc[m][n][2] = wg_mfma(af[m][pa], right[pb], acc[m][n][2]);
                        } else {
                            acc[m][n][0] = wg_mfma(af[m][pa], c[pb], acc[m][n][0]);
                        }
                    }
                }
            }
            if (do_bias) {
#pragma unroll
                for (int m = 0; m < C::MTL; ++m)
#pragma unroll
                    for (int p = 0; p < 3; ++p) accb[m] = wg_mfma(af[m][2 - p], ones, accb[m]);
            }
            __syncthreads();
        }
    }

    // ---- reduce: transpose the block's partial dW through LDS, then row-contiguous atomics (as conv_wgrad_pc_kernel)
    __syncthreads();
    float* out_s = smem;                             // [OUT_ROWS][OUT_ROW]
    const int ncol = min(C::CIN_T, a.Cin - cin0) * KW;
    const int slot = a.nslots > 1 ? (int)(blockIdx.x % a.nslots) : 0;
    float* dwp = a.dw + (size_t)slot * a.slot_w;
#pragma unroll 1
    for (int part = 0; part < C::COUT_T / C::OUT_ROWS; ++part) {
        if (part > 0) __syncthreads();
        if (consumer && wmi == part) {               // a consumer wave's 64 cout rows = one part
#pragma unroll
            for (int m = 0; m < C::MTL; ++m)
#pragma unroll
                for (int n = 0; n < C::NTL; ++n)
#pragma unroll
                    for (int kk = 0; kk < KW; ++kk)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            out_s[(m * 16 + lq * 4 + r) * C::OUT_ROW + ((wni * C::NTL + n) * 16 + lr) * KW + kk] = acc[m][n][kk][r];
        }
        __syncthreads();
        if (nSteps > 0) {
            for (int row = wave; row < C::OUT_ROWS; row += 8) {
                const int cout = cout0 + part * C::OUT_ROWS + row;
                if (cout >= a.Cout) break;
                float* dst = dwp + ((size_t)cout * a.Cin + cin0) * KW;
                for (int col = lane; col < ncol; col += 64) atomicAdd(dst + col, out_s[row * C::OUT_ROW + col]);
            }
        }
    }
    if (consumer && do_bias && lr == 0 && nSteps > 0) {
#pragma unroll
        for (int m = 0; m < C::MTL; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cout = cout0 + (wmi * C::MTL + m) * 16 + lq * 4 + r;
                if (cout < a.Cout) atomicAdd(&a.db[(size_t)slot * a.slot_b + cout], accb[m][r]);
            }
    }
}

// shapes whose weight-gradient kernel has the BN-backward dY loader (ConvWgradArgs::gx): the dispatch below must agree
bool conv_wgrad_bng_supported(int KH, int KW, int Cin, int Cout, int F, int T, int bf16, int per_cf) {
    (void)F;
    if (bf16 || KH != 3 || KW != 3 || (T & 3)) return false;
    return (Cin >= 64 && Cout >= 64) || (Cin == 32 && Cout == 32 && !per_cf);      // (the 32->32 form with per-row coefficients spills)
}

int conv_wgrad_launch(const ConvWgradArgs& a, int KH, int KW, hipStream_t s) {
    if (a.unpool_idx && (a.F % 2)) { set_error("conv_wgrad: unpool needs even F"); return PBSED_E_ARG; }
    // the loaders address one clip with 32-bit element offsets (buffer loads; 2^29 elements marks "out of range")
    if ((size_t)a.Cin * a.F * a.T >= (1ull << 28) || (size_t)a.Cout * a.F * a.T >= (1ull << 28)) {
        set_error("conv_wgrad: one clip of x / dy must stay below 1 GiB (Cin=%d Cout=%d F=%d T=%d)", a.Cin, a.Cout, a.F, a.T);
        return PBSED_E_ARG;
    }
    if (a.gx && !conv_wgrad_bng_supported(KH, KW, a.Cin, a.Cout, a.F, a.T, a.bf16, a.g_cf)) {
        set_error("conv_wgrad: no kernel with the BN-backward dY loader for %dx%d %d->%d (pbsed_conv_bwd_weight_bng_supported)", KH, KW, a.Cin, a.Cout);
        return PBSED_E_UNSUPPORTED;
    }
    if (wgrad_s16_takes(a, KH, KW)) return launch_wgrad_s16(a, s);      // 16-channel inputs: register-resident column walk
    // Conv1d layers of the fp32 path (F = 1 rows): exact three-way operand splits on the bf16 MFMA - fp32-class gradients - in
    // producer / consumer form from 64 channels on either side.  Measured at B = 32, T = 500: 256->256 k = 3 87 -> 67 us,
    // 2048->256 k = 1 245 -> 166 us (fp32-MFMA kernel before); a 256->256 k = 1 gradient is four 128 x 128 output tiles with
    // eight steps per block of a 64-way split - its fill and reduction outweigh the steps (41 -> 49 us), it stays on the
    // pipelined kernel below, like the narrower layers (up to 1 024 inputs; wider ones: the fp32 kernel's 128-wide cin tiles)
    if (!a.bf16 && KH == 1 && a.F == 1 && !a.unpool_idx && a.Cin >= 64 && a.Cout >= 64) {
        if (KW == 3) return launch_wgrad_cfg<Wgrad1dPcCfg<3>>(conv1d_wgrad_pc_kernel<3>, a, s);
        if (KW == 1 && a.Cin >= 512) return launch_wgrad_cfg<Wgrad1dPcCfg<1>>(conv1d_wgrad_pc_kernel<1>, a, s);
    }
    // 1x1 conv2d layers (F > 1: net_config 'deep'): from 128 channels on (whole 128 x 128 blocks) and without a pool the
    // producer / consumer kernel of the Conv1d layers over (clip, row, 32-t) units (256->256 at F = 16: 0.65 -> 0.36 ms,
    // 512->512 at F = 8: 1.13 -> 0.67; 64->64 at F = 64 with three quarters of the block idle: 0.15 -> 0.33, not taken); the
    // others - also under a pool - the pipelined bf16x3 kernel, which walks (clip, row, 128-t) chunks
    if (!a.bf16 && KH == 1 && KW == 1 && a.F > 1 && !a.unpool_idx && a.Cin >= 128 && a.Cout >= 128 && (a.T & 3) == 0)
        return launch_wgrad_cfg<Wgrad1dPcCfg<1>>(conv1d_wgrad_pc_kernel<1>, a, s);
    if (!a.bf16 && KH == 1 && KW == 1 && a.F > 1 && a.Cin >= 32 && a.Cout >= 32 && a.Cin < 1024)
        return launch_wgrad_cfg<WgradB16Cfg<1, 1, 2, 3>>(conv_wgrad_bf16_kernel<1, 1, 2, 3>, a, s);
    if (!a.bf16 && KH == 1 && a.F == 1 && !a.unpool_idx && a.Cin >= 32 && a.Cout >= 32 && a.Cin < 1024) {
        if (KW == 3) return launch_wgrad_cfg<WgradB16Cfg<1, 3, 2, 3>>(conv_wgrad_bf16_kernel<1, 3, 2, 3>, a, s);
        if (KW == 1) return launch_wgrad_cfg<WgradB16Cfg<1, 1, 2, 3>>(conv_wgrad_bf16_kernel<1, 1, 2, 3>, a, s);
    }
    // 3x3 layers of the fp32 path with >= 64 input and output channels: the producer / consumer column-walking form, 64 cout x
    // 64 cin blocks (128->128 0.326 ms, 128->256 0.317, 64->128 0.215, 64->64 0.218 against 0.432 / 0.419 / 0.240 / 0.258 of the
    // fp32 Winograd weight gradient); the 32->32 layer: the same kernel with its consumer waves slicing the step's time range
    // (KS = 2: 64 t per step, wave = 32 cout x 16 cin x one 32-t slice) - one block covers all of cout x cin: 0.278 -> 0.158 ms.
    // Measured and NOT taken (removed from the tree in round 4): the time-sliced form for 16->16 (0.197 -> 0.252 ms), 16->32
    // and 32->64 - with one or two MFMA tiles per wave and step nothing covers the LDS round trip and the shift arithmetic of
    // the next fragment, and 16->16 has 128 columns for 256 CUs
    if (!a.bf16 && KH == 3 && KW == 3 && (a.T & 3) == 0) {
        if (a.Cin >= 64 && a.Cout >= 64)
            return !a.gx ? launch_wgrad_cfg<WgradPcCfg<2, 2>>(conv_wgrad_pc_kernel<2, 2>, a, s)
                 : a.g_cf ? launch_wgrad_cfg<WgradPcCfg<2, 2, 1, 2, 2, true>>(conv_wgrad_pc_kernel<2, 2, 1, 2, 2, 2>, a, s)
                          : launch_wgrad_cfg<WgradPcCfg<2, 2, 1, 2, 2, true>>(conv_wgrad_pc_kernel<2, 2, 1, 2, 2, 1>, a, s);
        // (32->64 as a time-sliced 64 x 32 block, WgradPcCfg<2, 1, 2, 2, 2>: 0.162 ms against 0.149 of the fp32 Winograd form - not taken)
        if (a.Cin == 32 && a.Cout == 32)
            return !a.gx ? launch_wgrad_cfg<WgradPcCfg<1, 2, 2, 2, 1>>(conv_wgrad_pc_kernel<1, 2, 2, 2, 1>, a, s)
                         : launch_wgrad_cfg<WgradPcCfg<1, 2, 2, 2, 1, true>>(conv_wgrad_pc_kernel<1, 2, 2, 2, 1, 1>, a, s);
    }
    if (a.gx) { set_error("conv_wgrad: no kernel with the BN-backward dY loader for this shape (ask pbsed_conv_bwd_weight_bng_supported)"); return PBSED_E_UNSUPPORTED; }
    if (a.bf16 && a.Cin >= 32 && a.Cout >= 32) {       // bf16-MFMA operands (config 3); few-channel layers stay on the fp32 kernels
        if (KH == 3 && KW == 3) {
            if (a.Cout > 64) return launch_wgrad_cfg<WgradB16Cfg<3, 3, 4>>(conv_wgrad_bf16_kernel<3, 3, 4>, a, s);
            return launch_wgrad_cfg<WgradB16Cfg<3, 3, 2>>(conv_wgrad_bf16_kernel<3, 3, 2>, a, s);
        }
        if (KH == 1 && KW == 3) return launch_wgrad_cfg<WgradB16Cfg<1, 3, 2>>(conv_wgrad_bf16_kernel<1, 3, 2>, a, s);
        if (KH == 1 && KW == 1) return launch_wgrad_cfg<WgradB16Cfg<1, 1, 2>>(conv_wgrad_bf16_kernel<1, 1, 2>, a, s);
    }
    // Configurations measured on MI355X at B=32, T=500: one per channel regime.
    if (KH == 3 && KW == 3) {
        if (a.Cin == 1) return a.Cout >= 64 ? launch_wgrad<3, 3, 4, 1, 1, true>(a, s) : launch_wgrad<3, 3, 1, 1, 1, true>(a, s);
        if (a.Cout >= 64 && a.Cin >= 16) return launch_wgrad_cfg<WinoWgradCfg>(conv_wgrad_wino_kernel, a, s);
        if (a.Cout >= 64) return launch_wgrad<3, 3, 4, 1, 1>(a, s);
        // few output channels: the waves of a block also split the chunk's time range (KWAVES) and chunks are 4 rows
        // tall, otherwise a block is one or two waves and nothing hides the LDS / global latency
        if (a.Cout >= 32) return launch_wgrad<3, 3, 2, 1, 1, false, 2, 4>(a, s);
        return launch_wgrad<3, 3, 1, 1, 1, false, 4, 4>(a, s);
    }
    if (KH == 1 && KW == 3) return a.Cout >= 128 ? launch_wgrad<1, 3, 4, 2, 2>(a, s) : launch_wgrad<1, 3, 1, 1, 2>(a, s);
    if (KH == 1 && KW == 1) {
        // many input channels: 128-wide cin tiles halve the re-reads of dY (2048->256: 0.277 -> 0.248 ms); with few of
        // them the launch has too few block columns (256->256: 0.173 -> 0.247 ms)
        if (a.Cout >= 128 && a.Cin >= 1024) return launch_wgrad<1, 1, 4, 2, 8>(a, s);
        return a.Cout >= 128 ? launch_wgrad<1, 1, 4, 2, 4>(a, s) : launch_wgrad<1, 1, 1, 1, 4>(a, s);
    }
    set_error("conv_wgrad: unsupported kernel %dx%d", KH, KW);
    return PBSED_E_UNSUPPORTED;
}

}  // namespace pbsed
