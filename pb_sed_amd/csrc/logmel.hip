// Fused log-mel front-end for gfx950: waveform -> framing (hop 320, window 960, 'half' fading pad)
// -> periodic Blackman -> rFFT-1024 -> |X|^2 -> sparse triangular mel (128) -> log -> per-bin
// normalise -> clamp -> seq mask, written directly as [B,1,F,T].  The STFT is never materialised
// (the reference moves a 65.7 MB [B,1,T,513,2] tensor per batch-32 step through PCIe + HBM).
//
// Replaces: STFT config pb_sed/data_preparation/provider.py:315-323 (called at
// pb_sed/data_preparation/transform.py:53) + NormalizedLogMelExtractor call
// pb_sed/models/weak_label/crnn.py:86-90 (config pb_sed/experiments/weak_label_crnn/training.py:190-217).
//
// Algorithmic traffic = 4 B/sample in + 4 B/(mel bin, frame) out (896 000 B per 10 s clip); the kernel itself is bound
// by the VALU work of the FFTs (~550 wave instructions per frame) and the mel stage, not by HBM.  Block = 4 waves x FR
// frames; each wave runs whole 512-point complex FFTs (3 radix-8 Stockham passes, one butterfly per lane) in private LDS.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "fft512.h"

namespace pbsed {

constexpr int LM_SHIFT = 320, LM_WIN = 960, LM_FR = 16, LM_NMEL_MAX = 128;
constexpr int LM_MW_MAX = 1152;      // packed non-zero filter weights staged in LDS (128 HTK-mel filters over 513 bins: ~1030); two blocks share a CU (74.5 KB each)

struct LogmelArgs {
    const float* wav;        // [B][N]
    const float* window;     // [960]
    const float* twiddle;    // [1024][2] exp(-2 pi i q/1024)
    const int* mel_start;    // [F] first bin of filter m
    const int* mel_len;      // [F] number of bins
    const int* mel_off;      // [F] offset into mel_w
    const float* mel_w;      // flat weights
    const float* mean;       // [F]
    const float* inv_std;    // [F]
    const int* seq_len;      // [B] frames, or null
    float* out;              // [B][1][F][T]
    double* stats;           // optional [PBSED_STAT_SLOTS][F][2]: masked sum / sum of squares of the written values
    int B, N, T, F;
    float eps, clampv;
    const float* mel_pts;    // optional [B][F+2]: per-clip fractional bin positions of the (warped) filter edges / centres
    int pad_front;           // zero samples assumed in front of wav[0] (320 = the reference's 'half' fading; 0 for a slice
                             // cut out of the middle of a clip, pb_sed_amd/utils/segment.py)
    const int* frame_pos;    // optional [B][T]: first sample of every frame's window (may lie outside [0, N): zeros) - the
                             // time-warped STFT of the training pipeline (pb_sed/data_preparation/transform.py:36-45);
                             // null: frame t starts at 320 t - pad_front
};

// Per-mel sums of one block's output tile (frames past seq_len hold 0 and add nothing) into one of the slot copies.
__device__ __forceinline__ void logmel_tile_stats(const float* tile, double* stats, int F, int tid) {
    double* dst = stats + (size_t)(blockIdx.x % PBSED_STAT_SLOTS) * F * 2;
    for (int m = tid; m < F; m += 256) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int fl = 0; fl < LM_FR; ++fl) {
            const float v = tile[m * (LM_FR + 1) + fl];
            s += v;
            q = fmaf(v, v, q);
        }
        atomicAdd(dst + 2 * m, (double)s);
        atomicAdd(dst + 2 * m + 1, (double)q);
    }
}

// One mel value from a power-spectrum row.  Static filterbank: sparse rows of the packed table.  Per-clip warped
// filterbank (training-time MelWarping, pb_sed/experiments/weak_label_crnn/training.py:195-208): filter m is the
// triangle over the fractional bin positions pts[m] < pts[m+1] < pts[m+2] of THIS clip, normalised to unit sum
// (paderbox get_fbanks semantics), its weights computed on the fly - no per-clip table is ever materialised.
__device__ __forceinline__ float mel_triangle(const float* P, int bins, float lo, float c, float hi) {
    const float il = 1.f / (c - lo), ih = 1.f / (hi - c);
    const int k0 = max((int)ceilf(lo), 0), k1 = min((int)floorf(hi), bins - 1);
    float s = 0.f, wsum = 0.f;
    for (int k = k0; k <= k1; ++k) {
        const float w = fmaxf(fminf(((float)k - lo) * il, (hi - (float)k) * ih), 0.f);
        wsum += w;
        s = fmaf(P[k], w, s);
    }
    return wsum > 0.f ? s / wsum : 0.f;
}

__device__ __forceinline__ float mel_from_power(const float* P, int m, int bins, const int* mel_start, const int* mel_len,
                                                const int* mel_off, const float* mel_w, const float* pts) {
    float s = 0.f;
    if (pts) return mel_triangle(P, bins, pts[m], pts[m + 1], pts[m + 2]);
    const int st = mel_start[m], ln = mel_len[m];
    const float* w = mel_w + mel_off[m];
    for (int i = 0; i < ln; ++i) s = fmaf(P[st + i], w[i], s);
    return s;
}

// A wave's FFT buffers, power row and output-tile columns are private to it: between the phases of one frame only the
// wave's own LDS traffic has to be ordered (the LDS unit executes one wave's ds_* instructions in issue order), so the
// per-frame loop needs no block-wide barrier - the four waves of a block drift apart and hide each other's latencies.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---- the fused front-end kernel.  Block = 16 frames of one clip, 4 waves; a wave transforms its 4 frames (two at a time),
// then the block runs the mel stage over all 16.  Built around few LDS instructions per frame (the round-2 form - samples,
// window and twiddles in LDS, ping-pong buffers, a gather loop per mel filter - was LDS-instruction-bound at ~250 wave-level
// ds_* per frame: 63 us for 32 clips; this one 40 us, 47 with the statistics atomics):
//  * samples straight from the clip through a raw buffer (out-of-range = 0 without a branch), the next pair of frames
//    requested during the current pair's passes;
//  * window, the pass twiddles and the rFFT twiddles of a lane's eight bins live in registers for all frames of the block;
//  * pass 1 reads the samples and multiplies by the window on the fly (no packed copy), the three radix-8 passes share ONE
//    buffer per wave, in place (a wave's LDS instructions execute in order: every lane has its eight inputs before the
//    first output is written), at index i + (i >> 3) - the 8-consecutive / stride-64 / stride-8 access patterns of the three
//    passes are then free of bank conflicts;
//  * pass 3 leaves Z[lane + 64 r] in registers; the mirror bins Z[512 - k] of the real-FFT unpacking come through
//    ds_bpermute from lane 64 - lane (register 7 - r), nothing is written back;
//  * the mel filterbank is a banded GEMM on the fp32 MFMA over the block's 16 frames: A = filter weights [16 filters x 4 bins]
//    (looked up in the packed sparse table, or the warped triangle computed on the fly), B = power rows [4 bins x 16 frames];
//    a group of 16 filters only walks its own band of bins (~150 MFMAs per block for 128 HTK-mel filters instead of a
//    variable-length gather loop whose longest lane does ~40 taps per frame).  The waves take groups (w, 7 - w) - narrow and
//    wide bands paired.
#ifndef LM_DBG
#define LM_DBG 0             // ablation switches of tools/kernel_ablation.sh (never set in the product build): 1 no FFT frames, 2 no mel, 4 no sample staging
#endif
constexpr int LM_PS = 516;           // power-row stride in floats (4 mod 64 banks: the 16 frames x 4 bins of a B fragment hit 64 banks)
constexpr int LM_FB = 576;           // cpx per wave FFT buffer: 512 + 512 / 8 padding

__device__ __forceinline__ int fft_phys(int i) { return i + (i >> 3); }

__global__ __launch_bounds__(256) void logmel_kernel(LogmelArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    cpx* bufs = reinterpret_cast<cpx*>(smem);                     // [4 waves][2][LM_FB]
    float* P = reinterpret_cast<float*>(bufs + 8 * LM_FB);        // [LM_FR][LM_PS] power rows
    float* mw = P + LM_FR * LM_PS;                                // packed filter weights (static filterbank)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane >> 4, lr = lane & 15;
    const int nTt = (a.T + LM_FR - 1) / LM_FR;
    // neighbouring 16-frame tiles of a clip write the two halves of the same 128-byte output lines: workgroups go to the XCDs
    // round-robin, so the linear id is re-read as (xcd, slot) and every XCD gets a contiguous range of tiles
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (bid & 7) * (int)(gridDim.x >> 3) + (bid >> 3);
    const int b = bid / nTt, t0 = (bid % nTt) * LM_FR;
    const int* fpos = a.frame_pos ? a.frame_pos + (size_t)b * a.T : nullptr;
    // the clip as a raw buffer: samples before its start / past its end read as 0 without a branch (byte offsets wrap to
    // values above the clip's size; the launcher keeps clips below 2^29 samples)
    const __amdgpu_buffer_rsrc_t clip = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wav + (size_t)b * a.N), 0, (unsigned)a.N * 4u, 0x00020000);
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    // samples 2 n, 2 n + 1 (n = lane + 64 r) of the frame starting at sample s; a regular grid starts frames at even samples,
    // so a pair never straddles the clip's start and one 8-byte load per pair is in range or not per dword
    auto load_frame = [&](int fl, u32x2_t (&raw)[8]) __attribute__((always_inline)) {
        const int t = t0 + fl;
        if (t >= a.T || (LM_DBG & 4)) {
#pragma unroll
            for (int r = 0; r < 8; ++r) raw[r] = u32x2_t{0u, 0u};
            return;
        }
        const int s = fpos ? fpos[t] : t * LM_SHIFT - a.pad_front;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int n = lane + 64 * r;
            if (2 * n >= LM_WIN) { raw[r] = u32x2_t{0u, 0u}; continue; }
            const int i0 = s + 2 * n;
            const unsigned off = (unsigned)i0 * 4u;
            if (fpos || (a.pad_front & 1)) {
                // arbitrary starts: a pair may straddle the clip's start (i0 = -1); two loads whose offsets do not differ by a
                // constant 4 (the compiler would fuse them into one 8-byte load whose second half wraps past 2^32 = out of range)
                const unsigned off1 = i0 + 1 >= 0 ? (unsigned)(i0 + 1) * 4u : 0x80000000u;
                raw[r] = u32x2_t{__builtin_amdgcn_raw_buffer_load_b32(clip, i0 >= 0 ? off : 0x80000000u, 0, 0),
                                 __builtin_amdgcn_raw_buffer_load_b32(clip, off1, 0, 0)};
            } else {
                raw[r] = __builtin_amdgcn_raw_buffer_load_b64(clip, off, 0, 0);
            }
        }
    };
    u32x2_t raw0[8], raw1[8];
    load_frame(wave * (LM_FR / 4), raw0);                         // the first two frames' samples are on their way during the set-up
    load_frame(wave * (LM_FR / 4) + 1, raw1);

    const float* pts = a.mel_pts ? a.mel_pts + (size_t)b * (a.F + 2) : nullptr;
    const int n_mw = pts ? 0 : a.mel_off[a.F - 1] + a.mel_len[a.F - 1];
    for (int i = tid; i < n_mw; i += 256) mw[i] = a.mel_w[i];

    // per-lane constants: window pairs of the samples 2 n, 2 n + 1, twiddles of pass 2 (k = lane & 7), pass 3 (k = lane) and of
    // the real-FFT unpacking of the bins lane + 64 r
    const float2* twid = reinterpret_cast<const float2*>(a.twiddle);
    float2 win2[8];
    cpx tw2[7], tw3[7], twp[8];
    const int k2 = lane & 7;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int n = lane + 64 * r;
        win2[r] = (2 * n < LM_WIN) ? *reinterpret_cast<const float2*>(a.window + 2 * n) : make_float2(0.f, 0.f);
        const float2 w = twid[lane + 64 * r];
        twp[r] = cpx{w.x, w.y};
        if (r > 0) {
            const float2 u = twid[(r * k2 * 16) & 1023], v = twid[(r * lane * 2) & 1023];
            tw2[r - 1] = cpx{u.x, u.y};
            tw3[r - 1] = cpx{v.x, v.y};
        }
    }

    // the mel stage's descriptors: this lane's A row (filter 16 g + lr) and the constants of its four output rows
    // (filters 16 g + 4 lq + i); requested here, used after the FFTs
    struct MelDesc { int st, en, off; float lo, hi, il, ih, wsum, mean[4], is[4]; };
    auto load_desc = [&](int g) __attribute__((always_inline)) -> MelDesc {
        MelDesc d;
        const int m = 16 * g + lr;
        d.st = 1 << 30; d.en = 0; d.off = 0; d.lo = d.hi = d.il = d.ih = 0.f; d.wsum = 1.f;
        if (m < a.F) {
            if (pts) {
                d.lo = pts[m]; d.hi = pts[m + 2];
                const float cc = pts[m + 1];
                d.il = 1.f / (cc - d.lo); d.ih = 1.f / (d.hi - cc);
                d.st = max((int)ceilf(d.lo), 0); d.en = min((int)floorf(d.hi), 512) + 1;
                d.wsum = 0.f;
                for (int k = d.st; k < d.en; ++k) d.wsum += fmaxf(fminf(((float)k - d.lo) * d.il, (d.hi - (float)k) * d.ih), 0.f);
            } else {
                d.st = a.mel_start[m]; d.en = d.st + a.mel_len[m]; d.off = a.mel_off[m] - d.st;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int mo = min(16 * g + 4 * lq + i, a.F - 1);
            d.mean[i] = a.mean[mo]; d.is[i] = a.inv_std[mo];
        }
        return d;
    };
    const int nG = (a.F + 15) / 16;
    // groups of a wave: (w, 7 - w) of every eight - narrow and wide bands paired
    auto group_of = [&](int gi) { return 8 * (gi >> 1) + ((gi & 1) ? 7 - wave : wave); };
    MelDesc d0 = load_desc(min(group_of(0), nG - 1)), d1 = load_desc(min(group_of(1), nG - 1));

    // two frames at a time per wave (two independent chains of LDS round trips), each in its own buffer; a frame past the
    // clip's last one transforms zeros (its power row must be finite for the MFMAs)
    cpx* buf0 = bufs + (2 * wave) * LM_FB;
    cpx* buf1 = buf0 + LM_FB;
    const int j0 = (lane - k2) * 8 + k2;
    const int mirror = (64 - lane) & 63;
    auto power_row = [&](const cpx (&v)[8], float* Prow) __attribute__((always_inline)) {
        // |X[k]|^2 of the 1024-point real FFT, k = lane + 64 r: needs Z[(512 - k) & 511] = register 7 - r of lane 64 - lane
        // (lane 0: its own register 8 - r)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            cpx zr = cpx{__shfl(v[7 - r].x, mirror), __shfl(v[7 - r].y, mirror)};
            if (lane == 0) zr = v[(8 - r) & 7];
            const cpx zk = v[r];
            const cpx e = cpx{.5f * (zk.x + zr.x), .5f * (zk.y - zr.y)};
            const cpx d = cpx{zk.x - zr.x, zk.y + zr.y};
            const cpx o = cpx{.5f * d.y, -.5f * d.x};
            const cpx xk = cadd(e, cmul(twp[r], o));
            Prow[lane + 64 * r] = xk.x * xk.x + xk.y * xk.y;
        }
        if (lane == 0) { const float n = v[0].x - v[0].y; Prow[512] = n * n; }
        else if (lane < 4) Prow[512 + lane] = 0.f;
    };
#pragma unroll
    for (int fp = 0; fp < LM_FR / 8; ++fp) {
        const int fl = wave * (LM_FR / 4) + 2 * fp;
        cpx v[8], u[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            v[r] = cpx{__uint_as_float(raw0[r].x) * win2[r].x, __uint_as_float(raw0[r].y) * win2[r].y};
            u[r] = cpx{__uint_as_float(raw1[r].x) * win2[r].x, __uint_as_float(raw1[r].y) * win2[r].y};
        }
        if (fp + 1 < LM_FR / 8) { load_frame(fl + 2, raw0); load_frame(fl + 3, raw1); }     // the next pair's samples during this pair's passes
        if (LM_DBG & 1) continue;
        // pass 1 (no twiddles): butterfly `lane` of stride 64 -> outputs 8 lane .. 8 lane + 7
        dft8(v); dft8(u);
#pragma unroll
        for (int r = 0; r < 8; ++r) { buf0[fft_phys(8 * lane + r)] = v[r]; buf1[fft_phys(8 * lane + r)] = u[r]; }
        wave_lds_sync();
        // pass 2
#pragma unroll
        for (int r = 0; r < 8; ++r) { v[r] = buf0[fft_phys(lane + 64 * r)]; u[r] = buf1[fft_phys(lane + 64 * r)]; }
#pragma unroll
        for (int r = 1; r < 8; ++r) { v[r] = cmul(v[r], tw2[r - 1]); u[r] = cmul(u[r], tw2[r - 1]); }
        dft8(v); dft8(u);
        wave_lds_sync();
#pragma unroll
        for (int r = 0; r < 8; ++r) { buf0[fft_phys(j0 + 8 * r)] = v[r]; buf1[fft_phys(j0 + 8 * r)] = u[r]; }
        wave_lds_sync();
        // pass 3: Z[lane + 64 r] stays in registers
#pragma unroll
        for (int r = 0; r < 8; ++r) { v[r] = buf0[fft_phys(lane + 64 * r)]; u[r] = buf1[fft_phys(lane + 64 * r)]; }
#pragma unroll
        for (int r = 1; r < 8; ++r) { v[r] = cmul(v[r], tw3[r - 1]); u[r] = cmul(u[r], tw3[r - 1]); }
        dft8(v); dft8(u);
        wave_lds_sync();                                           // the next pair's pass 1 rewrites the buffers
        power_row(v, P + fl * LM_PS);
        power_row(u, P + (fl + 1) * LM_PS);
    }
    if (LM_DBG & 1)
        for (int k = tid; k < LM_FR * LM_PS; k += 256) P[k] = 0.f;
    __syncthreads();

    // ---- mel as a banded GEMM: out[m][frame] = sum_k W[m][k] P[frame][k]
    const int sl = a.seq_len ? min(a.seq_len[b], a.T) : a.T;
    double* stat = a.stats ? a.stats + (size_t)(bid % PBSED_STAT_SLOTS) * a.F * 2 : nullptr;
    const float* prow = P + lr * LM_PS;
    const int t = t0 + lr;
    const bool tv = t < a.T;
    for (int gi = 0; !(LM_DBG & 2); ++gi) {
        const int g = group_of(gi);
        if (8 * (gi >> 1) >= nG) break;
        if (g >= nG) continue;
        const MelDesc d = gi == 0 ? d0 : gi == 1 ? d1 : load_desc(g);
        int kb = d.st, ke = d.en;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { kb = min(kb, __shfl_xor(kb, o)); ke = max(ke, __shfl_xor(ke, o)); }
        kb &= ~3;
        // four k-steps per trip: the weights and power values of a trip are requested together, two accumulators alternate
        // (a dependent fp32 MFMA chain issues every ~40 clocks)
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f}, acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
        // kb / ke are the same in every lane (all four 16-lane rows reduce the same filters): scalar loop bounds.  The weight
        // look-up is straight-line (an index select, one LDS read, a value select): no branch around the read, or every
        // read would be waited for on its own
        const int kbu = __builtin_amdgcn_readfirstlane(kb), keu = (LM_DBG & 16) ? kbu : __builtin_amdgcn_readfirstlane(ke);
        auto fetch = [&](int k0, float (&w)[4], float (&pv)[4], auto warped_c) __attribute__((always_inline)) {
            constexpr bool WARPED = decltype(warped_c)::value;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = min(k0 + 4 * q + lq, LM_PS - 1);      // steps past the band: weight 0 x a finite value
                const bool in = (k >= d.st) & (k < d.en) & (k0 + 4 * q < keu);
                float wv;
                if constexpr (WARPED) wv = fmaxf(fminf(((float)k - d.lo) * d.il, (d.hi - (float)k) * d.ih), 0.f);
                else wv = mw[in ? d.off + k : 0];
                w[q] = in ? wv : 0.f;
                pv[q] = prow[k];
            }
        };
        auto band = [&](auto warped_c) __attribute__((always_inline)) {
            float w[4], pv[4];
            fetch(kbu, w, pv, warped_c);
            for (int k0 = kbu; k0 < keu; k0 += 16) {
                float wn[4], pn[4];
                fetch(k0 + 16, wn, pn, warped_c);                  // the next trip's operands during this trip's MFMAs
                acc = mfma16(w[0], pv[0], acc);
                acc2 = mfma16(w[1], pv[1], acc2);
                acc = mfma16(w[2], pv[2], acc);
                acc2 = mfma16(w[3], pv[3], acc2);
#pragma unroll
                for (int q = 0; q < 4; ++q) { w[q] = wn[q]; pv[q] = pn[q]; }
            }
        };
        if (pts) band(std::true_type{}); else band(std::false_type{});
        acc += acc2;
        // D[row 4 lq + i = filter][col lr = frame]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int mo = 16 * g + 4 * lq + i;
            float s = acc[i];
            if (pts) {
                const float ws = __shfl(d.wsum, 4 * lq + i);
                s = ws > 0.f ? s / ws : 0.f;
            }
            float val = 0.f;
            if (mo < a.F) {
                val = (logf(s + a.eps) - d.mean[i]) * d.is[i];
                val = fminf(fmaxf(val, -a.clampv), a.clampv);
                val = (tv && t < sl) ? val : 0.f;
                if (tv && !(LM_DBG & 8)) a.out[((size_t)b * a.F + mo) * a.T + t] = val;
            }
            if (stat) {
                const float sum = wave_sum16(val), sq = wave_sum16(val * val);
                if (lr == 0 && mo < a.F) {
                    atomicAdd(stat + 2 * mo, (double)sum);
                    atomicAdd(stat + 2 * mo + 1, (double)sq);
                }
            }
        }
    }
}

// The reference's own input contract: a CPU-computed complex STFT inputs['stft'] [B,1,T,bins,2] fp32
// (pb_sed/models/weak_label/crnn.py:31,80-83) -> |X|^2 -> sparse triangular mel -> log -> norm -> clamp -> seq mask,
// written as [B,1,F,T] (pb_sed/models/weak_label/crnn.py:86-90).  HBM-bound: 8 B/(bin,frame) in + 4 B/(mel,frame) out.
// Block = LM_FR frames of one clip: the re/im pairs are read as coalesced float2 rows, squared into an LDS power tile
// [LM_FR][bins+1], each thread then owns (mel, frame) outputs; the output tile is transposed in LDS for t-contiguous stores.
struct LogmelStftArgs {
    const float* stft;       // [B][T][bins][2]
    const int* mel_start;
    const int* mel_len;
    const int* mel_off;
    const float* mel_w;
    const float* mean;
    const float* inv_std;
    const int* seq_len;
    float* out;              // [B][1][F][T]
    double* stats;           // optional, as LogmelArgs::stats
    int B, T, F, bins;
    float eps, clampv;
    const float* mel_pts;    // optional, as LogmelArgs::mel_pts
};

__global__ __launch_bounds__(256) void logmel_from_stft_kernel(LogmelStftArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int PS = a.bins + 1;                                    // odd row stride: frames land in different banks
    float* P = smem;                                              // [LM_FR][PS]
    float* tile = P + LM_FR * PS;                                 // [F][LM_FR+1]
    float* mw = tile + a.F * (LM_FR + 1);                         // packed filter weights (static filterbank)
    const int tid = threadIdx.x;
    const int n_mw = a.mel_pts ? 0 : a.mel_off[a.F - 1] + a.mel_len[a.F - 1];
    for (int i = tid; i < n_mw; i += 256) mw[i] = a.mel_w[i];
    const int nTt = (a.T + LM_FR - 1) / LM_FR;
    const int b = blockIdx.x / nTt, t0 = (blockIdx.x % nTt) * LM_FR;
    const int nfr = min(LM_FR, a.T - t0);
    const float2* src = reinterpret_cast<const float2*>(a.stft) + ((size_t)b * a.T + t0) * a.bins;
    const int total = nfr * a.bins;                               // the nfr frames are contiguous in memory
    for (int i = tid; i < total; i += 256) {
        const float2 v = src[i];
        const int fl = i / a.bins, k = i - fl * a.bins;
        P[fl * PS + k] = fmaf(v.x, v.x, v.y * v.y);
    }
    __syncthreads();
    const int sl = a.seq_len ? min(a.seq_len[b], a.T) : a.T;
    for (int i = tid; i < a.F * LM_FR; i += 256) {
        const int fl = i % LM_FR, m = i / LM_FR;                  // 16 consecutive lanes share a filter (broadcast weights)
        float v = 0.f;
        if (fl < nfr) {
            const float s = mel_from_power(P + fl * PS, m, a.bins, a.mel_start, a.mel_len, a.mel_off, mw,
                                           a.mel_pts ? a.mel_pts + (size_t)b * (a.F + 2) : nullptr);
            v = (logf(s + a.eps) - a.mean[m]) * a.inv_std[m];
            v = fminf(fmaxf(v, -a.clampv), a.clampv);
            if (t0 + fl >= sl) v = 0.f;
        }
        tile[m * (LM_FR + 1) + fl] = v;
    }
    __syncthreads();
    if (a.stats) logmel_tile_stats(tile, a.stats, a.F, tid);
    for (int i = tid; i < a.F * LM_FR; i += 256) {
        const int m = i / LM_FR, fl = i % LM_FR;
        if (fl < nfr) a.out[((size_t)b * a.F + m) * a.T + t0 + fl] = tile[m * (LM_FR + 1) + fl];
    }
}

// Training-time augmentation of the normalised log-mel features (NormalizedLogMelExtractor config of
// pb_sed/experiments/weak_label_crnn/training.py:209-216; padertorch semantics restated): additive
// noise scale[b] * noise, then one time mask [t_on, t_off) and one frequency mask [f_on, f_off) per clip set to 0,
// then the sequence mask again.  The random draws (scales, mask positions, the N(0,1) field) are the caller's.
__global__ __launch_bounds__(256) void augment_logmel_kernel(float* __restrict__ x, const float* __restrict__ noise,
                                                             const float* __restrict__ noise_scale,
                                                             const int* __restrict__ masks /*[B][4]*/,
                                                             const int* __restrict__ seq_len,
                                                             const float* __restrict__ mean, const float* __restrict__ inv_std,
                                                             float clampv, int B, int F, int T) {
    const size_t total = (size_t)B * F * T;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int t = i % T, f = (i / T) % F, b = i / ((size_t)T * F);
        float v = x[i];
        if (mean) v = fminf(fmaxf((v - mean[f]) * inv_std[f], -clampv), clampv);   // raw log-mel of a statistics-tracking pass
        if (noise) v = fmaf(noise_scale[b], noise[i], v);
        bool dropped = seq_len && t >= seq_len[b];
        if (masks) {
            const int* m = masks + 4 * b;
            dropped = dropped || (t >= m[0] && t < m[1]) || (f >= m[2] && f < m[3]);
        }
        x[i] = dropped ? 0.f : v;
    }
}

// Cumulative per-mel statistics of the feature normalisation (padertorch Normalization(momentum=None) inside
// NormalizedLogMelExtractor): running mean / power over every valid (clip, frame) seen so far, and the
// (mean, 1/std) pair the normalising pass uses.  One block.
__global__ __launch_bounds__(256) void feature_norm_update_kernel(const double* __restrict__ stats, double count,
                                                                  float* __restrict__ running_mean, float* __restrict__ running_power,
                                                                  double* __restrict__ num_tracked, float eps,
                                                                  float* __restrict__ mean, float* __restrict__ inv_std, int F) {
    const double n0 = *num_tracked;
    __syncthreads();
    for (int m = threadIdx.x; m < F; m += blockDim.x) {
        double s = 0., q = 0.;
        for (int k = 0; k < PBSED_STAT_SLOTS; ++k) {
            s += stats[((size_t)k * F + m) * 2];
            q += stats[((size_t)k * F + m) * 2 + 1];
        }
        const double n1 = n0 + count;
        const double mu = n1 > 0. ? (n0 * (double)running_mean[m] + s) / n1 : 0.;
        const double pw = n1 > 0. ? (n0 * (double)running_power[m] + q) / n1 : 1.;
        running_mean[m] = (float)mu;
        running_power[m] = (float)pw;
        mean[m] = (float)mu;
        inv_std[m] = (float)(1. / sqrt(fmax(pw - mu * mu, 0.) + (double)eps));
    }
    __syncthreads();
    if (threadIdx.x == 0) *num_tracked = n0 + count;
}

}  // namespace pbsed

using namespace pbsed;

extern "C" int pbsed_feature_norm_update(const double* stats, double count, float* running_mean, float* running_power,
                                         double* num_tracked, float eps, float* mean, float* inv_std, int F, void* stream) {
    if (!stats || !running_mean || !running_power || !num_tracked || !mean || !inv_std || F < 1) {
        set_error("feature_norm_update: null argument");
        return PBSED_E_ARG;
    }
    hipLaunchKernelGGL(feature_norm_update_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, stats, count, running_mean,
                       running_power, num_tracked, eps, mean, inv_std, F);
    return check_launch("feature_norm_update");
}

extern "C" int pbsed_augment_logmel(float* x, const float* noise, const float* noise_scale, const int* masks,
                                    const int* seq_len, const float* mean, const float* inv_std, float clampv,
                                    int B, int F, int T, void* stream) {
    if ((noise && !noise_scale) || (mean && !inv_std)) { set_error("augment_logmel: noise needs noise_scale, mean needs inv_std"); return PBSED_E_ARG; }
    const size_t total = (size_t)B * F * T;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(augment_logmel_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, (hipStream_t)stream, x, noise, noise_scale,
                       masks, seq_len, mean, inv_std, clampv, B, F, T);
    return check_launch("augment_logmel");
}

static int logmel_launch(const float* wav, int B, int n_samples, int T, const int* seq_len_frames,
                         const float* window, const float* twiddle, const int* mel_start,
                         const int* mel_len, const int* mel_off, const float* mel_w, int mel_w_count, int F,
                         const float* mean, const float* inv_std, float eps, float clampv,
                         float* out, double* stats, int pad_front, const int* frame_pos, const float* mel_pts, void* stream) {
    if (pad_front < 0 || pad_front > LM_WIN) { set_error("logmel: pad_front %d", pad_front); return PBSED_E_ARG; }
    if (F > LM_NMEL_MAX * 4 || F < 1 || T < 1) { set_error("logmel: bad F=%d T=%d", F, T); return PBSED_E_ARG; }
    if (mel_w_count > LM_MW_MAX || mel_w_count < 0) { set_error("logmel: %d packed filter weights (max %d)", mel_w_count, LM_MW_MAX); return PBSED_E_UNSUPPORTED; }
    LogmelArgs a{wav, window, twiddle, mel_start, mel_len, mel_off, mel_w, mean, inv_std, seq_len_frames,
                 out, stats, B, n_samples, T, F, eps, clampv, mel_pts, pad_front, frame_pos};
    const int nTt = (T + LM_FR - 1) / LM_FR;
    if (n_samples >= (1 << 29)) { set_error("logmel: clips of %d samples (the loader addresses 2^29)", n_samples); return PBSED_E_UNSUPPORTED; }
    const size_t lds = (LM_FR * LM_PS + LM_MW_MAX) * sizeof(float) + 8 * LM_FB * sizeof(cpx);
    PBSED_DYN_LDS_ONCE(logmel_kernel, lds);
    hipLaunchKernelGGL(logmel_kernel, dim3(B * nTt), dim3(256), lds, (hipStream_t)stream, a);
    return check_launch("logmel_fwd");
}

extern "C" int pbsed_logmel_fwd(const float* wav, int B, int n_samples, int T, const int* seq_len_frames,
                                const float* window, const float* twiddle, const int* mel_start,
                                const int* mel_len, const int* mel_off, const float* mel_w, int mel_w_count, int F,
                                const float* mean, const float* inv_std, float eps, float clampv,
                                float* out, double* stats, int pad_front, const float* mel_pts, void* stream) {
    return logmel_launch(wav, B, n_samples, T, seq_len_frames, window, twiddle, mel_start, mel_len, mel_off, mel_w, mel_w_count,
                         F, mean, inv_std, eps, clampv, out, stats, pad_front, nullptr, mel_pts, stream);
}

extern "C" int pbsed_logmel_fwd_frames(const float* wav, int B, int n_samples, int T, const int* seq_len_frames,
                                       const int* frame_pos, const float* window, const float* twiddle, const int* mel_start,
                                       const int* mel_len, const int* mel_off, const float* mel_w, int mel_w_count, int F,
                                       const float* mean, const float* inv_std, float eps, float clampv,
                                       float* out, double* stats, const float* mel_pts, void* stream) {
    if (!frame_pos) { set_error("logmel_fwd_frames: frame_pos is null"); return PBSED_E_ARG; }
    return logmel_launch(wav, B, n_samples, T, seq_len_frames, window, twiddle, mel_start, mel_len, mel_off, mel_w, mel_w_count,
                         F, mean, inv_std, eps, clampv, out, stats, 0, frame_pos, mel_pts, stream);
}

extern "C" int pbsed_logmel_from_stft(const float* stft, int B, int T, int bins, const int* seq_len_frames,
                                      const int* mel_start, const int* mel_len, const int* mel_off, const float* mel_w,
                                      int mel_w_count, int F, const float* mean, const float* inv_std, float eps, float clampv,
                                      float* out, double* stats, const float* mel_pts, void* stream) {
    if (F < 1 || T < 1 || bins < 2 || B < 1) { set_error("logmel_from_stft: bad B=%d T=%d bins=%d F=%d", B, T, bins, F); return PBSED_E_ARG; }
    if (mel_w_count > LM_MW_MAX || mel_w_count < 0) { set_error("logmel_from_stft: %d packed filter weights (max %d)", mel_w_count, LM_MW_MAX); return PBSED_E_UNSUPPORTED; }
    const size_t lds = ((size_t)LM_FR * (bins + 1) + (size_t)F * (LM_FR + 1) + LM_MW_MAX) * sizeof(float);
    if (lds > 160 * 1024) { set_error("logmel_from_stft: %d bins x %d filters need %zu B of LDS", bins, F, lds); return PBSED_E_UNSUPPORTED; }
    LogmelStftArgs a{stft, mel_start, mel_len, mel_off, mel_w, mean, inv_std, seq_len_frames, out, stats, B, T, F, bins, eps, clampv, mel_pts};
    PBSED_DYN_LDS_ONCE(logmel_from_stft_kernel, lds);
    const int nTt = (T + LM_FR - 1) / LM_FR;
    hipLaunchKernelGGL(logmel_from_stft_kernel, dim3(B * nTt), dim3(256), lds, (hipStream_t)stream, a);
    return check_launch("logmel_from_stft");
}
