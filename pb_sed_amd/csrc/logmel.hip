// Fused log-mel front-end for gfx950: waveform -> framing (hop 320, window 960, 'half' fading pad)
// -> periodic Blackman -> rFFT-1024 -> |X|^2 -> sparse triangular mel (128) -> log -> per-bin
// normalise -> clamp -> seq mask, written directly as [B,1,F,T].  The STFT is never materialised
// (the reference moves a 65.7 MB [B,1,T,513,2] tensor per batch-32 step through PCIe + HBM).
//
// Replaces: STFT config pb_sed/data_preparation/provider.py:315-323 (called at
// pb_sed/data_preparation/transform.py:53) + NormalizedLogMelExtractor call
// pb_sed/models/weak_label/crnn.py:86-90 (config pb_sed/experiments/weak_label_crnn/training.py:190-217).
//
// HBM-bound by design: algorithmic traffic = 4 B/sample in + 4 B/(mel bin, frame) out
// (896 000 B per 10 s clip).  Block = 4 waves x FR frames; each wave runs whole 512-point complex
// FFTs (3 radix-8 Stockham passes, one butterfly per lane) in its private LDS ping-pong buffers.
#include "common.h"
#include "fft512.h"

namespace pbsed {

constexpr int LM_SHIFT = 320, LM_WIN = 960, LM_FR = 16, LM_NMEL_MAX = 128;
constexpr int LM_MW_MAX = 1152;      // packed non-zero filter weights staged in LDS (128 HTK-mel filters over 513 bins: ~1030); sized so that two blocks share a CU (79.3 KB each)

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

__global__ __launch_bounds__(256) void logmel_kernel(LogmelArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NS_SAMP = (LM_FR - 1) * LM_SHIFT + LM_WIN;      // 5760
    float* samp = smem;                                           // [5760]
    float* win = samp + NS_SAMP;                                  // [960]
    cpx* tw = reinterpret_cast<cpx*>(win + LM_WIN);               // [1024]
    cpx* bufs = tw + 1024;                                        // [4 waves][2][512]
    float* tile = reinterpret_cast<float*>(bufs + 4 * 2 * 512);   // [F][LM_FR+1]
    float* mw = tile + a.F * (LM_FR + 1);                         // packed filter weights (static filterbank)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nTt = (a.T + LM_FR - 1) / LM_FR;
    const int b = blockIdx.x / nTt, t0 = (blockIdx.x % nTt) * LM_FR;
    const long s0 = (long)t0 * LM_SHIFT - a.pad_front;
    const float* wav = a.wav + (size_t)b * a.N;
    const int* fpos = a.frame_pos ? a.frame_pos + (size_t)b * a.T : nullptr;
    if (!fpos) {
        for (int i = tid; i < NS_SAMP; i += 256) {
            const long n = s0 + i;
            samp[i] = (n >= 0 && n < a.N) ? wav[n] : 0.f;
        }
    }
    const int n_mw = a.mel_pts ? 0 : a.mel_off[a.F - 1] + a.mel_len[a.F - 1];
    for (int i = tid; i < n_mw; i += 256) mw[i] = a.mel_w[i];
    for (int i = tid; i < LM_WIN; i += 256) win[i] = a.window[i];
    for (int i = tid; i < 1024; i += 256) {
        const float2 w = reinterpret_cast<const float2*>(a.twiddle)[i];
        tw[i] = cpx{w.x, w.y};
    }
    __syncthreads();

    cpx* A = bufs + wave * 1024;
    cpx* Bf = A + 512;
    const int sl = a.seq_len ? min(a.seq_len[b], a.T) : a.T;
    // per-lane filter descriptors (filters m = lane, lane + 64, ...) stay in registers for all frames of the block
    constexpr int MPL = LM_NMEL_MAX * 4 / 64;
    int f_st[MPL], f_ln[MPL], f_off[MPL];
    float f_mean[MPL], f_is[MPL], f_lo[MPL], f_c[MPL], f_hi[MPL];
    const float* pts = a.mel_pts ? a.mel_pts + (size_t)b * (a.F + 2) : nullptr;
#pragma unroll
    for (int i = 0; i < MPL; ++i) {
        const int m = lane + 64 * i;
        if (m < a.F) {
            f_mean[i] = a.mean[m]; f_is[i] = a.inv_std[m];
            if (pts) { f_lo[i] = pts[m]; f_c[i] = pts[m + 1]; f_hi[i] = pts[m + 2]; }
            else { f_st[i] = a.mel_start[m]; f_ln[i] = a.mel_len[m]; f_off[i] = a.mel_off[m]; }
        }
    }
    for (int fi = 0; fi < LM_FR / 4; ++fi) {
        const int fl = wave * (LM_FR / 4) + fi;
        const int t = t0 + fl;
        const float* x = samp + fl * LM_SHIFT;
        // pack windowed real frame (zero padded 960 -> 1024) as 512 complex
        if (fpos) {
            // warped framing: the windows of neighbouring frames start at arbitrary samples, so each wave reads its frame
            // straight from the clip (L2-resident: every sample is touched by ~3 frames)
            const long s = t < a.T ? (long)fpos[t] : (long)a.N;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int n = lane + r * 64;
                cpx v = cpx{0.f, 0.f};
                if (2 * n < LM_WIN) {
                    const long i0 = s + 2 * n, i1 = i0 + 1;
                    v = cpx{(i0 >= 0 && i0 < a.N ? wav[i0] : 0.f) * win[2 * n], (i1 >= 0 && i1 < a.N ? wav[i1] : 0.f) * win[2 * n + 1]};
                }
                A[n] = v;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int n = lane + r * 64;
                cpx v = cpx{0.f, 0.f};
                if (2 * n < LM_WIN) v = cpx{x[2 * n] * win[2 * n], x[2 * n + 1] * win[2 * n + 1]};
                A[n] = v;
            }
        }
        wave_lds_sync();
        fft512_pass<1>(A, Bf, tw, lane);
        wave_lds_sync();
        fft512_pass<8>(Bf, A, tw, lane);
        wave_lds_sync();
        fft512_pass<64>(A, Bf, tw, lane);
        wave_lds_sync();
        float* P = reinterpret_cast<float*>(A);                 // [513] power spectrum
        for (int k = lane; k <= 512; k += 64) P[k] = rfft1024_power(Bf, tw, k);
        wave_lds_sync();
#pragma unroll
        for (int i = 0; i < MPL; ++i) {
            const int m = lane + 64 * i;
            if (m >= a.F) break;
            float s = 0.f;
            if (pts) {
                s = mel_triangle(P, 513, f_lo[i], f_c[i], f_hi[i]);
            } else {
                const float* p = P + f_st[i];
                const float* w = mw + f_off[i];
                for (int j = 0; j < f_ln[i]; ++j) s = fmaf(p[j], w[j], s);
            }
            float v = (logf(s + a.eps) - f_mean[i]) * f_is[i];
            v = fminf(fmaxf(v, -a.clampv), a.clampv);
            tile[m * (LM_FR + 1) + fl] = (t < sl) ? v : 0.f;
        }
        wave_lds_sync();
    }
    __syncthreads();
    if (a.stats) logmel_tile_stats(tile, a.stats, a.F, tid);
    for (int i = tid; i < a.F * LM_FR; i += 256) {
        const int m = i / LM_FR, fl = i % LM_FR;
        if (t0 + fl < a.T) a.out[((size_t)b * a.F + m) * a.T + t0 + fl] = tile[m * (LM_FR + 1) + fl];
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
    const size_t lds = ((LM_FR - 1) * LM_SHIFT + LM_WIN + LM_WIN) * sizeof(float) + 1024 * sizeof(cpx) +
                       4 * 2 * 512 * sizeof(cpx) + (size_t)F * (LM_FR + 1) * sizeof(float) + LM_MW_MAX * sizeof(float);
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
