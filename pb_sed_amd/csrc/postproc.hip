// Ensemble post-processing on the GPU (bit-exact twins of the reference's numpy/scipy host code):
//   mean over models + sequence mask      pb_sed/models/base/inference.py:142-147
//   per-class / per-variant median filter pb_sed/filters.py:56-83 (scipy.signal.medfilt, zero padded)
//   boundaries filter (step filter fwd/bwd, cummax, min)   pb_sed/filters.py:112-135,
//                                                          pb_sed/models/base/inference.py:266-289
//   threshold -> change points -> event frame indices      sed_scores_eval scores_to_event_list
//                                                          (call site experiments/strong_label_crnn/inference.py:147-150)
// Rows are the flattened leading dims of a [B, (n,) K, T] score tensor; every kernel takes per-row
// parameters (filter length / threshold) so per-class and per-variant settings are one launch.
#include "common.h"

namespace pbsed {

// out[row][t] = ((s0 + s1) + ... ) / M * [t < seq_len[row / rows_per_clip]]   (numpy mean over axis 0 in f32)
struct MeanArgs { const float* s[16]; };
__global__ void ensemble_mean_mask_kernel(MeanArgs a, int M, float* __restrict__ out, const int* seq_len,
                                          int rows_per_clip, int R, int T) {
    const size_t total = (size_t)R * T;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int t = i % T, row = i / T;
        float s = a.s[0][i];
        for (int m = 1; m < M; ++m) s += a.s[m][i];
        s = s / (float)M;
        out[i] = (t < seq_len[row / rows_per_clip]) ? s : 0.f * s;     // s * mask, mask in {0,1}
    }
}

// Zero-padded running median of odd length n[row] (n == 1: copy).  One block per row, row staged in LDS as order-preserving
// integer keys.  Exact selection in O(32 n) per output instead of O(n^2): the median is the (h+1)-th smallest of the
// window's n values (zeros outside [0, T) included), found by bisecting on the key space - 32 counting passes over the
// window, no candidate loop, no divergence between the outputs of a wave (tested up to n = 301, the reference's tuning
// range, pb_sed/experiments/strong_label_crnn/tuning.py:64).  -0.0 is folded into +0.0 (scipy's sort treats them equal
// and the padding is +0.0).
__device__ __forceinline__ unsigned med_key(float v) {
    unsigned u = __float_as_uint(v + 0.f);                        // -0.0 + 0.0 = +0.0
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);            // monotone: a < b  <=>  key(a) < key(b)
}
__device__ __forceinline__ float med_unkey(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ __launch_bounds__(256) void medfilt_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                      const int* __restrict__ n_row, int R, int T) {
    extern __shared__ unsigned xk[];
    const int row = blockIdx.x;
    const float* x = in + (size_t)row * T;
    float* y = out + (size_t)row * T;
    const int n = n_row[row];
    if (n <= 1) {
        for (int t = threadIdx.x; t < T; t += blockDim.x) y[t] = x[t];
        return;
    }
    for (int t = threadIdx.x; t < T; t += blockDim.x) xk[t] = med_key(x[t]);
    __syncthreads();
    const int h = n / 2;
    const unsigned zero_key = 0x80000000u;
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        const int a0 = max(t - h, 0), a1 = min(t + h, T - 1);
        const int nzero = n - (a1 - a0 + 1);                    // padded zeros inside the window
        // smallest key K with #{window values <= K} >= h + 1
        unsigned lo = 0u, hi = 0xffffffffu;
        while (lo < hi) {
            const unsigned mid = lo + ((hi - lo) >> 1);
            int cnt = (zero_key <= mid) ? nzero : 0;
            for (int k = a0; k <= a1; ++k) cnt += xk[k] <= mid;
            if (cnt >= h + 1) hi = mid; else lo = mid + 1;
        }
        y[t] = med_unkey(lo);
    }
}

// boundariesfilt: F = stepfilt(x, n), Rv = stepfilt(flip(x), n) in f64 (n == 0: identity), then
// out = min(cummax(F), flip(cummax(Rv))) cast to f32.  One block per row; the scans are sequential.
__global__ __launch_bounds__(256) void boundariesfilt_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                             double* __restrict__ out64, const int* __restrict__ n_row,
                                                             int R, int T) {
    extern __shared__ double sh[];
    double* F = sh;          // [T]
    double* Rv = sh + T;     // [T]  (indexed in flipped time)
    const int row = blockIdx.x;
    const float* x = in + (size_t)row * T;
    const int n = n_row[row], h = n / 2;
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        if (n > 0) {
            const double w = 1.0 / (double)h;
            double f = 0.0, r = 0.0;
            for (int k = 0; k < n; ++k) {                        // valid correlation of the padded row
                const int i = t + k - h;
                const double c = (k < h) ? -w : w;
                const double xf = (i >= 0 && i < T) ? (double)x[i] : 0.0;
                const double xr = (i >= 0 && i < T) ? (double)x[T - 1 - i] : 0.0;
                f += xf * c;
                r += xr * c;
            }
            F[t] = f; Rv[t] = r;
        } else {
            F[t] = (double)x[t]; Rv[t] = (double)x[T - 1 - t];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) for (int t = 1; t < T; ++t) F[t] = fmax(F[t], F[t - 1]);
    if (threadIdx.x == 64) for (int t = 1; t < T; ++t) Rv[t] = fmax(Rv[t], Rv[t - 1]);
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        const double v = fmin(F[t], Rv[T - 1 - t]);
        if (out) out[(size_t)row * T + t] = (float)v;
        if (out64) out64[(size_t)row * T + t] = v;
    }
}

// det = score > thr[row] (strict) on frames < len[row]; rising edges -> onsets, falling -> offsets.
// events[row][e] = (onset_frame, offset_frame), counts[row] = number of events (<= max_events).
__global__ void event_frames_kernel(const float* __restrict__ scores, const float* __restrict__ thr,
                                    const int* __restrict__ len, int* __restrict__ events, int* __restrict__ counts,
                                    int R, int T, int max_events) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= R) return;
    const float* s = scores + (size_t)row * T;
    const float th = thr[row];
    const int L = min(len[row], T);
    int n = 0, on = -1;
    bool prev = false;
    for (int t = 0; t < L; ++t) {
        const bool d = s[t] > th;
        if (d && !prev) on = t;
        if (!d && prev) {
            if (n < max_events) { events[((size_t)row * max_events + n) * 2] = on; events[((size_t)row * max_events + n) * 2 + 1] = t; }
            ++n;
        }
        prev = d;
    }
    if (prev) {
        if (n < max_events) { events[((size_t)row * max_events + n) * 2] = on; events[((size_t)row * max_events + n) * 2 + 1] = L; }
        ++n;
    }
    counts[row] = n;
}

}  // namespace pbsed

using namespace pbsed;

extern "C" {

int pbsed_ensemble_mean_mask(const float* const* scores, int n_models, float* out, const int* seq_len,
                             int rows_per_clip, int R, int T, void* stream) {
    if (n_models < 1 || n_models > 16) { set_error("ensemble_mean_mask: 1..16 models"); return PBSED_E_ARG; }
    MeanArgs a{};
    for (int m = 0; m < n_models; ++m) a.s[m] = scores[m];
    const size_t total = (size_t)R * T;
    const int nb = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    hipLaunchKernelGGL(ensemble_mean_mask_kernel, dim3(nb ? nb : 1), dim3(256), 0, (hipStream_t)stream, a, n_models, out,
                       seq_len, rows_per_clip, R, T);
    return check_launch("ensemble_mean_mask");
}

int pbsed_medfilt(const float* in, float* out, const int* n_row, int R, int T, void* stream) {
    if ((size_t)T * sizeof(float) > 160 * 1024) { set_error("medfilt: T too long for LDS staging"); return PBSED_E_ARG; }
    hipLaunchKernelGGL(medfilt_kernel, dim3(R), dim3(256), T * sizeof(float), (hipStream_t)stream, in, out, n_row, R, T);
    return check_launch("medfilt");
}

int pbsed_boundariesfilt(const float* in, float* out, double* out64, const int* n_row, int R, int T, void* stream) {
    if ((size_t)T * 2 * sizeof(double) > 64 * 1024) { set_error("boundariesfilt: T too long"); return PBSED_E_ARG; }
    hipLaunchKernelGGL(boundariesfilt_kernel, dim3(R), dim3(256), 2 * T * sizeof(double), (hipStream_t)stream, in, out,
                       out64, n_row, R, T);
    return check_launch("boundariesfilt");
}

int pbsed_event_frames(const float* scores, const float* thr, const int* len, int* events, int* counts, int R, int T,
                       int max_events, void* stream) {
    hipLaunchKernelGGL(event_frames_kernel, dim3((R + 63) / 64), dim3(64), 0, (hipStream_t)stream, scores, thr, len,
                       events, counts, R, T, max_events);
    return check_launch("event_frames");
}

}  // extern "C"
