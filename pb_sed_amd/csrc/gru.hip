// GRU recurrence (forward scan + BPTT) for gfx950: one launch per time step, every launch covering
// all independent chains (the FBCRNN's forward and time-reversed GRUs, or the BiCRNN's two
// directions).  A same-stream kernel boundary (~1.5 us) is cheaper on MI355X than a software grid
// barrier (~4 us, MI355X_MICROARCH.md price list), so the 500-step loop is a launch sequence, not a
// persistent kernel; the recurrent matmul h_{t-1} W_hh^T runs on fp32 MFMA with W_hh/h fragments
// loaded straight from L2 as 16-byte rows (K is permuted consistently for A and B, no LDS staging).
//
// Reference op sites: torch.nn.GRU inside padertorch's GRU wrapper, called at
// pb_sed/models/weak_label/crnn.py:61-67 (rnn_fwd / rnn_bwd) and pb_sed/models/strong_label/crnn.py:92;
// packed-sequence semantics (state frozen / output 0 past seq_len), reverse = per-sequence time flip.
//
// Layout: time-major [T][B][*] for everything the scan touches (coalesced per step); the conv
// kernels' [B,C,T] tensors are converted by transpose kernels (also here).
#include "common.h"
#include "pbsed_internal.h"

namespace pbsed {

struct GruChain {
    const float* gi;      // [T][B][3H]  W_ih x_t + b_ih
    const float* w;       // fwd: W_hh [3H][H];  bwd: W_hh^T [H][3H]
    const float* b_hh;    // [3H]
    float* hs;            // [T][B][H] outputs (0 past seq_len)
    float* save;          // [T][B][4][H]: r, z, n, (W_hn h + b_hn)
    // backward only
    const float* dy;      // [T][B][H] grad wrt outputs
    float* dgi;           // [T][B][3H]
    float* dgh;           // [T][B][3H]
    float* dhz;           // [T][B][H]  dh_total * z
    int reverse;          // 1: scan runs t = T-1 .. 0
};

struct GruStepArgs {
    GruChain c[2];
    const int* seq_len;   // [B]
    int B, H, T, step;    // step = scan index 0..T-1
};

// ---------------------------------------------------------------------------------- forward step
// grid (H/16, ceil(B/16), nchains), 256 threads.  Block: 16 hidden units x 16 batch rows.
__global__ __launch_bounds__(256) void gru_step_fwd_kernel(GruStepArgs a) {
    __shared__ float red[4][3][64][4];
    const GruChain& ch = a.c[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane >> 4, lr = lane & 15;
    const int j0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    const int H = a.H, B = a.B;
    const int t = ch.reverse ? a.T - 1 - a.step : a.step;
    const int tp = ch.reverse ? t + 1 : t - 1;          // previous step in scan order
    const bool has_prev = a.step > 0;
    const float* hprev = has_prev ? ch.hs + (size_t)tp * B * H : nullptr;

    f32x4 acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (has_prev) {
        const int kq = H / 4;                              // K range per wave
        const int bb = b0 + lr;
        for (int kb = wave * kq; kb < (wave + 1) * kq; kb += 16) {
            const int k = kb + lq * 4;
            float4 hv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bb < B) hv = *reinterpret_cast<const float4*>(hprev + (size_t)bb * H + k);
            float4 wv[3];
#pragma unroll
            for (int g = 0; g < 3; ++g)
                wv[g] = *reinterpret_cast<const float4*>(ch.w + (size_t)(g * H + j0 + lr) * H + k);
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                acc[g] = mfma16(wv[g].x, hv.x, acc[g]);
                acc[g] = mfma16(wv[g].y, hv.y, acc[g]);
                acc[g] = mfma16(wv[g].z, hv.z, acc[g]);
                acc[g] = mfma16(wv[g].w, hv.w, acc[g]);
            }
        }
    }
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][g][lane][r] = acc[g][r];
    __syncthreads();
    // thread -> (batch row bb, unit u); D[row = unit][col = batch] sits in lane (u>>2)*16+bb, reg u&3
    const int u = tid & 15, bb = tid >> 4;
    const int b = b0 + bb, j = j0 + u;
    if (b >= B) return;
    const int src = (u >> 2) * 16 + bb, reg = u & 3;
    float gh[3];
#pragma unroll
    for (int g = 0; g < 3; ++g)
        gh[g] = red[0][g][src][reg] + red[1][g][src][reg] + red[2][g][src][reg] + red[3][g][src][reg]
                + ch.b_hh[g * H + j];
    const float* gi = ch.gi + ((size_t)t * B + b) * 3 * H;
    const float r = 1.f / (1.f + expf(-(gi[j] + gh[0])));
    const float z = 1.f / (1.f + expf(-(gi[H + j] + gh[1])));
    const float n = tanhf(gi[2 * H + j] + r * gh[2]);
    const float hp = has_prev ? hprev[(size_t)b * H + j] : 0.f;
    const bool active = t < a.seq_len[b];
    const float h = (1.f - z) * n + z * hp;
    ch.hs[((size_t)t * B + b) * H + j] = active ? h : 0.f;
    if (ch.save) {
        float* sv = ch.save + ((size_t)t * B + b) * 4 * H;
        sv[j] = r; sv[H + j] = z; sv[2 * H + j] = n; sv[3 * H + j] = gh[2];
    }
}

// --------------------------------------------------------------------------------- backward step
// Runs the scan backwards: launch `step` handles scan index s = T-1-step.  carry into s comes from
// scan index s+1 (processed by the previous launch): dh_{s+1}*z_{s+1} + dgh_{s+1} W_hh.
__global__ __launch_bounds__(256) void gru_step_bwd_kernel(GruStepArgs a) {
    __shared__ float red[4][64][4];
    const GruChain& ch = a.c[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane >> 4, lr = lane & 15;
    const int j0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    const int H = a.H, B = a.B, G = 3 * a.H;
    const int s = a.T - 1 - a.step;
    const int t = ch.reverse ? a.T - 1 - s : s;
    const int tn = ch.reverse ? t - 1 : t + 1;          // next step in scan order (already processed)
    const int tp = ch.reverse ? t + 1 : t - 1;          // previous step in scan order
    const bool has_next = a.step > 0;
    const bool has_prev = s > 0;

    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (has_next) {
        const float* dghn = ch.dgh + (size_t)tn * B * G;
        const int kq = G / 4;
        const int bb = b0 + lr;
        for (int kb = wave * kq; kb < (wave + 1) * kq; kb += 16) {
            const int k = kb + lq * 4;
            float4 dv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bb < B) dv = *reinterpret_cast<const float4*>(dghn + (size_t)bb * G + k);
            const float4 wv = *reinterpret_cast<const float4*>(ch.w + (size_t)(j0 + lr) * G + k);
            acc = mfma16(wv.x, dv.x, acc);
            acc = mfma16(wv.y, dv.y, acc);
            acc = mfma16(wv.z, dv.z, acc);
            acc = mfma16(wv.w, dv.w, acc);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][lane][r] = acc[r];
    __syncthreads();
    const int u = tid & 15, bb = tid >> 4;
    const int b = b0 + bb, j = j0 + u;
    if (b >= B) return;
    const int src = (u >> 2) * 16 + bb, reg = u & 3;
    const size_t tb = (size_t)t * B + b;
    const bool active = t < a.seq_len[b];
    float dr = 0.f, dz = 0.f, dn = 0.f, dnr = 0.f, dhzv = 0.f;
    if (active) {
        float carry = 0.f;
        if (has_next)
            carry = red[0][src][reg] + red[1][src][reg] + red[2][src][reg] + red[3][src][reg]
                    + ch.dhz[((size_t)tn * B + b) * H + j];
        const float dh = ch.dy[tb * H + j] + carry;
        const float* sv = ch.save + tb * 4 * H;
        const float r = sv[j], z = sv[H + j], n = sv[2 * H + j], ghn = sv[3 * H + j];
        const float hp = has_prev ? ch.hs[((size_t)tp * B + b) * H + j] : 0.f;
        dn = dh * (1.f - z) * (1.f - n * n);
        dz = dh * (hp - n) * z * (1.f - z);
        dr = dn * ghn * r * (1.f - r);
        dnr = dn * r;
        dhzv = dh * z;
    }
    float* dgi = ch.dgi + tb * G;
    float* dgh = ch.dgh + tb * G;
    dgi[j] = dr; dgi[H + j] = dz; dgi[2 * H + j] = dn;
    dgh[j] = dr; dgh[H + j] = dz; dgh[2 * H + j] = dnr;
    ch.dhz[tb * H + j] = dhzv;
}

// ------------------------------------------------------------------------------------ transposes
// [B, C, T] <-> [T, B, C] through a 32x33 LDS tile (both sides coalesced).  `shift`: the time-major
// side is read/written at t + shift (zero fill out of range) - used to build h_{t-1} tensors.
__global__ __launch_bounds__(256) void bct_to_tbc_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                         int B, int C, int T) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, t = t0 + tx;
        tile[i][tx] = (c < C && t < T) ? src[((size_t)b * C + c) * T + t] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i, c = c0 + tx;
        if (t < T && c < C) dst[((size_t)t * B + b) * C + c] = tile[tx][i];
    }
}

__global__ __launch_bounds__(256) void tbc_to_bct_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                         int B, int C, int T, int shift) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int t = t0 + i + shift, c = c0 + tx;
        tile[i][tx] = (t >= 0 && t < T && t0 + i < T && c < C) ? src[((size_t)t * B + b) * C + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, t = t0 + tx;
        if (c < C && t < T) dst[((size_t)b * C + c) * T + t] = tile[tx][i];
    }
}

__global__ void transpose2d_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int C) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8)
        tile[i][tx] = (r0 + i < R && c0 + tx < C) ? src[(size_t)(r0 + i) * C + c0 + tx] : 0.f;
    __syncthreads();
    for (int i = ty; i < 32; i += 8)
        if (c0 + i < C && r0 + tx < R) dst[(size_t)(c0 + i) * R + r0 + tx] = tile[tx][i];
}

}  // namespace pbsed

using namespace pbsed;

extern "C" {

int pbsed_bct_to_tbc(const float* src, float* dst, int B, int C, int T, void* stream) {
    dim3 grid((T + 31) / 32, (C + 31) / 32, B);
    hipLaunchKernelGGL(bct_to_tbc_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, B, C, T);
    return check_launch("bct_to_tbc");
}

int pbsed_tbc_to_bct(const float* src, float* dst, int B, int C, int T, int shift, void* stream) {
    dim3 grid((T + 31) / 32, (C + 31) / 32, B);
    hipLaunchKernelGGL(tbc_to_bct_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, B, C, T, shift);
    return check_launch("tbc_to_bct");
}

int pbsed_transpose2d(const float* src, float* dst, int R, int C, void* stream) {
    dim3 grid((C + 31) / 32, (R + 31) / 32);
    hipLaunchKernelGGL(transpose2d_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, R, C);
    return check_launch("transpose2d");
}

static int gru_check(int nchains, int B, int H, int T) {
    if (nchains < 1 || nchains > 2 || B < 1 || T < 1 || H < 16 || (H % 16) || ((H / 4) % 16)) {
        set_error("gru: unsupported shape nchains=%d B=%d H=%d T=%d (H must be a multiple of 64)", nchains, B, H, T);
        return PBSED_E_ARG;
    }
    return PBSED_OK;
}

// Forward scan of `nchains` independent single-layer GRU chains over T steps.
// gi[c]: [T][B][3H]; w_hh[c]: [3H][H]; b_hh[c]: [3H]; hs[c]: [T][B][H]; save[c]: [T][B][4][H] or null.
int pbsed_gru_scan_fwd(int nchains, const float* const* gi, const float* const* w_hh,
                       const float* const* b_hh, float* const* hs, float* const* save,
                       const int* reverse, const int* seq_len_dev, int B, int H, int T, void* stream) {
    if (int e = gru_check(nchains, B, H, T)) return e;
    GruStepArgs a{};
    for (int c = 0; c < nchains; ++c) {
        a.c[c].gi = gi[c]; a.c[c].w = w_hh[c]; a.c[c].b_hh = b_hh[c]; a.c[c].hs = hs[c];
        a.c[c].save = save ? save[c] : nullptr; a.c[c].reverse = reverse[c];
    }
    a.seq_len = seq_len_dev; a.B = B; a.H = H; a.T = T;
    dim3 grid(H / 16, (B + 15) / 16, nchains);
    for (int s = 0; s < T; ++s) {
        a.step = s;
        hipLaunchKernelGGL(gru_step_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    }
    return check_launch("gru_scan_fwd");
}

// BPTT.  w_hh_t[c]: W_hh^T [H][3H]; dy[c]: [T][B][H]; outputs dgi[c], dgh[c]: [T][B][3H];
// dhz[c]: [T][B][H] scratch.
int pbsed_gru_scan_bwd(int nchains, const float* const* w_hh_t, const float* const* hs,
                       const float* const* save, const float* const* dy, float* const* dgi,
                       float* const* dgh, float* const* dhz, const int* reverse,
                       const int* seq_len_dev, int B, int H, int T, void* stream) {
    if (int e = gru_check(nchains, B, H, T)) return e;
    if (((3 * H / 4) % 16)) { set_error("gru bwd: 3H/4 must be a multiple of 16"); return PBSED_E_ARG; }
    GruStepArgs a{};
    for (int c = 0; c < nchains; ++c) {
        a.c[c].w = w_hh_t[c]; a.c[c].hs = const_cast<float*>(hs[c]);
        a.c[c].save = const_cast<float*>(save[c]); a.c[c].dy = dy[c];
        a.c[c].dgi = dgi[c]; a.c[c].dgh = dgh[c]; a.c[c].dhz = dhz[c]; a.c[c].reverse = reverse[c];
    }
    a.seq_len = seq_len_dev; a.B = B; a.H = H; a.T = T;
    dim3 grid(H / 16, (B + 15) / 16, nchains);
    for (int s = 0; s < T; ++s) {
        a.step = s;
        hipLaunchKernelGGL(gru_step_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    }
    return check_launch("gru_scan_bwd");
}

}  // extern "C"
