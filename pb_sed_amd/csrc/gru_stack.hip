// Multi-layer unidirectional GRU stacks as a layer-wavefront of per-time-step launches (gfx950).
//
// Launch s runs scan step s of layer 0 and scan step s-1 of layer 1 (... s-l of layer l) for every
// chain (FBCRNN: forward GRU + time-reversed GRU), so a 2-layer x 2-chain recurrence needs T+1
// dependent launches instead of 2T, and each launch has 2x the blocks.  Upper layers compute their
// input projection W_ih x_t in the step (x_t = lower layer's output of the previous launch); BPTT
// mirrors it: lower layers compute dy_t = dgi_upper,t W_ih_upper in the step.  Matmuls run on fp32
// MFMA with operands read straight from L2 as 16-byte rows (consistent K permutation for A and B);
// H is a template parameter so every operand load of a step is issued before the first MFMA.
//
// Reference op site: torch.nn.GRU(num_layers=2) inside padertorch's GRU wrapper,
// pb_sed/models/weak_label/crnn.py:61-67,338-340 (config training.py:243-248).
#include "common.h"
#include "pbsed_internal.h"

namespace pbsed {

constexpr int GRU_MAX_LAYERS = 4;
constexpr int GRU_MAX_CHAINS = 2;

struct GruStackLayer {
    const float* gi;      // layer 0: [T][B][3H] (W_ih x + b_ih precomputed);  else null
    const float* w_ih;    // layer > 0: [3H][H]          (bwd: W_ih of the layer ABOVE, transposed [H][3H])
    const float* b_ih;    // layer > 0: [3H]
    const float* w_hh;    // fwd: [3H][H]                (bwd: W_hh^T [H][3H])
    const float* b_hh;    // [3H]
    float* hs;            // [T][B][H] outputs (0 past seq_len)
    float* save;          // [T][B][4][H]: r, z, n, W_hn h + b_hn
    const float* dy;      // bwd, top layer: [T][B][H] grad wrt outputs
    float* dgi;           // bwd: [T][B][3H]
    float* dgh;           // bwd: [T][B][3H]
    float* dhz;           // bwd: [T][B][H]
};

struct GruStackArgs {
    GruStackLayer lc[GRU_MAX_CHAINS][GRU_MAX_LAYERS];
    int reverse[GRU_MAX_CHAINS];
    const int* seq_len;
    int B, T, nchains, nlayers, launch;
};

// acc[g] += W[g*H + j0 + lr][k..] * v[b0 + lr][k..] over this wave's quarter of K = KB*64.
template <int KB, int NG>
__device__ __forceinline__ void mm_rows(f32x4 (&acc)[NG], const float* __restrict__ w, int wstride, int gstride,
                                        const float* __restrict__ v, int vstride, bool vvalid, int wave, int lq) {
    float4 vv[KB], wv[KB][NG];
#pragma unroll
    for (int i = 0; i < KB; ++i) {
        const int k = (wave * KB + i) * 16 + lq * 4;
        vv[i] = vvalid ? *reinterpret_cast<const float4*>(v + k) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int g = 0; g < NG; ++g) wv[i][g] = *reinterpret_cast<const float4*>(w + (size_t)g * gstride + k);
    }
#pragma unroll
    for (int i = 0; i < KB; ++i)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            acc[g] = mfma16(wv[i][g].x, vv[i].x, acc[g]);
            acc[g] = mfma16(wv[i][g].y, vv[i].y, acc[g]);
            acc[g] = mfma16(wv[i][g].z, vv[i].z, acc[g]);
            acc[g] = mfma16(wv[i][g].w, vv[i].w, acc[g]);
        }
}

// ------------------------------------------------------------------------------------------ forward
// grid (H/16, ceil(B/16), nchains*nlayers), 256 threads; H = KB*64.
template <int KB, int NW>
__global__ __launch_bounds__(NW * 64) void gru_stack_fwd_kernel(GruStackArgs a) {
    constexpr int H = KB * NW * 16;
    __shared__ float red[NW][4][64][4];
    const int chain = blockIdx.z % a.nchains, layer = blockIdx.z / a.nchains;
    const int step = a.launch - layer;
    if (step < 0 || step >= a.T) return;
    const GruStackLayer& L = a.lc[chain][layer];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane >> 4, lr = lane & 15;
    const int j0 = blockIdx.x * 16, b0 = blockIdx.y * 16, B = a.B;
    const bool rev = a.reverse[chain] != 0;
    const int t = rev ? a.T - 1 - step : step;
    const int tp = rev ? t + 1 : t - 1;
    const bool has_prev = step > 0;
    // epilogue operands first: their HBM latency hides behind the matmuls
    const int u = tid & 15, bb = tid >> 4, b = b0 + bb, j = j0 + u;
    const bool bv = tid < 256 && b < B;
    float gi_r = 0.f, gi_z = 0.f, gi_n = 0.f, hp = 0.f;
    if (bv) {
        if (layer == 0) {
            const float* gi = L.gi + ((size_t)t * B + b) * 3 * H;
            gi_r = gi[j]; gi_z = gi[H + j]; gi_n = gi[2 * H + j];
        } else {
            gi_r = L.b_ih[j]; gi_z = L.b_ih[H + j]; gi_n = L.b_ih[2 * H + j];
        }
        if (has_prev) hp = L.hs[((size_t)tp * B + b) * H + j];
    }
    const float bh_r = L.b_hh[j0 + u], bh_z = L.b_hh[H + j0 + u], bh_n = L.b_hh[2 * H + j0 + u];
    const int sl = bv ? a.seq_len[b] : 0;

    f32x4 acc[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    f32x4 acci[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const bool rowv = (b0 + lr) < B;
    if (has_prev)
        mm_rows<KB, 3>(acc, L.w_hh + (size_t)(j0 + lr) * H, H, H * H, L.hs + ((size_t)tp * B + b0 + lr) * H, H, rowv,
                       wave, lq);
    if (layer > 0) {
        const float* x = a.lc[chain][layer - 1].hs + ((size_t)t * B + b0 + lr) * H;
        mm_rows<KB, 3>(acci, L.w_ih + (size_t)(j0 + lr) * H, H, H * H, x, H, rowv, wave, lq);
    }
    // r and z only ever appear as sums; n keeps its input / hidden parts apart
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[wave][0][lane][r] = acc[0][r] + acci[0][r];
        red[wave][1][lane][r] = acc[1][r] + acci[1][r];
        red[wave][2][lane][r] = acc[2][r];
        red[wave][3][lane][r] = acci[2][r];
    }
    __syncthreads();
    if (!bv) return;
    const int src = (u >> 2) * 16 + bb, reg = u & 3;
    float s[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        s[g] = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) s[g] += red[w][g][src][reg];
    }
    const float ghn = s[2] + bh_n;
    const float r = 1.f / (1.f + expf(-(gi_r + s[0] + bh_r)));
    const float z = 1.f / (1.f + expf(-(gi_z + s[1] + bh_z)));
    const float n = tanhf(gi_n + s[3] + r * ghn);
    const float h = (1.f - z) * n + z * hp;
    const size_t tb = (size_t)t * B + b;
    L.hs[tb * H + j] = (t < sl) ? h : 0.f;
    if (L.save) {
        float* sv = L.save + tb * 4 * H;
        sv[j] = r; sv[H + j] = z; sv[2 * H + j] = n; sv[3 * H + j] = ghn;
    }
}

// ------------------------------------------------------------------------------------------ backward
// Launch s: top layer handles reversed scan index s, layer l handles s - (nlayers-1-l).
// w_hh holds W_hh^T [H][3H]; for l < top, w_ih holds (W_ih of layer l+1)^T [H][3H].
template <int KB, int NW>
__global__ __launch_bounds__(NW * 64) void gru_stack_bwd_kernel(GruStackArgs a) {
    constexpr int H = KB * NW * 16, G = 3 * H, KB3 = 3 * KB;
    __shared__ float red[NW][2][64][4];
    const int chain = blockIdx.z % a.nchains, layer = blockIdx.z / a.nchains;
    const int top = a.nlayers - 1;
    const int bstep = a.launch - (top - layer);
    if (bstep < 0 || bstep >= a.T) return;
    const GruStackLayer& L = a.lc[chain][layer];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane >> 4, lr = lane & 15;
    const int j0 = blockIdx.x * 16, b0 = blockIdx.y * 16, B = a.B;
    const bool rev = a.reverse[chain] != 0;
    const int s = a.T - 1 - bstep;                     // forward scan index handled now
    const int t = rev ? a.T - 1 - s : s;
    const int tn = rev ? t - 1 : t + 1;                // next scan step (done by the previous launch)
    const int tp = rev ? t + 1 : t - 1;
    const bool has_next = bstep > 0, has_prev = s > 0;
    const int u = tid & 15, bb = tid >> 4, b = b0 + bb, j = j0 + u;
    const bool bv = tid < 256 && b < B;
    const size_t tb = (size_t)t * B + b;
    float r = 0.f, z = 0.f, n = 0.f, ghn = 0.f, hp = 0.f, dyv = 0.f, dhzn = 0.f;
    int sl = 0;
    if (bv) {
        sl = a.seq_len[b];
        const float* sv = L.save + tb * 4 * H;
        r = sv[j]; z = sv[H + j]; n = sv[2 * H + j]; ghn = sv[3 * H + j];
        if (has_prev) hp = L.hs[((size_t)tp * B + b) * H + j];
        if (layer == top) dyv = L.dy[tb * H + j];
        if (has_next) dhzn = L.dhz[((size_t)tn * B + b) * H + j];
    }
    f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}}, accy[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
    const bool rowv = (b0 + lr) < B;
    if (has_next)
        mm_rows<KB3, 1>(acc, L.w_hh + (size_t)(j0 + lr) * G, G, 0, L.dgh + ((size_t)tn * B + b0 + lr) * G, G, rowv,
                        wave, lq);
    if (layer < top)
        mm_rows<KB3, 1>(accy, L.w_ih + (size_t)(j0 + lr) * G, G, 0,
                        a.lc[chain][layer + 1].dgi + ((size_t)t * B + b0 + lr) * G, G, rowv, wave, lq);
#pragma unroll
    for (int q = 0; q < 4; ++q) { red[wave][0][lane][q] = acc[0][q]; red[wave][1][lane][q] = accy[0][q]; }
    __syncthreads();
    if (!bv) return;
    const int src = (u >> 2) * 16 + bb, reg = u & 3;
    float carry = 0.f, dylow = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) { carry += red[w][0][src][reg]; dylow += red[w][1][src][reg]; }
    float dr = 0.f, dz = 0.f, dn = 0.f, dnr = 0.f, dhzv = 0.f;
    if (t < sl) {
        const float dh = (layer == top ? dyv : dylow) + (has_next ? carry + dhzn : 0.f);
        dn = dh * (1.f - z) * (1.f - n * n);
        dz = dh * (hp - n) * z * (1.f - z);
        dr = dn * ghn * r * (1.f - r);
        dnr = dn * r;
        dhzv = dh * z;
    }
    float* dgi = L.dgi + tb * G;
    float* dgh = L.dgh + tb * G;
    dgi[j] = dr; dgi[H + j] = dz; dgi[2 * H + j] = dn;
    dgh[j] = dr; dgh[H + j] = dz; dgh[2 * H + j] = dnr;
    L.dhz[tb * H + j] = dhzv;
}

template <int KB, int NW>
static void launch_stack(bool bwd, GruStackArgs& a, dim3 grid, hipStream_t s) {
    const int nl = a.T + a.nlayers - 1;
    for (int i = 0; i < nl; ++i) {
        a.launch = i;
        if (bwd) hipLaunchKernelGGL((gru_stack_bwd_kernel<KB, NW>), grid, dim3(NW * 64), 0, s, a);
        else hipLaunchKernelGGL((gru_stack_fwd_kernel<KB, NW>), grid, dim3(NW * 64), 0, s, a);
    }
}

}  // namespace pbsed

using namespace pbsed;

static int stack_check(int nchains, int nlayers, int B, int H, int T) {
    if (nchains < 1 || nchains > GRU_MAX_CHAINS || nlayers < 1 || nlayers > GRU_MAX_LAYERS || B < 1 || T < 1 ||
        !(H == 64 || H == 128 || H == 256 || H == 512)) {
        set_error("gru_stack: unsupported nchains=%d nlayers=%d B=%d H=%d T=%d (H in {64,128,256,512})", nchains,
                  nlayers, B, H, T);
        return PBSED_E_ARG;
    }
    return PBSED_OK;
}

#define DISPATCH_KB(H, BWD, a, grid, s)                            \
    switch (H) {                                                   \
        case 64: launch_stack<1, 4>(BWD, a, grid, s); break;       \
        case 128: launch_stack<1, 8>(BWD, a, grid, s); break;      \
        case 256: launch_stack<2, 8>(BWD, a, grid, s); break;      \
        default: launch_stack<4, 8>(BWD, a, grid, s); break;       \
    }

extern "C" {

// Forward scan of nchains independent UNIDIRECTIONAL stacks of nlayers GRU layers (hidden = input = H above
// layer 0).  Pointer tables are HOST arrays indexed [chain*nlayers + layer] of device pointers:
// gi0[chain] [T][B][3H]; w_ih/b_ih (layer > 0; entries for layer 0 ignored); w_hh [3H][H]; b_hh; hs; save.
int pbsed_gru_stack_fwd(int nchains, int nlayers, const float* const* gi0, const float* const* w_ih,
                        const float* const* b_ih, const float* const* w_hh, const float* const* b_hh,
                        float* const* hs, float* const* save, const int* reverse, const int* seq_len, int B, int H,
                        int T, void* stream) {
    if (int e = stack_check(nchains, nlayers, B, H, T)) return e;
    GruStackArgs a{};
    for (int c = 0; c < nchains; ++c) {
        a.reverse[c] = reverse[c];
        for (int l = 0; l < nlayers; ++l) {
            GruStackLayer& L = a.lc[c][l];
            const int i = c * nlayers + l;
            L.gi = l == 0 ? gi0[c] : nullptr;
            L.w_ih = l > 0 ? w_ih[i] : nullptr; L.b_ih = l > 0 ? b_ih[i] : nullptr;
            L.w_hh = w_hh[i]; L.b_hh = b_hh[i]; L.hs = hs[i]; L.save = save ? save[i] : nullptr;
        }
    }
    a.seq_len = seq_len; a.B = B; a.T = T; a.nchains = nchains; a.nlayers = nlayers;
    dim3 grid(H / 16, (B + 15) / 16, nchains * nlayers);
    DISPATCH_KB(H, false, a, grid, (hipStream_t)stream);
    return check_launch("gru_stack_fwd");
}

// BPTT of the same stacks.  w_hh_t[i] = W_hh^T [H][3H]; w_ih_up_t[i] = (W_ih of layer l+1)^T [H][3H] (ignored
// for the top layer); dy_top[chain] [T][B][H]; outputs dgi/dgh [T][B][3H], scratch dhz [T][B][H].
int pbsed_gru_stack_bwd(int nchains, int nlayers, const float* const* w_hh_t, const float* const* w_ih_up_t,
                        const float* const* hs, const float* const* save, const float* const* dy_top,
                        float* const* dgi, float* const* dgh, float* const* dhz, const int* reverse,
                        const int* seq_len, int B, int H, int T, void* stream) {
    if (int e = stack_check(nchains, nlayers, B, H, T)) return e;
    GruStackArgs a{};
    for (int c = 0; c < nchains; ++c) {
        a.reverse[c] = reverse[c];
        for (int l = 0; l < nlayers; ++l) {
            GruStackLayer& L = a.lc[c][l];
            const int i = c * nlayers + l;
            L.w_hh = w_hh_t[i]; L.w_ih = l < nlayers - 1 ? w_ih_up_t[i] : nullptr;
            L.hs = const_cast<float*>(hs[i]); L.save = const_cast<float*>(save[i]);
            L.dy = l == nlayers - 1 ? dy_top[c] : nullptr;
            L.dgi = dgi[i]; L.dgh = dgh[i]; L.dhz = dhz[i];
        }
    }
    a.seq_len = seq_len; a.B = B; a.T = T; a.nchains = nchains; a.nlayers = nlayers;
    dim3 grid(H / 16, (B + 15) / 16, nchains * nlayers);
    DISPATCH_KB(H, true, a, grid, (hipStream_t)stream);
    return check_launch("gru_stack_bwd");
}

}  // extern "C"
