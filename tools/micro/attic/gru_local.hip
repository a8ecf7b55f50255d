// One-layer GRU scans WITHOUT an inter-workgroup exchange, for the bf16 training mode (gfx950).
//
// Op site: the BiGRU layers of the tag-conditioned strong-label CRNN (pb_sed/models/strong_label/crnn.py:88-93: torch.nn.GRU,
// bidirectional, one launch per layer with both directions as chains) in BASELINE.json configs[2] (bf16 operands).
//
// The persistent scans of gru_stack.hip spread one recurrence over 16 workgroups because its fp32-class product (bf16x3:
// six part products) needs 16 CUs' MFMA time per step - and pay an L2 / fabric hand-off per time step for it (0.8 of a
// step's 1.4 .. 2.3 us).  With ONE bf16 part per operand the whole recurrent matrix of a direction fits ONE CU:
// W_hh [768 x 256] bf16 = 384 KB = 48 MFMA fragments per wave of an 8-wave block (36 in registers, 12 in LDS), the state
// travels through LDS (16 rows x 256 units bf16, double-buffered, one block barrier per step) and the step is bound by the
// CU's own MFMA time: 16 x 768 x 256 MACs = 384 v_mfma_f32_16x16x32_bf16 = 1 536 clocks per SIMD.
//
// Block = (chain, 16-row batch tile), 512 threads = 8 waves; wave w owns hidden units 32 w .. 32 w + 31 of all three gates:
//   D[m = unit 4 lq + r of a 16-unit tile][n = batch row lr] += A[m][k] B[k][n],  A = W_hh rows (bf16, rounded to nearest even
//   once), B = h_{t-1} (bf16, rounded when it is written to LDS), fp32 accumulation, fp32 state and gate arithmetic in the lane
//   that owns (4 consecutive units, one batch row): gi / hs / the saved factors move as 16-byte accesses.
// Same results contract as gru_granule_fwd_body<.., XS = 1> (operands of the recurrent product rounded to bf16, everything
// else fp32; h = 0 past the sequence; save = the five factors BPTT multiplies dh_t with), without the tagged-LSB truncation
// of the exchanged state (there is no exchange).
#include <cstdlib>

#include "common.h"
#include "pbsed_internal.h"

namespace pbsed {

namespace {

constexpr int GL_H = 256, GL_G = 3 * GL_H;
#ifndef GL_NW_
#define GL_NW_ 8
#endif
#ifndef GL_DBG
#define GL_DBG 0            // ablation switches (tools/micro/gru_local_bench.py; never set in the product build): 1 no gi loads, 2 no output stores
#endif
constexpr int GL_NW = GL_NW_;                             // waves per block
constexpr int GL_NT = GL_H / GL_NW / 16;                  // 16-unit tiles per wave and gate
constexpr int GL_KS = GL_H / 32;                          // k-steps of the forward product (K = H)
constexpr int GL_KREG = GL_NW == 4 ? 8 : 6;               // ... of them with the W fragments in registers (the rest: LDS)
constexpr int GL_HROW = GL_H * 2 + 16;                    // bytes of a state row in LDS (odd multiple of 16: conflict-free b128 reads)
constexpr int GL_WLDS = GL_NW * 3 * GL_NT * (GL_KS - GL_KREG) * 1024;      // W fragments in LDS: [wave][gate, tile][k-step][lane] x 16 B
constexpr int GL_FWD_LDS = GL_WLDS + 2 * 16 * GL_HROW + GL_G * 4;

__device__ __forceinline__ float gl_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float gl_tanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.8853900817779268f * x)); }

__device__ __forceinline__ u32x4_t gl_pack8(const float* p) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    return u32x4_t{pack_bf16_rne(a.x, a.y), pack_bf16_rne(a.z, a.w), pack_bf16_rne(b.x, b.y), pack_bf16_rne(b.z, b.w)};
}

}  // namespace

__global__ __launch_bounds__(GL_NW * 64) void gru_local_fwd_kernel(GruLocalArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* w_lds = lds;
    unsigned char* h_lds = lds + GL_WLDS;                   // [2][16 rows][GL_HROW]
    float* bias_lds = reinterpret_cast<float*>(lds + GL_WLDS + 2 * 16 * GL_HROW);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lq = lane >> 4, lr = lane & 15;
    const int chain = blockIdx.y, b0 = blockIdx.x * 16, B = a.B, T = a.T;
    const bool rev = a.reverse[chain] != 0;
    const float* __restrict__ W = a.w_hh[chain];
    const float* __restrict__ gi = a.gi[chain];
    float* __restrict__ hs = a.hs[chain];
    float* __restrict__ save = a.save[chain];
    const int u0 = wave * 16 * GL_NT;

    // ---- W_hh fragments: A[m = lr][k = 8 lq + i] = W[(g H + u0 + 16 hf + lr)][32 ks + 8 lq + i]
    u32x4_t wr[3][GL_NT][GL_KREG];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int hf = 0; hf < GL_NT; ++hf) {
            const float* row = W + (size_t)(g * GL_H + u0 + 16 * hf + lr) * GL_H + 8 * lq;
#pragma unroll
            for (int ks = 0; ks < GL_KREG; ++ks) wr[g][hf][ks] = gl_pack8(row + 32 * ks);
#pragma unroll
            for (int ks = GL_KREG; ks < GL_KS; ++ks)
                *reinterpret_cast<u32x4_t*>(w_lds + (((wave * 3 * GL_NT + g * GL_NT + hf) * (GL_KS - GL_KREG) + (ks - GL_KREG)) * 64 + lane) * 16) = gl_pack8(row + 32 * ks);
        }
    for (int i = tid; i < GL_G; i += GL_NW * 64) bias_lds[i] = a.b_hh[chain][i];
    for (int i = tid; i < 2 * 16 * GL_HROW / 4; i += GL_NW * 64) reinterpret_cast<unsigned*>(h_lds)[i] = 0u;

    const int b = b0 + lr;
    const bool bv = b < B;
    const int sl = bv ? a.seq_len[b] : 0;
    float hp[GL_NT][4];
#pragma unroll
    for (int hf = 0; hf < GL_NT; ++hf)
#pragma unroll
        for (int r = 0; r < 4; ++r) hp[hf][r] = 0.f;
    // byte offset of this lane's B fragment (row lr, k-group lq) and of its own 4-unit slots in a state image
    const unsigned hb_off = (unsigned)(lr * GL_HROW + 16 * lq);
    const unsigned hw_off = (unsigned)(lr * GL_HROW + (u0 + 4 * lq) * 2);
    __syncthreads();

    for (int s = 0; s < T; ++s) {
        const int t = rev ? T - 1 - s : s;
        const unsigned char* h_cur = h_lds + (s & 1) * 16 * GL_HROW;
        unsigned char* h_nxt = h_lds + ((s + 1) & 1) * 16 * GL_HROW;
        const size_t tb = (size_t)t * B + b;
        // the input projections of this step (used after the MFMAs: the loads fly during them)
        float4 gin[3][GL_NT];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int hf = 0; hf < GL_NT; ++hf)
                gin[g][hf] = (bv && !(GL_DBG & 1)) ? *reinterpret_cast<const float4*>(gi + tb * GL_G + g * GL_H + u0 + 16 * hf + 4 * lq) : make_float4(0.f, 0.f, 0.f, 0.f);
        // accumulators start at b_hh
        f32x4 acc[3][GL_NT];
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int hf = 0; hf < GL_NT; ++hf) {
                const float4 bb = *reinterpret_cast<const float4*>(bias_lds + g * GL_H + u0 + 16 * hf + 4 * lq);
                acc[g][hf] = f32x4{bb.x, bb.y, bb.z, bb.w};
            }
        u32x4_t hb[2];
        hb[0] = *reinterpret_cast<const u32x4_t*>(h_cur + hb_off);
#pragma unroll
        for (int ks = 0; ks < GL_KS; ++ks) {
            if (ks + 1 < GL_KS) hb[(ks + 1) & 1] = *reinterpret_cast<const u32x4_t*>(h_cur + hb_off + (ks + 1) * 64);
#pragma unroll
            for (int g = 0; g < 3; ++g)
#pragma unroll
                for (int hf = 0; hf < GL_NT; ++hf) {
                    u32x4_t wf;
                    if (ks < GL_KREG) wf = wr[g][hf][ks < GL_KREG ? ks : 0];
                    else wf = *reinterpret_cast<const u32x4_t*>(w_lds + (((wave * 3 * GL_NT + g * GL_NT + hf) * (GL_KS - GL_KREG) + (ks - GL_KREG)) * 64 + lane) * 16);
                    acc[g][hf] = mfma_b16(wf, hb[ks & 1], acc[g][hf]);
                }
        }
        // gates, state, outputs: lane = (units u0 + 16 hf + 4 lq + r, batch row lr)
        const bool live = t < sl;
#pragma unroll
        for (int hf = 0; hf < GL_NT; ++hf) {
            const float gr[4] = {gin[0][hf].x, gin[0][hf].y, gin[0][hf].z, gin[0][hf].w};
            const float gz[4] = {gin[1][hf].x, gin[1][hf].y, gin[1][hf].z, gin[1][hf].w};
            const float gn[4] = {gin[2][hf].x, gin[2][hf].y, gin[2][hf].z, gin[2][hf].w};
            float hn[4], f0[4], f1[4], f2[4], f3[4], f4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ghn = acc[2][hf][r];
                const float rg = gl_sigmoid(gr[r] + acc[0][hf][r]);
                const float zg = gl_sigmoid(gz[r] + acc[1][hf][r]);
                const float ng = gl_tanh(gn[r] + rg * ghn);
                const float hprev = hp[hf][r];
                hn[r] = live ? (1.f - zg) * ng + zg * hprev : 0.f;
                const float cn = (1.f - zg) * (1.f - ng * ng);
                f0[r] = cn * ghn * rg * (1.f - rg); f1[r] = (hprev - ng) * zg * (1.f - zg); f2[r] = cn; f3[r] = cn * rg; f4[r] = zg;
                hp[hf][r] = hn[r];
            }
            *reinterpret_cast<uint2*>(h_nxt + hw_off + 32 * hf) = make_uint2(pack_bf16_rne(hn[0], hn[1]), pack_bf16_rne(hn[2], hn[3]));
            if (bv && !(GL_DBG & 2)) {
                const int u = u0 + 16 * hf + 4 * lq;
                *reinterpret_cast<float4*>(hs + tb * GL_H + u) = make_float4(hn[0], hn[1], hn[2], hn[3]);
                if (save) {
                    float* sv = save + tb * 5 * GL_H + u;
                    *reinterpret_cast<float4*>(sv) = make_float4(f0[0], f0[1], f0[2], f0[3]);
                    *reinterpret_cast<float4*>(sv + GL_H) = make_float4(f1[0], f1[1], f1[2], f1[3]);
                    *reinterpret_cast<float4*>(sv + 2 * GL_H) = make_float4(f2[0], f2[1], f2[2], f2[3]);
                    *reinterpret_cast<float4*>(sv + 3 * GL_H) = make_float4(f3[0], f3[1], f3[2], f3[3]);
                    *reinterpret_cast<float4*>(sv + 4 * GL_H) = make_float4(f4[0], f4[1], f4[2], f4[3]);
                }
            }
        }
        __syncthreads();
    }
}

bool gru_local_takes(int nlayers, int H, int bf16) {
    static const bool off = getenv("PBSED_GRU_LOCAL") && getenv("PBSED_GRU_LOCAL")[0] == '0';
    return !off && bf16 && nlayers == 1 && H == GL_H;
}

int gru_local_fwd_launch(const GruLocalArgs& a, hipStream_t s) {
    PBSED_DYN_LDS_ONCE(gru_local_fwd_kernel, GL_FWD_LDS);
    hipLaunchKernelGGL(gru_local_fwd_kernel, dim3((a.B + 15) / 16, a.nchains), dim3(GL_NW * 64), GL_FWD_LDS, s, a);
    return check_launch("gru_local_fwd");
}

}  // namespace pbsed
