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
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "common.h"
#include "gru_granule_map.h"
#include "pbsed_internal.h"

namespace pbsed {

constexpr int GRU_MAX_LAYERS = 4;
constexpr int GRU_MAX_CHAINS = 6;        // both directions of three networks in one launch (ensemble inference)

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
    int poll_delay, poll_delay_gate;   // granule kernels: first-poll delays (PollPacer) of the non-gate / gate waves
    int ring_xcd, nby;    // granule kernels: ring_xcd = H/16 > 0 selects the 1-D XCD-aware role mapping (granule_role)
    int local;            // 1 (BPTT with the ring-per-XCD mapping): the LAST ring in scan order - no projection group reads it, its
                          // only readers are its own blocks on its own XCD, whose L2 is their coherence point - publishes its state
                          // with a PLAIN store and looks at it first with a PLAIN load; only retries bypass the L1 (sc1)
    unsigned xcc_map;     // local = 1: the XCC_ID the placement probe saw for block ids = r (mod 8), four bits per residue r
                          // (xcd_placement below); a block of the local ring whose own XCC_ID differs raises error bit 2
    unsigned long long* prof;   // diagnostics (pbsed_gru_set_prof): shader-clock stamps of block `prof_block`, steps 200..231
    int prof_block;
};

// Gate activations of the persistent forward scan.  The libm forms (expf, a full-precision division, tanhf with its
// small-argument branch) are ~110 VALU instructions per gate thread ON the step's critical path (reduction -> gates ->
// publish: 730 of a step's 6 100 clocks in the shader-clock profile, 380 with these); the hardware forms are 12:
// sigmoid(x) = rcp(1 + exp2(-x log2 e)), tanh(x) = 1 - 2 rcp(1 + exp2(2 x log2 e)) (v_exp_f32 / v_rcp_f32: 1 ulp each; exp2
// overflowing to +inf gives rcp(inf) = 0, the right limit on both sides).  Absolute error <= 2e-7 per activation, the same
// class as the tagged LSB of the exchanged state; the launch-per-step kernels keep expf / tanhf.
__device__ __forceinline__ float gate_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float gate_tanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.8853900817779268f * x)); }

// shader-clock stamp of one lane (diagnostics only; s_memtime waits for the wave's outstanding LDS / scalar loads)
__device__ __forceinline__ void prof_stamp(unsigned long long* p) { *p = __builtin_readcyclecounter(); }

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

// split form: issue every operand load of a step first, run the MFMAs afterwards (overlaps the L2/MALL
// latencies of the recurrent and the input-projection operands)
template <int KB, int NG>
__device__ __forceinline__ void mm_load(float4 (&vv)[KB], float4 (&wv)[KB][NG], const float* __restrict__ w, size_t gstride,
                                        const float* __restrict__ v, bool vvalid, int wave, int lq) {
#pragma unroll
    for (int i = 0; i < KB; ++i) {
        const int k = (wave * KB + i) * 16 + lq * 4;
        vv[i] = vvalid ? *reinterpret_cast<const float4*>(v + k) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int g = 0; g < NG; ++g) wv[i][g] = *reinterpret_cast<const float4*>(w + g * gstride + k);
    }
}
template <int KB, int NG>
__device__ __forceinline__ void mm_fma(f32x4 (&acc)[NG], const float4 (&vv)[KB], const float4 (&wv)[KB][NG]) {
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
// grid (H/16, ceil(B/16), nchains*nlayers), NW*64 threads; H = KB*NW*16.
template <int KB, int NW>
__global__ __launch_bounds__(NW * 64) void gru_stack_fwd_kernel(GruStackArgs a) {
    constexpr int H = KB * NW * 16;
    __shared__ float red[NW][3][64][4];
    const int chain = blockIdx.z % a.nchains, layer = blockIdx.z / a.nchains;
    const int step = a.launch - layer;
    if (step < 0 || step >= a.T) return;
    const int bxj = blockIdx.x;
    const GruStackLayer& L = a.lc[chain][layer];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane >> 4, lr = lane & 15;
    const int j0 = bxj * 16, b0 = blockIdx.y * 16, B = a.B;
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
    const bool rowv = (b0 + lr) < B;
    constexpr int HW = NW / 2;
    if (layer == 0) {
        // all waves split K of the recurrent matmul
        if (has_prev)
            mm_rows<KB, 3>(acc, L.w_hh + (size_t)(j0 + lr) * H, H, H * H, L.hs + ((size_t)tp * B + b0 + lr) * H, H, rowv,
                           wave, lq);
    } else {
        // upper layers: waves [0, HW) do W_hh h_{t-1}, waves [HW, NW) do W_ih x_t - the two operand streams are
        // fetched concurrently instead of back to back (the step is bound by their L2/MALL latency)
        const bool is_ih = wave >= HW;
        const int wq = is_ih ? wave - HW : wave;
        const float* W = (is_ih ? L.w_ih : L.w_hh) + (size_t)(j0 + lr) * H;
        const float* V = is_ih ? a.lc[chain][layer - 1].hs + ((size_t)t * B + b0 + lr) * H
                               : L.hs + ((size_t)tp * B + b0 + lr) * H;
        const bool act = is_ih || has_prev;
        if (act) mm_rows<2 * KB, 3>(acc, W, H, H * H, V, H, rowv, wq, lq);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        red[wave][0][lane][r] = acc[0][r];
        red[wave][1][lane][r] = acc[1][r];
        red[wave][2][lane][r] = acc[2][r];
    }
    __syncthreads();
    if (!bv) return;
    const int src = (u >> 2) * 16 + bb, reg = u & 3;
    float s[4] = {0.f, 0.f, 0.f, 0.f};      // r, z, W_hn h, W_in x
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        s[0] += red[w][0][src][reg];
        s[1] += red[w][1][src][reg];
        if (layer == 0 || w < HW) s[2] += red[w][2][src][reg]; else s[3] += red[w][2][src][reg];
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
    __shared__ float red[NW][64][4];
    const int chain = blockIdx.z % a.nchains, layer = blockIdx.z / a.nchains;
    const int top = a.nlayers - 1;
    const int bstep = a.launch - (top - layer);
    if (bstep < 0 || bstep >= a.T) return;
    const int bxj = blockIdx.x;
    const GruStackLayer& L = a.lc[chain][layer];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lq = lane >> 4, lr = lane & 15;
    const int j0 = bxj * 16, b0 = blockIdx.y * 16, B = a.B;
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
    f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
    const bool rowv = (b0 + lr) < B;
    constexpr int HW = NW / 2;
    if (layer == top) {
        if (has_next)
            mm_rows<KB3, 1>(acc, L.w_hh + (size_t)(j0 + lr) * G, G, 0, L.dgh + ((size_t)tn * B + b0 + lr) * G, G, rowv,
                            wave, lq);
    } else {
        // waves [0, HW): carry = dgh_next W_hh; waves [HW, NW): dy = dgi_upper,t W_ih_upper (concurrent streams)
        const bool is_dy = wave >= HW;
        const int wq = is_dy ? wave - HW : wave;
        const float* W = (is_dy ? L.w_ih : L.w_hh) + (size_t)(j0 + lr) * G;
        const float* V = is_dy ? a.lc[chain][layer + 1].dgi + ((size_t)t * B + b0 + lr) * G
                               : L.dgh + ((size_t)tn * B + b0 + lr) * G;
        if (is_dy || has_next) mm_rows<2 * KB3, 1>(acc, W, G, 0, V, G, rowv, wq, lq);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) red[wave][lane][q] = acc[0][q];
    __syncthreads();
    if (!bv) return;
    const int src = (u >> 2) * 16 + bb, reg = u & 3;
    float carry = 0.f, dylow = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        if (layer == top || w < HW) carry += red[w][src][reg]; else dylow += red[w][src][reg];
    }
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


typedef unsigned int __attribute__((address_space(1))) gu32;

// Partial sums of one (wave, gate) in LDS: the accumulator fragment (64 lanes x 4 registers) with its four lane groups 80 floats
// apart instead of 64 (see gru_granule_fwd_body): writer lane (lq, lr) -> lq * 80 + lr * 4, gate thread (bb, u) -> its element.
constexpr int RED_ROW = 4 * 80;

// ============================================================================================
// Persistent scans ("granules"): the exchanged values ARE the flags.  Every h value is published as one 4-byte word
// - the fp32 value with its mantissa LSB replaced by the call's parity bit - with a single write-through (sc1) store
// into a [T][B][H] array that is written exactly once per call; consumers poll the words they need with
// L1-bypassing 16-byte loads until all parity bits equal the call's (cdna_hip_programming.md Guideline 16, form R2
// with a 1-bit tag: no fence, no drain, no counter).  The workspace is zero before its first use and every call
// flips the parity, so the previous call's words never match.  The exchanged quantity is DEFINED as the truncated
// value (LSB cleared, <= 1 ulp): producer, consumers and the saved outputs all use exactly that number.  Half the
// bytes of {tag, value} granules: the polls (16 KB per workgroup and step instead of 32) are what loads the fabric.
// W fragments stay in registers for all T steps, the previous state of a thread's own unit too.
// ============================================================================================

__device__ __forceinline__ float tag_clear(float v) { return __uint_as_float(__float_as_uint(v) & ~1u); }
__device__ __forceinline__ void publish(gu32* p, float v_cleared, unsigned parity) {
    __hip_atomic_store(p, __float_as_uint(v_cleared) | parity, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// XCD-local form (GruStackArgs::local): a plain store - it reaches the XCD's L2 (the L1 is write-through), which is where
// every reader of this word looks; no write-through to the fabric, whose completion is what a racing sc1 poll waits for.
__device__ __forceinline__ void publish_local(gu32* p, float v_cleared, unsigned parity) {
    __hip_atomic_store(p, __float_as_uint(v_cleared) | parity, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The XCD this wave runs on: XCC_ID[3:0] of HW_REG_XCC_ID (hardware register 20 on gfx942 / gfx950).
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u; }

// Error word of a scan: bit 0 = a hand-off timed out, bit 1 = a block of an XCD-local ring found itself on another XCD than
// the placement probe promised (its plain stores then never reach its readers' L2: the time-out that follows has a cause).
__device__ __forceinline__ void raise_error(unsigned* err_flag, unsigned bit) {
    __hip_atomic_fetch_or((gu32*)err_flag, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// First-poll pacing.  Polls that come before the data only add fabric traffic and slow everybody's hand-off down (two
// batches in flight per wave: 2x slower scans), and a missed first poll costs a full round trip.  The waves that do
// not take part in the gate phase reach the next step's poll a gate phase early, the gate waves one store-to-visible
// latency early; both sleep a fixed number of 64-clock units before the first poll of a step (measured defaults in
// granule_poll_delays; a self-tuning delay over-shoots during the pipeline fill and ends up slower).
struct PollPacer {
    int delay;
    __device__ __forceinline__ void wait() const {
        for (int d = 0; d < delay; ++d) __builtin_amdgcn_s_sleep(1);
    }
};

// One wave polls the words of its K range [k0, k0 + 16*NL) for 16 batch rows with 16-byte write-through-visible
// (sc1) buffer loads: lane (lq, lr) reads, per load n, the four words k0 + n*16 + lq*4 + {0..3} of row lr.  The exchange
// arrays are TILE-MAJOR - [T][batch tile][H/16 producers][16 rows][16 units] - so one load instruction covers exactly the
// contiguous 1 KB tile one producer block published (8 whole 128-byte lines instead of 16 half lines of a row-major
// [T][B][H] array) and a producer's 256 publishing threads write one contiguous 1 KB; nothing is fetched twice.  All loads of a step are issued
// before any tag is looked at (one fabric round trip per step); out[n] = the four (tag-cleared) values.
template <int NL, int STEP = 1024>
__device__ __forceinline__ int poll_batch(float4 (&out)[NL], __amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned parity,
                                          bool valid, unsigned* err_flag, bool local = false) {
    u32x4_t q[NL];
#pragma unroll
    for (int n = 0; n < NL; ++n) q[n] = u32x4_t{0u, 0u, 0u, 0u};
    int spin = 0;
    for (;; ++spin) {
        bool ok = true;
        asm volatile("" ::: "memory");                  // the load builtins are not volatile: every attempt loads again
        if (valid) {
            // local: every word has its own location, written once per call - the FIRST look may be a plain load (the L1 was
            // invalidated at the launch and cannot hold a line it never read; inside an XCD the L2 is the coherence point);
            // a look that came too early leaves the stale line in the L1, so every retry bypasses it (sc1)
            if (local && spin == 0) {
#pragma unroll
                for (int n = 0; n < NL; ++n) q[n] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + n * STEP, 0, 0);
            } else {
#pragma unroll
                for (int n = 0; n < NL; ++n) q[n] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + n * STEP, 0, /*aux = sc1*/ 16);
            }
            unsigned all1 = 1u, any1 = 0u;
#pragma unroll
            for (int n = 0; n < NL; ++n) {
                all1 &= q[n].x & q[n].y & q[n].z & q[n].w;
                any1 |= q[n].x | q[n].y | q[n].z | q[n].w;
            }
            ok = parity ? (all1 & 1u) != 0u : (any1 & 1u) == 0u;
        }
        if (__all(ok)) break;
        if (spin > (1 << 18)) {                     // bounded: raise the error flag and go on with garbage
            raise_error(err_flag, 1u);
            break;
        }
        if (spin > 8) __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int n = 0; n < NL; ++n)
        out[n] = make_float4(__uint_as_float(q[n].x & ~1u), __uint_as_float(q[n].y & ~1u), __uint_as_float(q[n].z & ~1u),
                             __uint_as_float(q[n].w & ~1u));
    return spin;
}

// The X3 variants of the scans run their products with bf16x3 operands on the bf16 MFMA (Bf3 / mfma_x3 in common.h).

// Block roles of the granule scans.  Every (chain, layer) has a RING of H/16 x ceil(B/16) blocks that carries the
// recurrence (h_{t-1} -> h_t, or dh_{t+1} -> dh_t) and, for every layer boundary, a group of the same size of
// PROJECTION blocks that turn the neighbouring ring's step output into this ring's step input (forward:
// gi_t = W_ih h^{l-1}_t + b_ih; backward: dy_t = W_ih^{l+1,T} dgi^{l+1}_t) and publish it as granules too.  The
// projections have no recurrence, run ahead of the ring that consumes them, and keep every block at K = H per step.
// Group id within a chain: 0 = ring of the first layer in scan order, 2m-1 = projection into / 2m = ring of the m-th.
struct GranuleRole {
    int bx, by, chain, gid;
    bool idle;
};
__device__ __forceinline__ GranuleRole granule_role(const GruStackArgs& a) {
    GranuleRole r;
    r.idle = false;
    if (a.ring_xcd) {
        // 1-D grid, XCD = block id % 8: ring units (chain, layer, batch tile) one per XCD so that the recurrence's
        // hand-off stays inside one L2; the projection blocks (no recurrence) are dealt round-robin over all XCDs
#include "gru_granule_role.inc"
    } else {
        r.bx = blockIdx.x; r.by = blockIdx.y;
        r.chain = blockIdx.z % a.nchains; r.gid = blockIdx.z / a.nchains;
    }
    return r;
}

// One thread waits for NQ words it alone consumes (stride `stride` words); they were requested earlier (q holds the
// first answers) and are normally there already.
template <int NQ>
__device__ __forceinline__ void wait_own_granules(unsigned (&q)[NQ], const gu32* p, size_t stride, unsigned parity,
                                                  unsigned* err_flag) {
    for (int spin = 0;; ++spin) {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < NQ; ++i) ok = ok && (q[i] & 1u) == parity;
        if (ok) break;
        if (spin > (1 << 18)) {
            raise_error(err_flag, 1u);
            break;
        }
        if (spin > 8) __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int i = 0; i < NQ; ++i) q[i] = __hip_atomic_load(p + i * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// GW: four dedicated gate waves (threads 0..255: reduction, gate maths, publish, saves) in front of the NW contraction
// waves.  Vector memory operations of a wave complete in issue order, so a wave that publishes h_t with a write-through
// store and then polls for step t + 1 waits for that store's acknowledgement (0.35 us/step) before its poll counts as
// returned; with GW the polling waves never store and the publishing waves never poll on the critical path.
// NB: 16-row batch tiles per block (1, or 2 for more than 32 clips: the rings of 64 clips then need 192 instead of 384
// co-resident blocks and stay one launch; a block's two tiles share the W fragments, their polls are issued together).
template <int NL, int NB>
__device__ __forceinline__ int poll_tiles(float4 (&out)[NB][NL], __amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned tile_stride,
                                           unsigned parity, const bool (&valid)[NB], unsigned* err_flag, bool local = false) {
    u32x4_t q[NB][NL];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int n = 0; n < NL; ++n) q[nb][n] = u32x4_t{0u, 0u, 0u, 0u};
    int spin = 0;
    for (;; ++spin) {
        unsigned all1 = 1u, any1 = 0u;
        asm volatile("" ::: "memory");                  // see poll_batch
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            if (valid[nb]) {
                if (local && spin == 0) {                 // see poll_batch
#pragma unroll
                    for (int n = 0; n < NL; ++n)
                        q[nb][n] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + nb * tile_stride + n * 1024, 0, 0);
                } else {
#pragma unroll
                    for (int n = 0; n < NL; ++n)
                        q[nb][n] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + nb * tile_stride + n * 1024, 0, /*aux = sc1*/ 16);
                }
            }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
            if (valid[nb]) {
#pragma unroll
                for (int n = 0; n < NL; ++n) {
                    all1 &= q[nb][n].x & q[nb][n].y & q[nb][n].z & q[nb][n].w;
                    any1 |= q[nb][n].x | q[nb][n].y | q[nb][n].z | q[nb][n].w;
                }
            }
        const bool ok = parity ? (all1 & 1u) != 0u : (any1 & 1u) == 0u;
        if (__all(ok)) break;
        if (spin > (1 << 18)) {                     // bounded: raise the error flag and go on with garbage
            raise_error(err_flag, 1u);
            break;
        }
        if (spin > 8) __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int n = 0; n < NL; ++n)
            out[nb][n] = make_float4(__uint_as_float(q[nb][n].x & ~1u), __uint_as_float(q[nb][n].y & ~1u),
                                     __uint_as_float(q[nb][n].z & ~1u), __uint_as_float(q[nb][n].w & ~1u));
    return spin;
}

template <int KB, int NW, bool GW, int XS = 0, int NB = 1>
__device__ __forceinline__ void gru_granule_fwd_body(const GruStackArgs& a, unsigned* gran_h_, unsigned* gran_gi_, unsigned epoch,
                                                     unsigned* err_flag, float (&red)[2][NW][NB][3][RED_ROW], int& s_err) {
    constexpr int H = KB * NW * 16, NL = KB;           // NL: 16-byte loads per lane (16 k each) of this wave's H/NW range
    constexpr int GWV = GW ? 4 : 0;
    const GranuleRole role = granule_role(a);
    if (role.idle) return;
    const int chain = role.chain, layer = (role.gid + 1) >> 1;
    const bool is_proj = role.gid & 1;
    const GruStackLayer& L = a.lc[chain][layer];
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) - GWV, lq = lane >> 4, lr = lane & 15;
    const bool is_mfma = wave >= 0;               // GW: waves 0..3 of the block only run the gate phase
    const int j0 = role.bx * 16, b0 = role.by * 16 * NB, B = a.B;
    const bool rev = a.reverse[chain] != 0;
    const int Bp = (B + 15) / 16 * 16;                       // the tile-major h array holds whole batch tiles
    const size_t per_cl = (size_t)a.T * Bp * H, per_cl_b = (size_t)a.T * B * H;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(gran_h_, 0, (unsigned)(per_cl * a.nchains * a.nlayers * 4), 0x00020000);
    const unsigned parity = epoch & 1u;
    // ring: h_t, tile-major; this block's first tile of step t starts at g_own + t * Bp * H, the next one H * 16 words on
    gu32* g_own = (gu32*)gran_h_ + PBSED_GM_RING_BASE(chain, a.nlayers, layer, per_cl, role.by * NB, H, role.bx);
    gu32* g_gi = (gu32*)gran_gi_ + (size_t)(chain * (a.nlayers - 1) + (layer > 0 ? layer - 1 : 0)) * per_cl_b * 3;  // [T][B][3][H]
    // gate thread -> (batch row, unit) row-major: a wave's global stores / loads are whole 64-byte row segments and a producer's
    // 1 KB exchange tile is [16 rows][16 units].  The partial sums a gate thread adds sit in the MFMA D layout (lane (u >> 2) *
    // 16 + bb, register u & 3): with the accumulators stored lane after lane those reads hit 16 banks four times each (1 000 of a
    // step's 6 100 clocks in the shader-clock profile), so the four lane groups of a fragment are 80 floats apart (RED_POS):
    // bank = (u >> 2) * 16 + (bb & 3) * 4 + (u & 3), all 64 different.
    const int u = tid & 15, bb = (tid >> 4) & 15, j = j0 + u;
    const int red_w = lq * 80 + lr * 4, red_r = (u >> 2) * 80 + bb * 4 + (u & 3);
    int b[NB], sl[NB];
    bool bv[NB], rowv[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        b[nb] = b0 + nb * 16 + bb;
        bv[nb] = tid < 256 && b[nb] < B;
        rowv[nb] = (b0 + nb * 16 + lr) < B;
        sl[nb] = bv[nb] ? a.seq_len[b[nb]] : 0;
    }
    const float* bias = is_proj ? L.b_ih : L.b_hh;
    const float bs_r = bias[j0 + u], bs_z = bias[H + j0 + u], bs_n = bias[2 * H + j0 + u];
    // every wave contracts its H/NW slice of K; within it a lane owns k = k0 + n*16 + lq*4 + {0..3} (the poll's
    // load pattern) and the weights follow the same order
    const int k0 = (is_mfma ? wave : 0) * NL * 16;
    constexpr int NM = (NL + 1) / 2;                 // X3: bf16 MFMAs (K = 32 = two of the poll's loads) per product term
    float4 wv[XS ? 1 : NL][3];
    BfOp<XS> w3[XS ? NM : 1][3];
    if (is_mfma) {
        const float* W = (is_proj ? L.w_ih : L.w_hh) + (size_t)(j0 + lr) * H + k0 + lq * 4;
        if constexpr (XS > 0) {
#pragma unroll
            for (int m = 0; m < NM; ++m)
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const float4 w0 = *reinterpret_cast<const float4*>(W + (size_t)g * H * H + (2 * m) * 16);
                    const float4 w1 = 2 * m + 1 < NL ? *reinterpret_cast<const float4*>(W + (size_t)g * H * H + (2 * m + 1) * 16)
                                                     : make_float4(0.f, 0.f, 0.f, 0.f);
                    w3[m][g] = make_op<XS>(w0, w1);
                }
        } else {
#pragma unroll
            for (int n = 0; n < NL; ++n)
#pragma unroll
                for (int g = 0; g < 3; ++g) wv[n][g] = *reinterpret_cast<const float4*>(W + (size_t)g * H * H + n * 16);
        }
    }
    // a projection reads h_t of the layer below, a ring its own h_{t-1}
    const unsigned cl_src = chain * a.nlayers + (is_proj ? layer - 1 : layer);
    const unsigned voff0 = PBSED_GM_POLL_OFFSET0(cl_src, per_cl, role.by * NB, H, k0, lr, lq);
    const unsigned step_t = (unsigned)(Bp * H * 4);           // bytes per time step of one (chain, layer)
    constexpr unsigned tile_bytes = 16u * H * 4u;            // one batch tile of one step
    float h_reg[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) h_reg[nb] = 0.f;
    const PollPacer pacer{(GW || threadIdx.x >= 256) ? a.poll_delay : a.poll_delay_gate};
    if (tid == 0) s_err = 0;
    __syncthreads();
    float gn_r[NB], gn_z[NB], gn_n[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) gn_r[nb] = gn_z[nb] = gn_n[nb] = 0.f;
    auto load_gi = [&](int st) {                    // first layer: input projection computed before the scan
        if (layer == 0) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                if (bv[nb]) {
                    const float* gi = L.gi + ((size_t)(rev ? a.T - 1 - st : st) * B + b[nb]) * 3 * H;
                    gn_r[nb] = gi[j]; gn_z[nb] = gi[H + j]; gn_n[nb] = gi[2 * H + j];
                }
        }
    };
    load_gi(0);
    // diagnostics: lane 0 of the first contraction wave (slots 0..5) and of the first gate wave (slots 8..12) of one block
    const bool prof_blk = a.prof != nullptr && (int)blockIdx.x == a.prof_block;
    const bool prof_c = prof_blk && tid == GWV * 64, prof_g = prof_blk && tid == 0;

    // NB = 2: the block's two batch tiles run half a step apart - tile 1 is polled, contracted and gated while tile 0's new
    // states travel to the ring's other blocks and back (the hand-off is ~0.8 us of a 1.8 us step) -
    // one barrier per tile and step; both tiles in one phase (poll both, contract both, gate both) put the hand-off
    // behind twice the work instead: 2.8 us per step at B = 64 against 1.8 us staggered.
    for (int step = 0; step < a.T; ++step) {
        const int t = rev ? a.T - 1 - step : step;
        const int tp = rev ? t + 1 : t - 1;
        const bool has_prev = step > 0;
        const int par = step & 1;                     // `red` is double-buffered per tile: one barrier per tile and step
        const bool prof_now = prof_blk && step >= 200 && step < 232;
        unsigned long long* pslot = a.prof + (prof_now ? (step - 200) * 16 : 0);
        if (prof_now && prof_c) prof_stamp(pslot + 0);
        if (prof_now && prof_g) prof_stamp(pslot + 8);
        // another block's time-out is looked for every 32 steps only: the agent-scope load stalls its wave
        if (tid == 0 && (step & 31) == 31 && __hip_atomic_load((gu32*)err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) s_err = 1;
        const bool contract = (is_proj || has_prev) && is_mfma;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const bool p0 = nb == 0;                  // the diagnostics follow the first tile
            float gi_r = gn_r[nb], gi_z = gn_z[nb], gi_n = gn_n[nb];
            f32x4 acc[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            float4 x[1][NL];
            if (contract) {
                pacer.wait();
                if (prof_now && prof_c && p0) prof_stamp(pslot + 1);
                const bool v1[1] = {rowv[nb]};
                const int spins = poll_tiles<NL, 1>(x, rsrc, voff0 + (unsigned)(is_proj ? t : tp) * step_t + nb * tile_bytes, tile_bytes,
                                                    parity, v1, err_flag, false);
                if (prof_now && prof_c && p0) { prof_stamp(pslot + 2); pslot[5] = (unsigned long long)spins; }
            }
            // requests issued behind the poll (loads return in order, anything older would hold the poll back):
            // next step's input projection (first layer) / this step's projected input granules (other rings)
            unsigned qg[3] = {0u, 0u, 0u};
            if (!is_proj && layer > 0 && bv[nb]) {
                const gu32* gp = g_gi + ((size_t)t * B + b[nb]) * 3 * H + j;
#pragma unroll
                for (int g = 0; g < 3; ++g) qg[g] = __hip_atomic_load(gp + (size_t)g * H, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (step + 1 < a.T && layer == 0 && bv[nb]) {
                const float* gi = L.gi + ((size_t)(rev ? a.T - 2 - step : step + 1) * B + b[nb]) * 3 * H;
                gn_r[nb] = gi[j]; gn_z[nb] = gi[H + j]; gn_n[nb] = gi[2 * H + j];
            }
            if (contract) {
                if constexpr (XS > 0) {
#pragma unroll
                    for (int m = 0; m < NM; ++m) {
                        const BfOp<XS> xs = make_op<XS>(x[0][2 * m], 2 * m + 1 < NL ? x[0][2 * m + 1] : make_float4(0.f, 0.f, 0.f, 0.f));
#pragma unroll
                        for (int g = 0; g < 3; ++g) acc[g] = mfma_op<XS>(w3[m][g], xs, acc[g]);
                    }
                } else {
#pragma unroll
                    for (int n = 0; n < NL; ++n)
#pragma unroll
                        for (int g = 0; g < 3; ++g) {
                            acc[g] = mfma16(wv[n][g].x, x[0][n].x, acc[g]);
                            acc[g] = mfma16(wv[n][g].y, x[0][n].y, acc[g]);
                            acc[g] = mfma16(wv[n][g].z, x[0][n].z, acc[g]);
                            acc[g] = mfma16(wv[n][g].w, x[0][n].w, acc[g]);
                        }
                }
            }
            if (is_mfma) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    red[par][wave][nb][0][red_w + r] = acc[0][r];
                    red[par][wave][nb][1][red_w + r] = acc[1][r];
                    red[par][wave][nb][2][red_w + r] = acc[2][r];
                }
            }
            if (prof_now && prof_c && p0) prof_stamp(pslot + 3);
            __syncthreads();
            if (prof_now && prof_c && p0) prof_stamp(pslot + 4);
            if (prof_now && prof_g && p0) prof_stamp(pslot + 9);
            // some hand-off timed out: every wave leaves - but the look at the flag is a dependent LDS round trip, and the gate
            // threads are on the step's critical path here (barrier -> reduction -> gates -> publish): they read it with the
            // partial sums and act on it after the publish (what they publish in a failed call is discarded with the call)
            const int err_seen = s_err;
            if (tid >= 256 && err_seen) return;
            if (bv[nb]) {
                const size_t tb = (size_t)t * B + b[nb];
                float sm[3] = {bs_r, bs_z, bs_n};
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    sm[0] += red[par][w][nb][0][red_r];
                    sm[1] += red[par][w][nb][1][red_r];
                    sm[2] += red[par][w][nb][2][red_r];
                }
                if (is_proj) {
                    gu32* dst = g_gi + tb * 3 * H + j;
#pragma unroll
                    for (int g = 0; g < 3; ++g) publish(dst + (size_t)g * H, tag_clear(sm[g]), parity);
                } else {
                    if (layer > 0) {
                        wait_own_granules<3>(qg, g_gi + tb * 3 * H + j, (size_t)H, parity, err_flag);
                        gi_r = __uint_as_float(qg[0] & ~1u); gi_z = __uint_as_float(qg[1] & ~1u);
                        gi_n = __uint_as_float(qg[2] & ~1u);
                    }
                    const float ghn = sm[2];
                    if (prof_now && prof_g && p0) prof_stamp(pslot + 10);
                    const float r = gate_sigmoid(gi_r + sm[0]);
                    const float z = gate_sigmoid(gi_z + sm[1]);
                    const float n = gate_tanh(gi_n + r * ghn);
                    const float hp = h_reg[nb];
                    const float h = tag_clear((t < sl[nb]) ? (1.f - z) * n + z * hp : 0.f);    // the state IS the truncated value
                    h_reg[nb] = h;
                    publish(g_own + PBSED_GM_RING_WORD_NB(t, Bp, H, nb, tid), h, parity);
                    if (prof_now && prof_g && p0) prof_stamp(pslot + 11);
                    L.hs[tb * H + j] = h;
                    if (L.save) {
                        // what BPTT multiplies dh_t with: d(r,z,n pre-activations)/dh, d(gh_n)/dh and z (granule save format)
                        float* sv = L.save + tb * 5 * H + j;
                        const float cn = (1.f - z) * (1.f - n * n);
                        sv[0] = cn * ghn * r * (1.f - r); sv[H] = (hp - n) * z * (1.f - z); sv[2 * H] = cn;
                        sv[3 * H] = cn * r; sv[4 * H] = z;
                    }
                }
            }
            if (prof_now && prof_g && nb == NB - 1) prof_stamp(pslot + 12);
            if (err_seen) return;
        }
    }
}

template <int KB, int NW, int XS, int NB>
__global__ __launch_bounds__((NW + 4) * 64) void gru_granule_fwd_gw_kernel(GruStackArgs a, unsigned* gran_h_, unsigned* gran_gi_,
                                                                         unsigned epoch, unsigned* err_flag) {
    extern __shared__ __attribute__((aligned(16))) float red_dyn[];          // [2][NW][NB][3][RED_ROW]: over 64 KB for NB = 2
    __shared__ int s_err;
    auto& red = *reinterpret_cast<float (*)[2][NW][NB][3][RED_ROW]>(red_dyn);
    gru_granule_fwd_body<KB, NW, true, XS, NB>(a, gran_h_, gran_gi_, epoch, err_flag, red, s_err);
}

// Backward twin.  Rings publish dh_t (masked by the sequence length) of their 16 units as granules [T][B][H];
// consumers rebuild the gate gradients they contract with as dh * (factor saved by the forward scan), the factors
// being plain loads issued one step ahead.  Projection blocks turn dh_t of the layer above into dy_t of the layer
// below (granules [T][B][H] as well); dh*z of a thread's own unit stays in a register.  Scan order = top layer first.
template <int KB, int NW, bool GW, int XS = 0>
__device__ __forceinline__ void gru_granule_bwd_body(const GruStackArgs& a, unsigned* gran_dh_, unsigned* gran_dy_, unsigned epoch,
                                                     unsigned* err_flag, float (&red)[2][NW][RED_ROW], int& s_err) {
    constexpr int H = KB * NW * 16, G = 3 * H, NL = KB;     // NL: 16-byte loads per lane (16 units each)
    constexpr int GWV = GW ? 4 : 0;                         // dedicated gate waves in front (see gru_granule_fwd_body)
    const GranuleRole role = granule_role(a);
    if (role.idle) return;
    const int chain = role.chain, top = a.nlayers - 1;
    const bool is_proj = role.gid & 1;
    const int layer = top - ((role.gid + 1) >> 1);    // ring: its layer; projection: the layer it produces dy for
    const bool ring_local = a.local && !is_proj && layer == 0;      // the bottom layer's ring: nobody else reads its dh
    // its plain stores are only seen inside ONE L2: every block of it checks that it runs where the placement probe said
    if (ring_local && threadIdx.x == 0 && xcc_id() != ((a.xcc_map >> (4 * (blockIdx.x & 7u))) & 15u)) raise_error(err_flag, 2u);
    const GruStackLayer& L = a.lc[chain][layer];
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) - GWV, lq = lane >> 4, lr = lane & 15;
    const bool is_mfma = wave >= 0;
    const int j0 = role.bx * 16, b0 = role.by * 16, B = a.B;
    const bool rev = a.reverse[chain] != 0;
    const int Bp = (B + 15) / 16 * 16;
    const size_t per_cl = (size_t)a.T * Bp * H, per_cl_b = (size_t)a.T * B * H;     // dh: tile-major (poll_batch); dy: [T][B][H]
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(gran_dh_, 0, (unsigned)(per_cl * a.nchains * a.nlayers * 4), 0x00020000);
    const unsigned parity = epoch & 1u;
    gu32* g_own = (gu32*)gran_dh_ + PBSED_GM_RING_BASE(chain, a.nlayers, layer, per_cl, role.by, H, role.bx);
    gu32* g_dy = (gu32*)gran_dy_ + (size_t)(chain * (a.nlayers - 1) + (layer < top ? layer : 0)) * per_cl_b;
    const int u = tid & 15, bb = (tid >> 4) & 15, b = b0 + bb, j = j0 + u;     // see gru_granule_fwd_body
    const int red_w = lq * 80 + lr * 4, red_r = (u >> 2) * 80 + bb * 4 + (u & 3);
    const bool bv = tid < 256 && b < B;
    const bool rowv = (b0 + lr) < B;
    const int sl = bv ? a.seq_len[b] : 0;
    // ring: contracts dgh of its own step done before (t_next) with W_hh; projection: dgi_t of the layer above with
    // that layer's W_ih.  Every wave takes H/NW hidden units jj = k0 + n*16 + lq*4 + {0..3} (the poll's load pattern).
    const int k0 = (is_mfma ? wave : 0) * NL * 16;
    const GruStackLayer& X = is_proj ? a.lc[chain][layer + 1] : L;
    constexpr int NM = (NL + 1) / 2;                 // X3: bf16 MFMAs per product term (see gru_granule_fwd_body)
    float4 wv[XS ? 1 : NL][3];
    BfOp<XS> w3[XS ? NM : 1][3];
    if (is_mfma) {
        const float* W = (is_proj ? L.w_ih : L.w_hh) + (size_t)(j0 + lr) * G + k0 + lq * 4;
        if constexpr (XS > 0) {
#pragma unroll
            for (int m = 0; m < NM; ++m)
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    w3[m][g] = make_op<XS>(*reinterpret_cast<const float4*>(W + g * H + (2 * m) * 16),
                                        2 * m + 1 < NL ? *reinterpret_cast<const float4*>(W + g * H + (2 * m + 1) * 16)
                                                       : make_float4(0.f, 0.f, 0.f, 0.f));
        } else {
#pragma unroll
            for (int n = 0; n < NL; ++n)
#pragma unroll
                for (int g = 0; g < 3; ++g) wv[n][g] = *reinterpret_cast<const float4*>(W + g * H + n * 16);
        }
    }
    const unsigned cl_src = chain * a.nlayers + (is_proj ? layer + 1 : layer);
    const unsigned voff0 = PBSED_GM_POLL_OFFSET0(cl_src, per_cl, role.by, H, k0, lr, lq);
    const unsigned step_t = (unsigned)(Bp * H * 4);
    float dhz_prev = 0.f;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const PollPacer pacer{(GW || threadIdx.x >= 256) ? a.poll_delay : a.poll_delay_gate};
    if (tid == 0) s_err = 0;
    __syncthreads();

    float pr[NL][4], pz[NL][4], pn[NL][4];
    float c_r = 0.f, c_z = 0.f, c_n = 0.f, c_nr = 0.f, z = 0.f, dyv = 0.f;
    float x_r = 0.f, x_z = 0.f, x_n = 0.f, x_nr = 0.f, x_zz = 0.f, x_dy = 0.f;    // the same for the next step
    auto load_operands = [&](int bs) __attribute__((always_inline)) {
        const int s = a.T - 1 - bs;
        const int t = rev ? a.T - 1 - s : s;
        const int tx = is_proj ? t : (rev ? t - 1 : t + 1);
        const bool act = (is_proj || bs > 0) && rowv && is_mfma;
#pragma unroll
        for (int n = 0; n < NL; ++n) {
            float4 vr = zero4, vz = zero4, vn = zero4;
            if (act) {
                const float* sv = X.save + ((size_t)tx * B + b0 + lr) * 5 * H + k0 + n * 16 + lq * 4;
                vr = *reinterpret_cast<const float4*>(sv);
                vz = *reinterpret_cast<const float4*>(sv + H);
                vn = *reinterpret_cast<const float4*>(sv + (is_proj ? 2 : 3) * H);
            }
            pr[n][0] = vr.x; pr[n][1] = vr.y; pr[n][2] = vr.z; pr[n][3] = vr.w;
            pz[n][0] = vz.x; pz[n][1] = vz.y; pz[n][2] = vz.z; pz[n][3] = vz.w;
            pn[n][0] = vn.x; pn[n][1] = vn.y; pn[n][2] = vn.z; pn[n][3] = vn.w;
        }
    };
    auto load_own = [&](int bs) __attribute__((always_inline)) {
        const int s = a.T - 1 - bs;
        const int t = rev ? a.T - 1 - s : s;
        if (bv && !is_proj) {
            const size_t tb = (size_t)t * B + b;
            const float* sv = L.save + tb * 5 * H + j;
            x_r = sv[0]; x_z = sv[H]; x_n = sv[2 * H]; x_nr = sv[3 * H]; x_zz = sv[4 * H];
            if (layer == top) x_dy = L.dy[tb * H + j];
        }
    };
    load_operands(0);
    load_own(0);
    const bool prof_blk = a.prof != nullptr && (int)blockIdx.x == a.prof_block;        // see gru_granule_fwd_body
    const bool prof_c = prof_blk && tid == GWV * 64, prof_g = prof_blk && tid == 0;

    for (int bstep = 0; bstep < a.T; ++bstep) {
        const int s = a.T - 1 - bstep;
        const int t = rev ? a.T - 1 - s : s;
        const int tn = rev ? t - 1 : t + 1;
        const bool has_next = bstep > 0;
        const size_t tb = (size_t)t * B + b;
        const int par = bstep & 1;
        const bool prof_now = prof_blk && bstep >= 200 && bstep < 232;
        unsigned long long* pslot = a.prof + (prof_now ? (bstep - 200) * 16 : 0);
        if (prof_now && prof_c) prof_stamp(pslot + 0);
        if (prof_now && prof_g) prof_stamp(pslot + 8);
        if (tid == 0 && (bstep & 31) == 31 && __hip_atomic_load((gu32*)err_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) s_err = 1;
        f32x4 acc[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        c_r = x_r; c_z = x_z; c_n = x_n; c_nr = x_nr; z = x_zz; dyv = x_dy;
        float4 dh4[NL];
        const bool contract = (is_proj || has_next) && is_mfma;
        if (contract) {
            pacer.wait();
            if (prof_now && prof_c) prof_stamp(pslot + 1);
            const int spins = poll_batch<NL>(dh4, rsrc, voff0 + (unsigned)(is_proj ? t : tn) * step_t, parity, rowv, err_flag, ring_local);
            if (prof_now && prof_c) { prof_stamp(pslot + 2); pslot[5] = (unsigned long long)spins; }
        }
        unsigned qd[1] = {0};                         // behind the poll: loads return in order
        if (!is_proj && layer < top && bv)
            qd[0] = __hip_atomic_load(g_dy + tb * H + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (bstep + 1 < a.T) load_own(bstep + 1);
        if constexpr (XS > 0) {
            if (contract) {
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const int n0 = 2 * m, n1 = 2 * m + 1 < NL ? 2 * m + 1 : 2 * m;
                    const float k1 = 2 * m + 1 < NL ? 1.f : 0.f;
                    const float4 d0 = dh4[n0], d1 = make_float4(dh4[n1].x * k1, dh4[n1].y * k1, dh4[n1].z * k1, dh4[n1].w * k1);
                    acc[0] = mfma_op<XS>(w3[m][0], make_op<XS>(make_float4(d0.x * pr[n0][0], d0.y * pr[n0][1], d0.z * pr[n0][2], d0.w * pr[n0][3]),
                                                        make_float4(d1.x * pr[n1][0], d1.y * pr[n1][1], d1.z * pr[n1][2], d1.w * pr[n1][3])), acc[0]);
                    acc[1] = mfma_op<XS>(w3[m][1], make_op<XS>(make_float4(d0.x * pz[n0][0], d0.y * pz[n0][1], d0.z * pz[n0][2], d0.w * pz[n0][3]),
                                                        make_float4(d1.x * pz[n1][0], d1.y * pz[n1][1], d1.z * pz[n1][2], d1.w * pz[n1][3])), acc[1]);
                    acc[2] = mfma_op<XS>(w3[m][2], make_op<XS>(make_float4(d0.x * pn[n0][0], d0.y * pn[n0][1], d0.z * pn[n0][2], d0.w * pn[n0][3]),
                                                        make_float4(d1.x * pn[n1][0], d1.y * pn[n1][1], d1.z * pn[n1][2], d1.w * pn[n1][3])), acc[2]);
                }
            }
        } else if (contract) {
#pragma unroll
            for (int n = 0; n < NL; ++n) {
                acc[0] = mfma16(wv[n][0].x, dh4[n].x * pr[n][0], acc[0]);
                acc[1] = mfma16(wv[n][1].x, dh4[n].x * pz[n][0], acc[1]);
                acc[2] = mfma16(wv[n][2].x, dh4[n].x * pn[n][0], acc[2]);
                acc[0] = mfma16(wv[n][0].y, dh4[n].y * pr[n][1], acc[0]);
                acc[1] = mfma16(wv[n][1].y, dh4[n].y * pz[n][1], acc[1]);
                acc[2] = mfma16(wv[n][2].y, dh4[n].y * pn[n][1], acc[2]);
                acc[0] = mfma16(wv[n][0].z, dh4[n].z * pr[n][2], acc[0]);
                acc[1] = mfma16(wv[n][1].z, dh4[n].z * pz[n][2], acc[1]);
                acc[2] = mfma16(wv[n][2].z, dh4[n].z * pn[n][2], acc[2]);
                acc[0] = mfma16(wv[n][0].w, dh4[n].w * pr[n][3], acc[0]);
                acc[1] = mfma16(wv[n][1].w, dh4[n].w * pz[n][3], acc[1]);
                acc[2] = mfma16(wv[n][2].w, dh4[n].w * pn[n][3], acc[2]);
            }
        }
        if (bstep + 1 < a.T) load_operands(bstep + 1);
        if (is_mfma) {
#pragma unroll
            for (int q = 0; q < 4; ++q) red[par][wave][red_w + q] = acc[0][q] + acc[1][q] + acc[2][q];
        }
        if (prof_now && prof_c) prof_stamp(pslot + 3);
        __syncthreads();
        if (prof_now && prof_c) prof_stamp(pslot + 4);
        if (prof_now && prof_g) prof_stamp(pslot + 9);
        const int err_seen = s_err;                    // acted on behind the publish (see gru_granule_fwd_body)
        if (tid >= 256 && err_seen) return;
        if (bv) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) sum += red[par][w][red_r];
            if (prof_now && prof_g) prof_stamp(pslot + 10);
            if (is_proj) {
                publish(g_dy + tb * H + j, tag_clear(sum), parity);
            } else {
                if (layer < top) {
                    wait_own_granules<1>(qd, g_dy + tb * H + j, 0, parity, err_flag);
                    dyv = __uint_as_float(qd[0] & ~1u);
                }
                float dr = 0.f, dz = 0.f, dn = 0.f, dnr = 0.f, dhzv = 0.f, dh = 0.f;
                if (t < sl) {
                    dh = tag_clear(dyv + (has_next ? sum + dhz_prev : 0.f));       // dh IS the truncated value
                    dn = dh * c_n; dz = dh * c_z; dr = dh * c_r; dnr = dh * c_nr;
                    dhzv = dh * z;
                }
                dhz_prev = dhzv;
                if (ring_local) publish_local(g_own + PBSED_GM_RING_WORD(t, Bp, H, tid), dh, parity);
                else publish(g_own + PBSED_GM_RING_WORD(t, Bp, H, tid), dh, parity);
                if (prof_now && prof_g) prof_stamp(pslot + 11);
                float* dgi = L.dgi + tb * G;
                float* dgh = L.dgh + tb * G;
                dgi[j] = dr; dgi[H + j] = dz; dgi[2 * H + j] = dn;
                dgh[j] = dr; dgh[H + j] = dz; dgh[2 * H + j] = dnr;
            }
        }
        if (prof_now && prof_g) prof_stamp(pslot + 12);
        if (err_seen) return;
    }
}

template <int KB, int NW, int XS>
__global__ __launch_bounds__((NW + 4) * 64) void gru_granule_bwd_gw_kernel(GruStackArgs a, unsigned* gran_dh_, unsigned* gran_dy_,
                                                                         unsigned epoch, unsigned* err_flag) {
    __shared__ float red[2][NW][RED_ROW];
    __shared__ int s_err;
    gru_granule_bwd_body<KB, NW, true, XS>(a, gran_dh_, gran_dy_, epoch, err_flag, red, s_err);
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

// 1-D grid of the XCD-aware role mapping (granule_role): per XCD, ring slots first, then projection slots.
// Returns false when an XCD could not hold all of its blocks at once: the round-robin dispatch gives XCD x the blocks
// with id % 8 = x, i.e. `slots` blocks each, and a scan only makes progress if every active block of it is resident -
// with one (12-wave, register-heavy) block per CU a ring pinned to an XCD can fill its 32 CUs and lock the projection
// blocks of that XCD out (found by tests/sweeps/fuzz_gru.py: H = 512, B = 16).  The caller then keeps the 3-D grid, whose
// blocks spread evenly over the XCDs.
// nb: 16-row batch tiles per block
//
// XCD placement probe (once per device): the XCD-local exchange below is only CORRECT when all workgroups of a ring - block
// ids that are equal mod 8 - run on one XCD.  That is how the dispatcher of an 8-XCD device in SPX mode deals a kernel's
// workgroups (MI355X_MICROARCH.md), but it is a property of the device configuration (6-XCD parts, CPX / DPX partitions and
// CU masks differ), so it is observed instead of assumed: launches of device_cus() single-wave blocks, each recording its
// XCC_ID; local is enabled only if eight distinct XCDs show up and XCC_ID is the SAME function of block id mod 8 in every launch.  The map goes to
// the scan (GruStackArgs::xcc_map), whose local-ring blocks re-check their own placement at every launch (error bit 2).
__global__ void xcc_probe_kernel(unsigned* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}
static std::atomic<int> g_xcd_local_allowed{1};       // pbsed_gru_set_xcd_local (any host thread may call it beside a launching one)
static std::mutex g_xcd_mu;                           // the probe's per-device verdicts below
static int g_xcd_state[64] = {0};                     // per device ordinal: 0 not probed, 1 verified, -1 refused
static bool xcd_placement(hipStream_t s, unsigned* map) {
    std::lock_guard<std::mutex> lock(g_xcd_mu);
    int (&state)[64] = g_xcd_state;
    static unsigned maps[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 63;
    if (state[dev] == 0) {
        state[dev] = -1;
        // three launches: n, n - 5 and n blocks.  The map of the first and the third must agree - a dispatcher that carried its
        // XCD pointer over from one kernel to the next would still deal ids equal mod 8 to one XCD, but the map the scans check
        // themselves against would rotate from launch to launch (the odd-sized launch in between shifts it): refused as well
        const int n = device_cus() < 64 ? 64 : device_cus();
        unsigned* d = nullptr;
        std::vector<unsigned> h(3 * n, 0xffu);
        if (hipMalloc(&d, 3 * n * sizeof(unsigned)) == hipSuccess) {
            const int blocks[3] = {n, n - 5, n};
            for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(xcc_probe_kernel, dim3(blocks[k]), dim3(64), 0, s, d + k * n);
            const bool ran = hipMemcpyAsync(h.data(), d, 3 * n * sizeof(unsigned), hipMemcpyDeviceToHost, s) == hipSuccess &&
                             hipStreamSynchronize(s) == hipSuccess;
            (void)hipFree(d);
            unsigned seen = 0, m = 0;
            bool periodic = ran;
            for (int k = 0; k < 3 && periodic; ++k)
                for (int i = 0; i < blocks[k] && periodic; ++i) {
                    periodic = h[k * n + i] < 16u && h[k * n + i] == h[i & 7];      // every launch: the FIRST launch's map
                    seen |= 1u << (h[k * n + i] & 15u);
                }
            for (int r = 0; r < 8; ++r) m |= (h[r] & 15u) << (4 * r);
            if (periodic && __builtin_popcount(seen) == 8) { state[dev] = 1; maps[dev] = m; }
        }
    }
    *map = maps[dev];
    return state[dev] == 1;
}

static bool granule_xcd_grid(GruStackArgs& a, int H, dim3* grid, hipStream_t s, int nb = 1, bool bwd = false) {
    const int cus_per_xcd = device_cus() / 8;           // per device ordinal (common.h)
    const int nby = (a.B + 16 * nb - 1) / (16 * nb), nj = H / 16;
    const int R = a.nchains * a.nlayers * nby, P = a.nchains * (a.nlayers - 1) * nby;
    const int slots = (R + 7) / 8 * nj + (P * nj + 7) / 8;
    if (slots > cus_per_xcd) return false;
    a.nby = nby;
    a.ring_xcd = nj;
    *grid = dim3(8 * slots, 1, 1);
    // BPTT: the last ring in scan order exchanges its states through its XCD's L2 only (GruStackArgs::local).  Measured on
    // MI355X, B 32, H 256, T 500: two-layer BPTT 1.346 -> 1.30 ms with every ring local (co-located groups), one-layer BiGRU BPTT
    // 2.376 -> 2.273 ms per two launches; the FORWARD scans do not gain (0.937 -> 0.943, 1.550 -> 1.577: an early sc1 poll is
    // parked at the L2 until the write-through store it raced completes and returns at once - a plain look cannot be parked,
    // it returns the stale line and pays a retry) and keep the sc1 exchange.
    // Only where the placement probe has SEEN the ring-per-XCD placement on this device (xcd_placement above).
    static const bool local_off = getenv("PBSED_GRU_XCD_LOCAL") && getenv("PBSED_GRU_XCD_LOCAL")[0] == '0';
    a.local = (bwd && !local_off && g_xcd_local_allowed.load(std::memory_order_relaxed) && xcd_placement(s, &a.xcc_map)) ? 1 : 0;
    return true;
}

// First-poll delays per DEVICE and scan kind (0: two-layer stacks, 1: one-layer scans, 2: two batch tiles per block), in
// 64-clock units: {forward, forward gate waves, BPTT, BPTT gate waves}.  The defaults are the measured optima below;
// PBSED_GRU_POLL_DELAYS="fwd,fwd_gate,bwd,bwd_gate" overrides them, and pbsed_gru_set_poll_delays lets the caller install what it
// measured on ITS device in ITS step (pb_sed_amd/ops.py tunes them at the first scan of every shape: the forward optimum is
// sharp and moves by a unit with clocks and with what ran before the scan, so a value baked in here can detune).
static int* granule_delay_table(int kind) {
    static int tab[64][3][4];
    static bool init[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 63;
    if (!init[dev]) {
        const int def[3][4] = {{25, 6, 20, 0}, {24, 6, 15, 0}, {44, 6, 20, 0}};
        int e4[4];
        const char* e = getenv("PBSED_GRU_POLL_DELAYS");
        const bool over = e && sscanf(e, "%d,%d,%d,%d", &e4[0], &e4[1], &e4[2], &e4[3]) == 4;
        for (int k = 0; k < 3; ++k)
            for (int i = 0; i < 4; ++i) tab[dev][k][i] = over ? e4[i] : def[k][i];
        init[dev] = true;
    }
    return tab[dev][kind];
}

static int granule_capacity(bool bwd, int H, int bf16, int nb);       // co-resident blocks of the scan kernel (below)
// diagnostics (pbsed_gru_set_prof)
static unsigned long long* g_prof_buf = nullptr;
static int g_prof_block = 0;
static void granule_common(GruStackArgs& a) {
    a.prof = g_prof_buf;
    a.prof_block = g_prof_block;
}

static void granule_poll_delays(bool bwd, GruStackArgs& a, int nb = 1) {
    // measured with the tile-major exchange arrays and bf16x3 products (tools/sweep_poll_delays.sh, B = 32, H = 256, T = 500).
    // Two-layer stacks (FBCRNN): forward 21 1.28 ms, 23 0.99, 24 0.94, 25 0.97, 26 1.04, 27 1.10; BPTT 14 1.37, 16 1.35,
    // 18..22 1.334, 24 1.36, 28 1.43.  One-layer scans (the BiGRU layers of the BiCRNN, two launches): forward 20 2.40,
    // 22 1.82, 23 1.78, 24 1.80, 26 1.90; BPTT 8 2.48, 14..17 2.44, 20 2.51, 26 2.70.
    // Two batch tiles per block (forward, 64 clips, two-layer stacks): 24 2.25 ms, 36 1.93, 40 1.79, 44 1.77, 48 1.80, 64 1.88
    // (the two per-chain launches it replaces: 1.90 ms).
    // The forward optimum sits one unit behind a cliff whose position moves by a unit with what ran before the scan (24 was
    // best with the convolution + transpose producing gi, 25 with the time-major projection: 24 1.05 ms, 25 0.97, 26 0.99): the
    // default is the safe side.  A per-wave feedback on missed first polls was tried: the extra state alone costs 0.2 ms.
    int* use = granule_delay_table(nb == 2 ? 2 : a.nlayers == 1 ? 1 : 0);
    a.poll_delay = use[bwd ? 2 : 0];
    a.poll_delay_gate = use[bwd ? 3 : 1];
}

// Ring-per-XCD placement (granule_role), both scan directions.  Measured on MI355X with the paced 4-byte polls: forward 1.35 ms
// with / 1.48 ms without; BPTT (round 3, tile-major exchange, bf16x3 products, delays tuned in place) 1.285 with / 1.335 without.
// Granule-exchange persistent forward scan (see gru_granule_fwd_kernel).  granules: device uint32 workspace of
// nchains*T*Bp*H*(nlayers + 3*(nlayers-1)) words, Bp = B rounded up to 16 (h_t of every layer, then the projected inputs of layers > 0) that
// must be ZERO before its first use; `epoch` must be odd on the first use of a workspace and change parity with
// every call that uses it (the words of the previous call then never match); same T, B, H for the life of a
// workspace.  err_flag: device uint32 (0 on entry; non-zero after a hand-off timed out: re-zero the workspace).
}  // extern "C"

// bf16: plain bf16 operands of the recurrent / projection products (fp32 state, accumulation and gate maths) - the scans of
// the bf16 training mode (BASELINE.json configs[2]); else fp32-class products (bf16x3).
static int gru_stack_fwd_granule_impl(int nchains, int nlayers, const float* const* gi0, const float* const* w_ih,
                                      const float* const* b_ih, const float* const* w_hh, const float* const* b_hh,
                                      float* const* hs, float* const* save, const int* reverse, const int* seq_len, int B,
                                      int H, int T, unsigned int* granules, unsigned int epoch, unsigned int* err_flag,
                                      int bf16, void* stream) {
    if (int e = stack_check(nchains, nlayers, B, H, T)) return e;
    if (epoch == 0 || !granules || !err_flag) { set_error("gru_stack_fwd_granule: need workspace and epoch != 0"); return PBSED_E_ARG; }
    const size_t Bp = (size_t)(B + 15) / 16 * 16;            // the exchanged states are stored in whole 16-row batch tiles
    if ((size_t)nchains * nlayers * T * Bp * H * 4 >= (1ull << 32)) { set_error("gru_stack_fwd_granule: workspace over 4 GiB"); return PBSED_E_ARG; }
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
    const int ngroups = nchains * (2 * nlayers - 1);     // rings + projection groups (see GranuleRole)
    // dedicated gate waves (1.35 -> 1.21 ms / 1.65 -> 1.49 ms at B = 32, H = 256, T = 500) and, up to H = 256, bf16x3 operands on
    // the bf16 MFMA (forward 1.13 -> 0.94 ms, BPTT 1.40 -> 1.33 ms); H = 512 keeps fp32-MFMA operands (the three-way W fragments
    // of K = 64 per wave spill: forward 3.2 -> 4.4 ms, BPTT 4.7 -> 7.5 ms measured in round 4)
    // every block has to be co-resident (one per CU, 7/8 of the device at most): past that, two batch tiles per block
    const int nb = (ngroups * (H / 16) * ((B + 15) / 16) > device_cus() * 7 / 8) ? 2 : 1;
    granule_poll_delays(false, a, nb);
    granule_common(a);
    dim3 grid(H / 16, (B + 16 * nb - 1) / (16 * nb), ngroups);
    if ((int)(grid.x * grid.y * grid.z) > granule_capacity(false, H, bf16, nb)) {
        set_error("gru_stack_fwd_granule: %u blocks cannot be co-resident on this device (capacity %d): use pbsed_gru_stack_fwd",
                  grid.x * grid.y * grid.z, granule_capacity(false, H, bf16, nb));
        return PBSED_E_UNSUPPORTED;
    }
    granule_xcd_grid(a, H, &grid, (hipStream_t)stream, nb);                    // one ring per XCD when every XCD can hold its share (else the 3-D grid)
    unsigned* gran_gi = granules + (size_t)nchains * nlayers * T * Bp * H;
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_GW(KB_, NW_, X3_, NB_)                                                                                  \
    do {                                                                                                             \
        auto kern = gru_granule_fwd_gw_kernel<KB_, NW_, X3_, NB_>;                                                   \
        const size_t lds = (size_t)2 * NW_ * NB_ * 3 * RED_ROW * sizeof(float);                                       \
        PBSED_DYN_LDS_ONCE(kern, lds);                                                                               \
        hipLaunchKernelGGL(kern, grid, dim3((NW_ + 4) * 64), lds, s, a, granules, gran_gi, epoch, err_flag);         \
    } while (0)
#define LAUNCH_GRANULE(KB_, NW_)                                                                                     \
    do {                                                                                                             \
        if (nb == 2) {                                                                                               \
            if (bf16 && KB_ < 4) LAUNCH_GW(KB_, NW_, 1, 2);                                                          \
            else if (KB_ < 4) LAUNCH_GW(KB_, NW_, 3, 2);                                                             \
            else LAUNCH_GW(KB_, NW_, 0, 2);                                                                          \
        } else {                                                                                                     \
            if (bf16) LAUNCH_GW(KB_, NW_, 1, 1); else LAUNCH_GW(KB_, NW_, 3, 1);                                     \
        }                                                                                                            \
    } while (0)
    switch (H) {
        case 64: LAUNCH_GRANULE(1, 4); break;
        case 128: LAUNCH_GRANULE(1, 8); break;
        case 256: LAUNCH_GRANULE(2, 8); break;
        default: LAUNCH_GRANULE(4, 8); break;
    }
#undef LAUNCH_GRANULE
#undef LAUNCH_GW
    return check_launch("gru_stack_fwd_granule");
}


// Granule-exchange persistent BPTT.  granules: device uint32 workspace of nchains*T*Bp*H*(2*nlayers-1) words (Bp as above)
// (dh_t of every layer, then dy_t of the layers below the top), zero before first use, epoch parity as above.
static int gru_stack_bwd_granule_impl(int nchains, int nlayers, const float* const* w_hh_t, const float* const* w_ih_up_t,
                                      const float* const* hs, const float* const* save, const float* const* dy_top,
                                      float* const* dgi, float* const* dgh, const int* reverse, const int* seq_len, int B, int H,
                                      int T, unsigned int* granules, unsigned int epoch, unsigned int* err_flag, int bf16,
                                      void* stream) {
    if (int e = stack_check(nchains, nlayers, B, H, T)) return e;
    if (epoch == 0 || !granules || !err_flag) { set_error("gru_stack_bwd_granule: need workspace and epoch != 0"); return PBSED_E_ARG; }
    const size_t Bp = (size_t)(B + 15) / 16 * 16;
    if ((size_t)nchains * nlayers * T * Bp * H * 4 >= (1ull << 32)) { set_error("gru_stack_bwd_granule: workspace over 4 GiB"); return PBSED_E_ARG; }
    GruStackArgs a{};
    for (int c = 0; c < nchains; ++c) {
        a.reverse[c] = reverse[c];
        for (int l = 0; l < nlayers; ++l) {
            GruStackLayer& L = a.lc[c][l];
            const int i = c * nlayers + l;
            L.w_hh = w_hh_t[i]; L.w_ih = l < nlayers - 1 ? w_ih_up_t[i] : nullptr;
            L.hs = const_cast<float*>(hs[i]); L.save = const_cast<float*>(save[i]);
            L.dy = l == nlayers - 1 ? dy_top[c] : nullptr;
            L.dgi = dgi[i]; L.dgh = dgh[i];
        }
    }
    a.seq_len = seq_len; a.B = B; a.T = T; a.nchains = nchains; a.nlayers = nlayers;
    const int ngroups = nchains * (2 * nlayers - 1);
    granule_poll_delays(true, a);
    granule_common(a);
    dim3 grid(H / 16, (B + 15) / 16, ngroups);
    if ((int)(grid.x * grid.y * grid.z) > granule_capacity(true, H, bf16, 1)) {
        set_error("gru_stack_bwd_granule: %u blocks cannot be co-resident on this device (capacity %d): use pbsed_gru_stack_bwd",
                  grid.x * grid.y * grid.z, granule_capacity(true, H, bf16, 1));
        return PBSED_E_UNSUPPORTED;
    }
    granule_xcd_grid(a, H, &grid, (hipStream_t)stream, 1, true);
    unsigned* gran_dy = granules + (size_t)nchains * nlayers * T * Bp * H;
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_GRANULE(KB_, NW_)                                                                                     \
    do {                                                                                                             \
        if (bf16 && KB_ < 4) {                                                                                       \
            hipLaunchKernelGGL((gru_granule_bwd_gw_kernel<KB_, NW_, 1>), grid, dim3((NW_ + 4) * 64), 0, s, a,         \
                               granules, gran_dy, epoch, err_flag);                                                  \
        } else if (KB_ < 4) {                                                                                        \
            hipLaunchKernelGGL((gru_granule_bwd_gw_kernel<KB_, NW_, 3>), grid, dim3((NW_ + 4) * 64), 0, s, a,         \
                               granules, gran_dy, epoch, err_flag);                                                  \
        } else {                                                                                                     \
            hipLaunchKernelGGL((gru_granule_bwd_gw_kernel<KB_, NW_, 0>), grid, dim3((NW_ + 4) * 64), 0, s, a,         \
                               granules, gran_dy, epoch, err_flag);                                                  \
        }                                                                                                            \
    } while (0)
    switch (H) {
        case 64: LAUNCH_GRANULE(1, 4); break;
        case 128: LAUNCH_GRANULE(1, 8); break;
        case 256: LAUNCH_GRANULE(2, 8); break;
        default: LAUNCH_GRANULE(4, 8); break;
    }
#undef LAUNCH_GRANULE
    return check_launch("gru_stack_bwd_granule");
}

// Co-residency is a precondition of the persistent scans (every block polls words other blocks publish): how many blocks of
// the kernel a scan of this shape would launch fit on the CURRENT device at once, by the occupancy API (registers, LDS and
// waves of that very instantiation) times the CU count, with the scans' own rule of one block per CU on top - the API may
// report one block per CU more than the hardware admits (MI355X_MICROARCH.md, residency), and a second block on a CU would
// share its memory queue with the first.  0 = the kernel cannot be resident at all.  Cached per (instantiation, device).
template <class Kern>
static int resident_blocks(Kern kern, int threads, size_t lds, int (&cache)[64]) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    int& c = cache[dev & 63];
    if (c == 0) {
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, threads, lds) != hipSuccess) occ = 0;
        c = occ >= 1 ? device_cus() : -1;
    }
    return c > 0 ? c : 0;
}

static int granule_capacity(bool bwd, int H, int bf16, int nb) {
#define CAP_FWD(KB_, NW_)                                                                                                          \
    do {                                                                                                                           \
        static int c1[64], c2[64], c3[64];                                                                                         \
        const size_t lds = (size_t)2 * NW_ * nb * 3 * RED_ROW * sizeof(float);                                                     \
        if (nb == 2) {                                                                                                             \
            if (bf16 && KB_ < 4) return resident_blocks(gru_granule_fwd_gw_kernel<KB_, NW_, 1, 2>, (NW_ + 4) * 64, lds, c1);       \
            if (KB_ < 4) return resident_blocks(gru_granule_fwd_gw_kernel<KB_, NW_, 3, 2>, (NW_ + 4) * 64, lds, c2);               \
            return resident_blocks(gru_granule_fwd_gw_kernel<KB_, NW_, 0, 2>, (NW_ + 4) * 64, lds, c3);                            \
        }                                                                                                                          \
        if (bf16) return resident_blocks(gru_granule_fwd_gw_kernel<KB_, NW_, 1, 1>, (NW_ + 4) * 64, lds, c1);                      \
        return resident_blocks(gru_granule_fwd_gw_kernel<KB_, NW_, 3, 1>, (NW_ + 4) * 64, lds, c2);                                \
    } while (0)
#define CAP_BWD(KB_, NW_)                                                                                                          \
    do {                                                                                                                           \
        static int c1[64], c2[64], c3[64];                                                                                         \
        if (bf16 && KB_ < 4) return resident_blocks(gru_granule_bwd_gw_kernel<KB_, NW_, 1>, (NW_ + 4) * 64, 0, c1);                \
        if (KB_ < 4) return resident_blocks(gru_granule_bwd_gw_kernel<KB_, NW_, 3>, (NW_ + 4) * 64, 0, c2);                        \
        return resident_blocks(gru_granule_bwd_gw_kernel<KB_, NW_, 0>, (NW_ + 4) * 64, 0, c3);                                     \
    } while (0)
    if (!bwd) {
        switch (H) {
            case 64: CAP_FWD(1, 4);
            case 128: CAP_FWD(1, 8);
            case 256: CAP_FWD(2, 8);
            default: CAP_FWD(4, 8);
        }
    }
    switch (H) {
        case 64: CAP_BWD(1, 4);
        case 128: CAP_BWD(1, 8);
        case 256: CAP_BWD(2, 8);
        default: CAP_BWD(4, 8);
    }
#undef CAP_FWD
#undef CAP_BWD
}

extern "C" {

// Blocks of the persistent scan kernel of this shape that can be co-resident on the current device (see granule_capacity):
// a caller runs pbsed_gru_stack_*_granule only for scans of at most this many blocks - nchains * (2 nlayers - 1) * (H / 16) *
// ceil(B / (16 * tiles_per_block)) - and the launch-per-step pbsed_gru_stack_fwd / _bwd otherwise; the granule entry points
// check the same bound and return PBSED_E_UNSUPPORTED instead of launching a scan that could never finish.
// Diagnostics: the persistent scans launched after this call write shader-clock stamps of block `block` (1-D block index of
// the XCD-aware grid), scan steps 200..231, to buf[32][16] (device, 4 KB): slots 0..4 = first contraction wave at step top /
// first poll issued / poll satisfied / partial sums written / past the barrier, 5 = poll attempts that missed, 8..12 = first
// gate wave at step top / past the barrier / partial sums reduced / state published / outputs stored.  buf = NULL switches
// it off.  tools/gru_scan_prof.py prints the timeline.
int pbsed_gru_set_prof(unsigned long long* buf, int block) {
    g_prof_buf = buf;
    g_prof_block = block;
    return PBSED_OK;
}

// The XCD-local exchange of the BPTT scans (GruStackArgs::local): on = 0 switches it off for the process (every ring then
// exchanges through write-through stores and sc1 loads, correct under any workgroup placement), on = 1 allows it again where
// the placement probe (run anew) verifies the device.  The caller turns it off when a scan reports error bit 2.  Returns the old value.
int pbsed_gru_set_xcd_local(int on) {
    const int old = g_xcd_local_allowed.exchange(on != 0, std::memory_order_relaxed);
    if (on) {                                         // allowed again: every device is probed anew at its next BPTT scan
        std::lock_guard<std::mutex> lock(g_xcd_mu);
        for (int& st : g_xcd_state) st = 0;
    }
    return old;
}

int pbsed_gru_get_poll_delays(int kind, int* out4 /*host*/) {
    if (kind < 0 || kind > 2 || !out4) { set_error("gru_get_poll_delays: kind 0..2"); return PBSED_E_ARG; }
    const int* t = granule_delay_table(kind);
    for (int i = 0; i < 4; ++i) out4[i] = t[i];
    return PBSED_OK;
}

int pbsed_gru_set_poll_delays(int kind, int fwd, int fwd_gate, int bwd, int bwd_gate) {
    if (kind < 0 || kind > 2 || fwd < 0 || fwd_gate < 0 || bwd < 0 || bwd_gate < 0 || fwd > 4096 || bwd > 4096) {
        set_error("gru_set_poll_delays: kind 0..2, delays 0..4096 units of 64 clocks");
        return PBSED_E_ARG;
    }
    int* t = granule_delay_table(kind);
    t[0] = fwd; t[1] = fwd_gate; t[2] = bwd; t[3] = bwd_gate;
    return PBSED_OK;
}

int pbsed_gru_granule_capacity(int H, int bwd, int bf16, int tiles_per_block) {
    if (H != 64 && H != 128 && H != 256 && H != 512) return 0;
    return granule_capacity(bwd != 0, H, bf16, tiles_per_block == 2 ? 2 : 1);
}

int pbsed_gru_stack_fwd_granule(int nchains, int nlayers, const float* const* gi0, const float* const* w_ih,
                                const float* const* b_ih, const float* const* w_hh, const float* const* b_hh,
                                float* const* hs, float* const* save, const int* reverse, const int* seq_len, int B,
                                int H, int T, unsigned int* granules, unsigned int epoch, unsigned int* err_flag,
                                void* stream) {
    return gru_stack_fwd_granule_impl(nchains, nlayers, gi0, w_ih, b_ih, w_hh, b_hh, hs, save, reverse, seq_len, B, H, T, granules,
                                      epoch, err_flag, 0, stream);
}

int pbsed_gru_stack_fwd_granule_bf16(int nchains, int nlayers, const float* const* gi0, const float* const* w_ih,
                                     const float* const* b_ih, const float* const* w_hh, const float* const* b_hh,
                                     float* const* hs, float* const* save, const int* reverse, const int* seq_len, int B,
                                     int H, int T, unsigned int* granules, unsigned int epoch, unsigned int* err_flag,
                                     void* stream) {
    return gru_stack_fwd_granule_impl(nchains, nlayers, gi0, w_ih, b_ih, w_hh, b_hh, hs, save, reverse, seq_len, B, H, T, granules,
                                      epoch, err_flag, 1, stream);
}

int pbsed_gru_stack_bwd_granule(int nchains, int nlayers, const float* const* w_hh_t, const float* const* w_ih_up_t,
                                const float* const* hs, const float* const* save, const float* const* dy_top,
                                float* const* dgi, float* const* dgh, const int* reverse, const int* seq_len, int B, int H,
                                int T, unsigned int* granules, unsigned int epoch, unsigned int* err_flag,
                                void* stream) {
    return gru_stack_bwd_granule_impl(nchains, nlayers, w_hh_t, w_ih_up_t, hs, save, dy_top, dgi, dgh, reverse, seq_len, B, H, T,
                                      granules, epoch, err_flag, 0, stream);
}

int pbsed_gru_stack_bwd_granule_bf16(int nchains, int nlayers, const float* const* w_hh_t, const float* const* w_ih_up_t,
                                     const float* const* hs, const float* const* save, const float* const* dy_top,
                                     float* const* dgi, float* const* dgh, const int* reverse, const int* seq_len, int B, int H,
                                     int T, unsigned int* granules, unsigned int epoch, unsigned int* err_flag,
                                     void* stream) {
    return gru_stack_bwd_granule_impl(nchains, nlayers, w_hh_t, w_ih_up_t, hs, save, dy_top, dgi, dgh, reverse, seq_len, B, H, T,
                                      granules, epoch, err_flag, 1, stream);
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
