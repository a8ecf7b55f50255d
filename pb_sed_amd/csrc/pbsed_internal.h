// Internal launch-argument structs shared by the kernel translation units and the C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pbsed {

struct ConvFwdArgs {
    // input activation [B, Cin, F, T] (DGRAD+unpool: [B, Cin, F/2, T] gathered through unpool_idx)
    const float* x;
    const float* wp;        // packed weights [KH*KW][CinP][CoutP] (pack_conv_weights)
    const float* bias;      // [Cout] or null
    const float* scale;     // BN-apply prologue x*scale[c]+shift[c] (null: no prologue, no masking)
    const float* shift;
    const int* seq_len;     // device [B] or null (= T)
    float* y;               // [B, Cout, Fo, T]
    uint8_t* pool_idx;      // POOL: argmax row (0/1) per pooled element, or null
    double* stats;          // epilogue statistics accumulators [Cstat][2] or null
    const uint8_t* unpool_idx;  // DGRAD: pool_idx of the forward conv (input is its pooled grad)
    // DGRAD BN+ReLU backward epilogue (null bx: plain store):
    const float* bx;        // raw forward input of the layer [B, Cout, F, T]
    const float* bmean;     // [Cout]
    const float* binvstd;   // [Cout]
    const float* bscale;    // [Cout] gamma*invstd
    const float* bshift;    // [Cout]
    int B, Cin, Cout, F, T, CinP, CoutP;
    const float* res;       // forward: residual added to the (pooled) output before the statistics, [B, Cout, Fo, T], or null
    int stats_cf;           // statistics per (cout, f) instead of per cout
    int relu;               // prologue applies ReLU
};

struct ConvWgradArgs {
    const float* x;         // forward input of the layer [B, Cin, F, T] (prologue recomputed)
    const float* scale;     // BN-apply prologue (null: none)
    const float* shift;
    const int* seq_len;
    const float* g;         // grad wrt conv output [B, Cout, Fo, T]
    const uint8_t* unpool_idx;  // non-null: layer had (2,1) pool, g is pooled
    float* dw;              // [Cout, Cin, KH, KW] (+=, must be zeroed by caller)
    float* db;              // [Cout] (+=) or null
    int B, Cin, Cout, F, T;
    int relu;
    int bf16;               // bf16-MFMA operands (fp32 accumulation and gradients) where the shape allows
    int nslots;             // > 1: dw/db point at nslots partial copies (stride slot_w / slot_b floats), block x -> x % nslots
    int slot_w, slot_b;
    // BN backward of the NEXT layer's input norm applied in the dY loader (pbsed_conv_bwd_weight_bng): g is that norm's dz
    // (the masked ReLU-backward gradient, same shape as the conv output), the gradient the kernel multiplies with is
    //   dY = k1[c] * dz + k2[c] * gx + k3[c]   for t < gseq[b], 0 beyond      (gx = the raw conv output the norm was applied to),
    // k = gcoef [3][Cout * (g_cf ? Fg : 1)] (pbsed_bn_bwd_coef); the blocks of the first cin tile also WRITE dY to gout
    // (same shape) for the layer's data gradient.  All null: g is dY itself.
    const float* gx;
    const float* gcoef;
    const int* gseq;
    float* gout;
    int g_cf;               // coefficients per (cout, output row) instead of per cout
    int xcd_cols;           // conv_wgrad_pc_kernel: XCD-contiguous column positions (launch_wgrad_cfg, PBSED_WGRAD_XCD_COLS)
};

void conv_fwd_tile_dims(int KH, int KW, int Cin, int Cout, int* ck, int* cout_t);
int conv_fwd_launch(const ConvFwdArgs& a, int KH, int KW, int pool, int dgrad, hipStream_t s);
int conv_wgrad_launch(const ConvWgradArgs& a, int KH, int KW, hipStream_t s);
bool conv_wgrad_bng_supported(int KH, int KW, int Cin, int Cout, int F, int T, int bf16, int per_cf);

}  // namespace pbsed
