// C-ABI glue: error reporting + convolution entry points (see include/pbsed.h for the contract).
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "common.h"
#include "pbsed_internal.h"

namespace pbsed {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return PBSED_E_HIP;
    }
    return PBSED_OK;
}

}  // namespace pbsed

using namespace pbsed;

extern "C" {

const char* pbsed_last_error(void) { return g_err; }

int pbsed_version(void) { return 1; }

int pbsed_conv_fwd(const float* x, const float* w_packed, const float* bias, const float* scale,
                   const float* shift, int relu, const int* seq_len, float* y, unsigned char* pool_idx,
                   double* stats, int stats_per_cf, int B, int Cin, int Cout, int F, int T, int KH, int KW,
                   int pool, void* stream) {
    ConvFwdArgs a{};
    a.x = x; a.wp = w_packed; a.bias = bias; a.scale = scale; a.shift = shift; a.seq_len = seq_len;
    a.y = y; a.pool_idx = pool_idx; a.stats = stats; a.stats_cf = stats_per_cf; a.relu = relu;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.F = F; a.T = T;
    int ck, ct;
    conv_fwd_tile_dims(KH, KW, Cin, Cout, &ck, &ct);
    a.CinP = (Cin + ck - 1) / ck * ck;
    a.CoutP = (Cout + ct - 1) / ct * ct;
    return conv_fwd_launch(a, KH, KW, pool, 0, (hipStream_t)stream);
}

// pbsed_conv_fwd + a residual connection ending at this layer: `residual` [B, Cout, Fo, T] is added to the (biased,
// pooled) output before it is stored and before the statistics are taken, so the next layer's batch norm sees the sum.
int pbsed_conv_fwd_res(const float* x, const float* w_packed, const float* bias, const float* scale,
                       const float* shift, int relu, const int* seq_len, float* y, unsigned char* pool_idx,
                       double* stats, int stats_per_cf, int B, int Cin, int Cout, int F, int T, int KH, int KW,
                       int pool, const float* residual, void* stream) {
    ConvFwdArgs a{};
    a.x = x; a.wp = w_packed; a.bias = bias; a.scale = scale; a.shift = shift; a.seq_len = seq_len;
    a.y = y; a.pool_idx = pool_idx; a.stats = stats; a.stats_cf = stats_per_cf; a.relu = relu; a.res = residual;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.F = F; a.T = T;
    int ck, ct;
    conv_fwd_tile_dims(KH, KW, Cin, Cout, &ck, &ct);
    a.CinP = (Cin + ck - 1) / ck * ck;
    a.CoutP = (Cout + ct - 1) / ct * ct;
    return conv_fwd_launch(a, KH, KW, pool, 0, (hipStream_t)stream);
}

// Data gradient.  g: grad wrt the forward conv's output [B,Cout,Fo,T] (pooled if unpool_idx);
// wd_packed: pack_conv_weights(dgrad=1).  Output dz [B,Cin,F,T]: if bx != null the result is already
// pushed back through mask -> ReLU -> BN-apply of the layer's prologue (dz wrt BN output) and
// stats [Cin][2] accumulates (sum dz, sum dz*xhat); otherwise plain grad wrt the conv input.
int pbsed_conv_bwd_data(const float* g, const float* wd_packed, const unsigned char* unpool_idx,
                        const int* seq_len, float* dz, const float* bx, const float* bmean,
                        const float* binvstd, const float* bscale, const float* bshift, int relu,
                        double* stats, int B, int Cin, int Cout, int F, int T, int KH, int KW,
                        void* stream) {
    ConvFwdArgs a{};
    a.x = g; a.wp = wd_packed; a.seq_len = seq_len; a.y = dz; a.unpool_idx = unpool_idx;
    a.bx = bx; a.bmean = bmean; a.binvstd = binvstd; a.bscale = bscale; a.bshift = bshift;
    a.relu = relu; a.stats = bx ? stats : nullptr;
    a.B = B; a.Cin = Cout; a.Cout = Cin; a.F = F; a.T = T;      // roles swapped
    int ck, ct;
    conv_fwd_tile_dims(KH, KW, Cout, Cin, &ck, &ct);
    a.CinP = (Cout + ck - 1) / ck * ck;
    a.CoutP = (Cin + ct - 1) / ct * ct;
    return conv_fwd_launch(a, KH, KW, 0, 1, (hipStream_t)stream);
}

int pbsed_conv_bwd_weight(const float* x, const float* scale, const float* shift, int relu,
                          const int* seq_len, const float* g, const unsigned char* unpool_idx, float* dw,
                          float* db, int B, int Cin, int Cout, int F, int T, int KH, int KW,
                          void* stream) {
    ConvWgradArgs a{};
    a.x = x; a.scale = scale; a.shift = shift; a.relu = relu; a.seq_len = seq_len; a.g = g;
    a.unpool_idx = unpool_idx; a.dw = dw; a.db = db;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.F = F; a.T = T;
    return conv_wgrad_launch(a, KH, KW, (hipStream_t)stream);
}

// Same contract with bf16-MFMA operands (x after its prologue and dY are rounded to bf16 while staged; products are
// accumulated, reduced and returned in fp32).  Layers with fewer than 32 input or output channels run the fp32 kernels.
int pbsed_conv_bwd_weight_bf16(const float* x, const float* scale, const float* shift, int relu,
                               const int* seq_len, const float* g, const unsigned char* unpool_idx, float* dw,
                               float* db, int B, int Cin, int Cout, int F, int T, int KH, int KW, void* stream) {
    ConvWgradArgs a{};
    a.x = x; a.scale = scale; a.shift = shift; a.relu = relu; a.seq_len = seq_len; a.g = g;
    a.unpool_idx = unpool_idx; a.dw = dw; a.db = db;
    a.B = B; a.Cin = Cin; a.Cout = Cout; a.F = F; a.T = T; a.bf16 = 1;
    return conv_wgrad_launch(a, KH, KW, (hipStream_t)stream);
}

int pbsed_memset_async(void* p, int value, size_t bytes, void* stream) {
    hipError_t e = hipMemsetAsync(p, value, bytes, (hipStream_t)stream);
    if (e != hipSuccess) { set_error("memset: %s", hipGetErrorString(e)); return PBSED_E_HIP; }
    return PBSED_OK;
}

}  // extern "C"
